/* CPU restatement (plain C + OpenMP) of the RadixAttention decode / extend / store path.
 * TEST INFRASTRUCTURE ONLY -- never linked into or called by the product (sglang_amd).
 * Used by tests (cross-checking the numpy oracle at larger sizes) and by bench.py's
 * cpu_baseline leg ("kind": "port").
 *
 * Algorithm follows the reference's native CPU backend:
 *   decode_attention_cpu  (python/sglang/kernels/aot/csrc/cpu/decode.cpp:1586): per
 *     (request, kv head) walk req_to_token[req_pool_idx][0:seq_len], flash-decoding with an
 *     fp32 running max / sum / accumulator; GQA heads of one kv head share the K/V row.
 *   extend_attention_cpu  (python/sglang/kernels/aot/csrc/cpu/extend.cpp:425): prefix from the
 *     paged cache + causal triangle over the new contiguous K/V.
 *   store_cache_cpu       (python/sglang/kernels/aot/csrc/cpu/kvcache.cpp:75): row scatter.
 * and the semantics pinned by oracle/radix_oracle.py (checked against each other in tests).
 * bf16 is converted with a 16-bit shift, math in fp32, outputs rounded to nearest-even.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bf16_to_f32(uint16_t x) {
  uint32_t u = (uint32_t)x << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

/* q [bs,Hq,D] bf16; k_buf/v_buf [slots,Hkv,D] bf16 (NHD); req_to_token int32[*,row_stride];
 * o [bs,Hq,D] bf16.  Returns 0. */
int rxo_decode_bf16(const uint16_t* q, const uint16_t* k_buf, const uint16_t* v_buf, uint16_t* o,
                    const int32_t* req_to_token, int64_t row_stride, const int64_t* req_pool_indices,
                    const int64_t* seq_lens, int bs, int hq, int hkv, int d, float sm_scale,
                    float logit_cap) {
  const int group = hq / hkv;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int b = 0; b < bs; ++b) {
    for (int kvh = 0; kvh < hkv; ++kvh) {
      const int32_t* toks = req_to_token + req_pool_indices[b] * row_stride;
      const int64_t n = seq_lens[b];
      float qf[16][256], acc[16][256], m[16], l[16];
      if (group > 16 || d > 256) continue;
      for (int g = 0; g < group; ++g) {
        const uint16_t* qp = q + ((int64_t)b * hq + kvh * group + g) * d;
        for (int i = 0; i < d; ++i) { qf[g][i] = bf16_to_f32(qp[i]) * sm_scale; acc[g][i] = 0.f; }
        m[g] = -INFINITY; l[g] = 0.f;
      }
      for (int64_t t = 0; t < n; ++t) {
        const int64_t slot = toks[t];
        const uint16_t* kp = k_buf + (slot * hkv + kvh) * d;
        const uint16_t* vp = v_buf + (slot * hkv + kvh) * d;
        float kf[256], vf[256];
        for (int i = 0; i < d; ++i) { kf[i] = bf16_to_f32(kp[i]); vf[i] = bf16_to_f32(vp[i]); }
        for (int g = 0; g < group; ++g) {
          float s = 0.f;
#pragma omp simd reduction(+ : s)
          for (int i = 0; i < d; ++i) s += qf[g][i] * kf[i];
          if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
          if (s > m[g]) {
            const float a = expf(m[g] - s);
            for (int i = 0; i < d; ++i) acc[g][i] *= a;
            l[g] *= a;
            m[g] = s;
          }
          const float p = expf(s - m[g]);
          l[g] += p;
          for (int i = 0; i < d; ++i) acc[g][i] += p * vf[i];
        }
      }
      for (int g = 0; g < group; ++g) {
        uint16_t* op = o + ((int64_t)b * hq + kvh * group + g) * d;
        const float inv = l[g] > 0.f ? 1.f / l[g] : 0.f;
        for (int i = 0; i < d; ++i) op[i] = f32_to_bf16(acc[g][i] * inv);
      }
    }
  }
  return 0;
}

/* extend: q [T,Hq,D], k_ext/v_ext [T,Hkv,D], o [T,Hq,D] bf16; prefix via kv_indptr/kv_indices
 * (int64) into k_buf/v_buf (NHD); qo_indptr int64[bs+1]; causal. */
int rxo_extend_bf16(const uint16_t* q, const uint16_t* k_ext, const uint16_t* v_ext, uint16_t* o,
                    const uint16_t* k_buf, const uint16_t* v_buf, const int64_t* qo_indptr,
                    const int32_t* kv_indptr, const int64_t* kv_indices, int bs, int hq, int hkv,
                    int d, float sm_scale, int causal) {
  const int group = hq / hkv;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int r = 0; r < bs; ++r) {
    for (int h = 0; h < hq; ++h) {
      const int kvh = h / group;
      const int64_t q0 = qo_indptr[r], e = qo_indptr[r + 1] - q0;
      const int64_t p0 = kv_indptr[r], pl = kv_indptr[r + 1] - p0;
      if (d > 256) continue;
      for (int64_t mi = 0; mi < e; ++mi) {
        float qf[256], acc[256], m = -INFINITY, l = 0.f;
        const uint16_t* qp = q + ((q0 + mi) * hq + h) * d;
        for (int i = 0; i < d; ++i) { qf[i] = bf16_to_f32(qp[i]) * sm_scale; acc[i] = 0.f; }
        const int64_t n_ext = causal ? mi + 1 : e;
        for (int64_t t = 0; t < pl + n_ext; ++t) {
          const uint16_t *kp, *vp;
          if (t < pl) {
            const int64_t slot = kv_indices[p0 + t];
            kp = k_buf + (slot * hkv + kvh) * d;
            vp = v_buf + (slot * hkv + kvh) * d;
          } else {
            kp = k_ext + ((q0 + t - pl) * hkv + kvh) * d;
            vp = v_ext + ((q0 + t - pl) * hkv + kvh) * d;
          }
          float s = 0.f;
#pragma omp simd reduction(+ : s)
          for (int i = 0; i < d; ++i) s += qf[i] * bf16_to_f32(kp[i]);
          if (s > m) {
            const float a = expf(m - s);
            for (int i = 0; i < d; ++i) acc[i] *= a;
            l *= a;
            m = s;
          }
          const float p = expf(s - m);
          l += p;
          for (int i = 0; i < d; ++i) acc[i] += p * bf16_to_f32(vp[i]);
        }
        uint16_t* op = o + ((q0 + mi) * hq + h) * d;
        const float inv = l > 0.f ? 1.f / l : 0.f;
        for (int i = 0; i < d; ++i) op[i] = f32_to_bf16(acc[i] * inv);
      }
    }
  }
  return 0;
}

/* store: cache[loc[i]] = src[i] (row_bytes each); loc == skip is skipped; OOB returns -1. */
int rxo_store(const uint8_t* k, const uint8_t* v, uint8_t* kc, uint8_t* vc, const int64_t* loc,
              int64_t n, int64_t row_bytes, int64_t size_limit, int64_t skip) {
  for (int64_t i = 0; i < n; ++i)
    if (loc[i] < 0 || loc[i] >= size_limit) return -1;
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) {
    if (loc[i] == skip) continue;
    memcpy(kc + loc[i] * row_bytes, k + i * row_bytes, row_bytes);
    memcpy(vc + loc[i] * row_bytes, v + i * row_bytes, row_bytes);
  }
  return 0;
}

int rxo_num_threads(void) {
  int n = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp master
    n = omp_get_num_threads();
  }
#endif
  return n;
}
