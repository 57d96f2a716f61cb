// Fused rotary embedding + KV-cache store: q and k are rotated in place and the rotated k plus v go
// straight into the paged pool in the same launch (16-bit pools or fp8 e4m3fn quant-on-write).
//
// Reference: the unfused pair RotaryEmbedding.forward (srt/layers/rotary_embedding/base.py: cos_sin_cache
// [max_pos, rot_dim] = [cos | sin], `_apply_rotary_emb` neox / gptj) followed by
// MHATokenToKVPool.set_kv_buffer (srt/mem_cache/memory_pool.py:2305-2381), and its fused forms
// kernels/ops/kvcache/rope_cache.py:101-… (fused_qk_rope_reshape_and_cache) and
// kernels/jit/csrc/elementwise/rope.cuh.  SURVEY.md §8f rank 3: this removes one read + one write of k
// (and a launch) from every layer of a decode step.
//
//   neox : pairs (i, i + rot/2):  o[i] = x[i] cos_i - x[i+rot/2] sin_i ;  o[i+rot/2] = x[i+rot/2] cos_i + x[i] sin_i
//   gptj : pairs (2i, 2i+1)    :  o[2i] = x[2i] cos_i - x[2i+1] sin_i ;   o[2i+1]   = x[2i+1] cos_i + x[2i] sin_i
// computed in fp32 from the fp32 cache, rounded once to the 16-bit dtype; columns >= rot_dim pass through.
// One wave per (token, head): q heads first, then the kv heads (which also carry the v row to the pool).
#include <cmath>

#include "rx_common.h"

namespace rx {

struct RopeArgs {
  uint16_t* q;
  uint16_t* k;
  const uint16_t* v;
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h;
  int64_t n;
  int32_t hq, hkv, d, dv, rot;
  const int64_t* positions;
  const float* cos_sin;
  int64_t cos_sin_stride;
  int32_t is_neox;
  // pool (optional)
  void* k_buf;
  void* v_buf;
  int32_t page_size, kv_fp8;
  int64_t kps, kts, khs, vps, vts, vhs;
  const void* loc;
  int32_t loc64;
  int64_t size_limit, skip_index;
  float k_scale, v_scale;
  int32_t* err_flag;
};

template <typename T>
__device__ __forceinline__ void put_elem(void* buf, bool fp8, int64_t off, float x, float scale) {
  if (fp8) {
    if (scale != 1.0f) x = T::to_f32(T::from_f32(x / scale));
    const uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0u, false);
    static_cast<uint8_t*>(buf)[off] = static_cast<uint8_t>(r & 0xffu);
  } else {
    static_cast<uint16_t*>(buf)[off] = T::from_f32(x);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rope_store_kernel(const RopeArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int heads = a.hq + a.hkv;
  if (unit >= a.n * heads) return;
  const int64_t t = unit / heads;
  const int hh = static_cast<int>(unit % heads);
  const bool is_k = hh >= a.hq;
  const int h = is_k ? hh - a.hq : hh;
  uint16_t* x = is_k ? a.k + t * a.k_stride_t + h * a.k_stride_h : a.q + t * a.q_stride_t + h * a.q_stride_h;
  const float* cs = a.cos_sin + a.positions[t] * a.cos_sin_stride;
  const int half = a.rot >> 1;

  // pool slot of this token (kv heads only)
  bool store = false;
  int64_t koff = 0, voff = 0;
  if (is_k && a.k_buf) {
    const int64_t idx = load_idx(a.loc, t, a.loc64);
    if (idx != a.skip_index) {
      if (idx < 0 || idx >= a.size_limit) {
        if (lane == 0 && a.err_flag) atomicOr(a.err_flag, RX_DEVERR_SLOT_OOB);
      } else {
        store = true;
        const int64_t pg = idx / a.page_size, off = idx % a.page_size;
        koff = pg * a.kps + off * a.kts + h * a.khs;
        voff = pg * a.vps + off * a.vts + h * a.vhs;
      }
    }
  }
  const bool fp8 = a.kv_fp8 != 0;
  // rotated part: pair index p -> elements (i0, i1)
  for (int p = lane; p < half; p += 64) {
    const int i0 = a.is_neox ? p : 2 * p, i1 = a.is_neox ? p + half : 2 * p + 1;
    const float c = cs[p], s = cs[half + p];
    const float x0 = T::to_f32(x[i0]), x1 = T::to_f32(x[i1]);
    const uint16_t o0 = T::from_f32(x0 * c - x1 * s), o1 = T::from_f32(x1 * c + x0 * s);
    x[i0] = o0;
    x[i1] = o1;
    if (store) {  // the pool holds exactly what attention would read back from k (rounded once)
      put_elem<T>(a.k_buf, fp8, koff + i0, T::to_f32(o0), a.k_scale);
      put_elem<T>(a.k_buf, fp8, koff + i1, T::to_f32(o1), a.k_scale);
    }
  }
  if (store) {
    for (int i = a.rot + lane; i < a.d; i += 64) put_elem<T>(a.k_buf, fp8, koff + i, T::to_f32(x[i]), a.k_scale);
    const uint16_t* vr = a.v + t * a.v_stride_t + h * a.v_stride_h;
    for (int i = lane; i < a.dv; i += 64) put_elem<T>(a.v_buf, fp8, voff + i, T::to_f32(vr[i]), a.v_scale);
  }
}


// ---- fused per-head RMSNorm of q and k + rotary embedding (+ KV store) ----------------------------------------------
// Reference: fused_qk_norm_rope (kernels/ops/attention/fused_qknorm_rope.py:37-100; kernel
// kernels/jit/csrc/elementwise/fused_qknorm_rope.cuh:78-246, frequencies compute_freq :42-63) -- what a QK-norm model
// (Qwen3-class) runs on its qkv projection in front of attention: per (token, head) x <- x * rsqrt(mean(x^2) + eps) * w
// with w = q_weight or k_weight [head_dim], then RoPE on the first rotary_dim columns with frequencies computed on the
// fly, freq_p = base^(-2 p / rotary_dim), under YaRN (factor != 1) blended as
//   ramp = clamp((p - low) / (high' - low), 0, 1), high' = high + 0.001 if |low - high| <= 1e-6,
//   freq_p <- (freq_p / factor) * ramp + freq_p * (1 - ramp),
// theta = position * freq_p, the rotated pair scaled by attention_factor; everything in fp32 from the 16-bit input,
// ONE rounding at the end (the reference kernel's order; its unfused form rounds after the norm as well).  v is not
// touched.  Beyond the reference: frequencies may come from a cos_sin_cache instead (the form RotaryEmbedding keeps),
// fp16 as well as bf16, positions int32 or int64, any head_dim <= 512 that is a multiple of 2, and -- as
// rx_rope_store_kv -- the finished k row and the v row can go to the paged pool in the same launch.
// One wave per (token, head); the normalised row is parked in 2 KiB of LDS per wave so that a lane reaches its
// rotation partner (rotary_dim / 2 columns away, or the adjacent column) whatever the head_dim.
struct QkNormRopeArgs {
  RopeArgs r;
  const uint16_t* q_weight;
  const uint16_t* k_weight;
  float eps, base, factor, low, high, attention_factor;
  float kf;  // -2 / rotary_dim * log2(base), computed on the HOST in double (the fast kernel's exp2 exponent per pair index)
  const void* positions;  // int32 or int64
  int32_t pos64, on_the_fly;
  int32_t cache_vec;  // the cache rows can be read 16 bytes at a time (aligned base and row stride, rotary_dim % 8 == 0)
};

template <typename T>
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(const QkNormRopeArgs qa) {
  const RopeArgs& a = qa.r;
  __shared__ float rows[4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * 4 + wv;
  const int heads = a.hq + a.hkv;
  if (unit >= a.n * heads) return;
  const int64_t t = unit / heads;
  const int hh = static_cast<int>(unit % heads);
  const bool is_k = hh >= a.hq;
  const int h = is_k ? hh - a.hq : hh;
  uint16_t* x = is_k ? a.k + t * a.k_stride_t + h * a.k_stride_h : a.q + t * a.q_stride_t + h * a.q_stride_h;
  const uint16_t* w = is_k ? qa.k_weight : qa.q_weight;
  float* row = rows[wv];
  // ---- RMSNorm
  float ss = 0.f;
  for (int i = lane; i < a.d; i += 64) {
    const float v = T::to_f32(x[i]);
    row[i] = v;
    ss += v * v;
  }
#pragma unroll
  for (int dlt = 32; dlt > 0; dlt >>= 1) ss += __shfl_xor(ss, dlt);
  const float rcp = __frsqrt_rn(ss / static_cast<float>(a.d) + qa.eps);
  for (int i = lane; i < a.d; i += 64) row[i] = row[i] * (rcp * T::to_f32(w[i]));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- pool slot of this token (kv heads only)
  bool store = false;
  int64_t koff = 0, voff = 0;
  if (is_k && a.k_buf) {
    const int64_t idx = load_idx(a.loc, t, a.loc64);
    if (idx != a.skip_index) {
      if (idx < 0 || idx >= a.size_limit) {
        if (lane == 0 && a.err_flag) atomicOr(a.err_flag, RX_DEVERR_SLOT_OOB);
      } else {
        store = true;
        const int64_t pg = idx / a.page_size, off = idx % a.page_size;
        koff = pg * a.kps + off * a.kts + h * a.khs;
        voff = pg * a.vps + off * a.vts + h * a.vhs;
      }
    }
  }
  const bool fp8 = a.kv_fp8 != 0;
  const int half = a.rot >> 1;
  const int64_t pos_i = load_idx(qa.positions, t, qa.pos64);
  const float pos = static_cast<float>(pos_i);
  const float* cs = qa.on_the_fly ? nullptr : a.cos_sin + pos_i * a.cos_sin_stride;
  for (int p = lane; p < half; p += 64) {
    const int i0 = a.is_neox ? p : 2 * p, i1 = a.is_neox ? p + half : 2 * p + 1;
    float c, sn;
    if (qa.on_the_fly) {
      float freq = powf(qa.base, -2.0f * static_cast<float>(p) / static_cast<float>(a.rot));
      if (qa.factor != 1.0f) {  // YaRN (fused_qknorm_rope.cuh:46-60)
        const float high_adj = (fabsf(qa.low - qa.high) <= 1e-6f) ? qa.high + 0.001f : qa.high;
        const float ramp = fminf(fmaxf((static_cast<float>(p) - qa.low) / (high_adj - qa.low), 0.0f), 1.0f);
        const float extr = 1.0f - ramp;
        freq = (freq / qa.factor) * (1.0f - extr) + freq * extr;
      }
      sincosf(pos * freq, &sn, &c);
    } else {
      c = cs[p];
      sn = cs[half + p];
    }
    const float e0 = row[i0], e1 = row[i1];
    const uint16_t o0 = T::from_f32((e0 * c - e1 * sn) * qa.attention_factor), o1 = T::from_f32((e1 * c + e0 * sn) * qa.attention_factor);
    x[i0] = o0;
    x[i1] = o1;
    if (store) {  // the pool holds exactly what attention would read back from k (rounded once)
      put_elem<T>(a.k_buf, fp8, koff + i0, T::to_f32(o0), a.k_scale);
      put_elem<T>(a.k_buf, fp8, koff + i1, T::to_f32(o1), a.k_scale);
    }
  }
  for (int i = a.rot + lane; i < a.d; i += 64) {  // normalised, not rotated
    const uint16_t o = T::from_f32(row[i]);
    x[i] = o;
    if (store) put_elem<T>(a.k_buf, fp8, koff + i, T::to_f32(o), a.k_scale);
  }
  if (store) {
    const uint16_t* vr = a.v + t * a.v_stride_t + h * a.v_stride_h;
    for (int i = lane; i < a.dv; i += 64) put_elem<T>(a.v_buf, fp8, voff + i, T::to_f32(vr[i]), a.v_scale);
  }
}


// The fast form: D / 8 lanes per head (16-byte loads and stores, 8 elements per lane), 64 / (D / 8) heads per wave, the
// sum of squares reduced and the neox partner fetched by lane shuffles inside the head's lane group -- no LDS, four
// times the bytes in flight per wave of the generic kernel above (which moved 1 TB/s: one 2-byte element per lane and
// load, an LDS round trip, libm's sincosf with its full range reduction).  Angles: freq = exp2(-2 p / rot * log2 base) on
// the transcendental unit, the angle taken to revolutions and reduced by fract() in front of v_sin / v_cos -- fp32
// throughout, as accurate as the fp32 angle itself (the reference kernel uses __sincosf on the same product).
// For head dims 64 / 128 / 256 / 512 with contiguous, 16-byte aligned head rows; neox needs rotary_dim % 16 == 0.
template <typename T, int LPH, bool NORM = true>
__global__ __launch_bounds__(256) void qknorm_rope_fast_kernel(const QkNormRopeArgs qa) {
  const RopeArgs& a = qa.r;
  constexpr int HPW = 64 / LPH;   // heads per wave and pass
  constexpr int UNR = 1;          // passes per wave (2, loads issued together, measured no faster: the kernel is past its latency bound)
  constexpr int D = LPH * 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int sub = lane % LPH, grp = lane / LPH;
  const int heads = a.hq + a.hkv;
  const int64_t total = a.n * heads;
  const int base_i = 8 * sub;
  const int half = a.rot >> 1;
  // (from the host: the approximate __log2f's relative error is multiplied by the position in the angle -- at 128 k it put
  // the fast and the generic kernel noticeably apart)
  const float kf = qa.kf;
  int64_t tt[UNR];
  int hd[UNR];
  bool isk[UNR], live[UNR];
  uint16_t* xp[UNR];
  u32x4 raw[UNR];
  int64_t pos_i[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    int64_t unit = ((static_cast<int64_t>(blockIdx.x) * 4 + wv) * UNR + u) * HPW + grp;
    live[u] = unit < total;
    if (!live[u]) unit = total - 1;  // (keeps the lane group's shuffles well defined; nothing is written)
    tt[u] = unit / heads;
    const int hh = static_cast<int>(unit % heads);
    isk[u] = hh >= a.hq;
    hd[u] = isk[u] ? hh - a.hq : hh;
    xp[u] = (isk[u] ? a.k + tt[u] * a.k_stride_t + hd[u] * a.k_stride_h : a.q + tt[u] * a.q_stride_t + hd[u] * a.q_stride_h) + base_i;
    raw[u] = *reinterpret_cast<const u32x4*>(xp[u]);
    pos_i[u] = load_idx(qa.positions, tt[u], qa.pos64);
  }
  u32x4 wq = {0, 0, 0, 0}, wk = {0, 0, 0, 0};
  if constexpr (NORM) {
    wq = *reinterpret_cast<const u32x4*>(qa.q_weight + base_i);
    wk = *reinterpret_cast<const u32x4*>(qa.k_weight + base_i);
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int64_t t = tt[u];
    const int h = hd[u];
    const bool is_k = isk[u];
    const u32x4 wraw = is_k ? wk : wq;
    float e[8], w[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      e[2 * i] = T::to_f32(static_cast<uint16_t>(raw[u][i] & 0xffffu));
      e[2 * i + 1] = T::to_f32(static_cast<uint16_t>(raw[u][i] >> 16));
      w[2 * i] = T::to_f32(static_cast<uint16_t>(wraw[i] & 0xffffu));
      w[2 * i + 1] = T::to_f32(static_cast<uint16_t>(wraw[i] >> 16));
    }
    if constexpr (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += e[i] * e[i];
#pragma unroll
      for (int m = LPH / 2; m > 0; m >>= 1) ss += __shfl_xor(ss, m);
      const float rcp = __frsqrt_rn(ss / static_cast<float>(D) + qa.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] *= rcp * w[i];
    }
    // ---- RoPE
    const float pos = static_cast<float>(pos_i[u]);
    const float* cs = qa.on_the_fly ? nullptr : a.cos_sin + pos_i[u] * a.cos_sin_stride;
    auto cos_sin_of = [&](int p, float& c, float& sn) {
      if (qa.on_the_fly) {
        float freq = __builtin_amdgcn_exp2f(static_cast<float>(p) * kf);
        if (qa.factor != 1.0f) {  // YaRN (fused_qknorm_rope.cuh:46-60)
          const float high_adj = (fabsf(qa.low - qa.high) <= 1e-6f) ? qa.high + 0.001f : qa.high;
          const float ramp = fminf(fmaxf((static_cast<float>(p) - qa.low) / (high_adj - qa.low), 0.0f), 1.0f);
          const float extr = 1.0f - ramp;
          freq = (freq / qa.factor) * (1.0f - extr) + freq * extr;
        }
        float rev = pos * (freq * 0.15915494309189535f);
        rev -= floorf(rev);
        sn = __builtin_amdgcn_sinf(rev);
        c = __builtin_amdgcn_cosf(rev);
      } else {
        c = cs[p];
        sn = cs[half + p];
      }
    };
    float o[8];
    if (a.is_neox) {
      // partner element = the same slot of the lane rot / 16 lanes away inside the head's group
      const int hl = half >> 3;                       // lanes per half
      const bool in_rot = base_i < a.rot;             // (rot % 16 == 0: a lane is wholly inside or outside)
      const bool low = sub < hl;
      const int partner = lane + (in_rot ? (low ? hl : -hl) : 0);
      const int p0 = base_i - (low ? 0 : half);       // first pair index of this lane: a multiple of 8
      float cv[8], sv[8];
      if (in_rot) {
        if (qa.on_the_fly || !qa.cache_vec) {
#pragma unroll
          for (int i = 0; i < 8; ++i) cos_sin_of(p0 + i, cv[i], sv[i]);
        } else {  // the cache row's 8 cosines and 8 sines of this lane as four 16-byte loads (16 scalar loads cost 2x the kernel)
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + p0), c1 = *reinterpret_cast<const f32x4*>(cs + p0 + 4);
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + half + p0), s1 = *reinterpret_cast<const f32x4*>(cs + half + p0 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            cv[i] = c0[i];
            cv[4 + i] = c1[i];
            sv[i] = s0[i];
            sv[4 + i] = s1[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float other = __shfl(e[i], partner);
        o[i] = in_rot ? (low ? e[i] * cv[i] - other * sv[i] : e[i] * cv[i] + other * sv[i]) * qa.attention_factor : e[i];
      }
    } else {
      float cv[4], sv[4];
      const int p0 = base_i >> 1;                     // a multiple of 4
      if (!qa.on_the_fly && qa.cache_vec && base_i + 8 <= a.rot) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + p0), s0 = *reinterpret_cast<const f32x4*>(cs + half + p0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          cv[i] = c0[i];
          sv[i] = s0[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (base_i + 2 * i + 1 < a.rot) cos_sin_of(p0 + i, cv[i], sv[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        if (base_i + i + 1 < a.rot) {
          const float c = cv[i >> 1], sn = sv[i >> 1];
          o[i] = (e[i] * c - e[i + 1] * sn) * qa.attention_factor;
          o[i + 1] = (e[i + 1] * c + e[i] * sn) * qa.attention_factor;
        } else {
          o[i] = e[i];
          o[i + 1] = e[i + 1];
        }
      }
    }
    if (!live[u]) continue;
    u32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = pack2<T>(o[2 * i], o[2 * i + 1]);
    *reinterpret_cast<u32x4*>(xp[u]) = out;
    // ---- pool store (kv heads only)
    if (is_k && a.k_buf) {
      const int64_t idx = load_idx(a.loc, t, a.loc64);
      if (idx != a.skip_index) {
        if (idx < 0 || idx >= a.size_limit) {
          if (sub == 0 && a.err_flag) atomicOr(a.err_flag, RX_DEVERR_SLOT_OOB);
        } else {
          const int64_t pg = idx / a.page_size, off = idx % a.page_size;
          const int64_t koff = pg * a.kps + off * a.kts + h * a.khs + base_i;
          const int64_t voff = pg * a.vps + off * a.vts + h * a.vhs + base_i;
          const u32x4 vraw = *reinterpret_cast<const u32x4*>(a.v + t * a.v_stride_t + h * a.v_stride_h + base_i);
          if (a.kv_fp8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              put_elem<T>(a.k_buf, true, koff + 2 * i, T::to_f32(static_cast<uint16_t>(out[i] & 0xffffu)), a.k_scale);
              put_elem<T>(a.k_buf, true, koff + 2 * i + 1, T::to_f32(static_cast<uint16_t>(out[i] >> 16)), a.k_scale);
              put_elem<T>(a.v_buf, true, voff + 2 * i, T::to_f32(static_cast<uint16_t>(vraw[i] & 0xffffu)), a.v_scale);
              put_elem<T>(a.v_buf, true, voff + 2 * i + 1, T::to_f32(static_cast<uint16_t>(vraw[i] >> 16)), a.v_scale);
            }
          } else {
            *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.k_buf) + koff) = out;
            *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(a.v_buf) + voff) = vraw;
          }
        }
      }
    }
  }
}

}  // namespace rx

using namespace rx;

extern "C" int rx_rope_store_kv(void* q, void* k, const void* v, int64_t q_stride_t, int64_t q_stride_h,
                                int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                int64_t n, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim,
                                int rotary_dim, const int64_t* positions, const float* cos_sin_cache,
                                int64_t cos_sin_stride, int is_neox, const rx_kv_layout* lay, const void* loc,
                                int loc_is_i64, int64_t size_limit, int64_t skip_index, float k_scale,
                                float v_scale, int dtype, int32_t* err_flag, void* stream) {
  RX_RANGE("rx_rope_store_kv");
  RX_REQUIRE(n >= 0, "rx_rope_store_kv: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(q && k && positions && cos_sin_cache, "rx_rope_store_kv: null pointer");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_rope_store_kv: dtype %d", dtype);
  RX_REQUIRE(num_q_heads > 0 && num_kv_heads > 0 && head_dim > 0, "rx_rope_store_kv: bad head geometry");
  RX_REQUIRE(rotary_dim > 0 && rotary_dim % 2 == 0 && rotary_dim <= head_dim,
             "rx_rope_store_kv: rotary_dim %d must be even and <= head_dim %d", rotary_dim, head_dim);
  RopeArgs a{};
  a.q = static_cast<uint16_t*>(q);
  a.k = static_cast<uint16_t*>(k);
  a.v = static_cast<const uint16_t*>(v);
  a.q_stride_t = q_stride_t; a.q_stride_h = q_stride_h;
  a.k_stride_t = k_stride_t; a.k_stride_h = k_stride_h;
  a.v_stride_t = v_stride_t; a.v_stride_h = v_stride_h;
  a.n = n; a.hq = num_q_heads; a.hkv = num_kv_heads; a.d = head_dim; a.dv = v_head_dim; a.rot = rotary_dim;
  a.positions = positions; a.cos_sin = cos_sin_cache; a.cos_sin_stride = cos_sin_stride; a.is_neox = is_neox;
  if (lay) {
    RX_REQUIRE(v && loc && lay->k_buf && lay->v_buf && v_head_dim > 0, "rx_rope_store_kv: pool store needs v, loc and the layout's buffers");
    RX_REQUIRE(lay->page_size >= 1 && size_limit > 0, "rx_rope_store_kv: bad page_size / size_limit");
    RX_REQUIRE(k_scale > 0.f && v_scale > 0.f, "rx_rope_store_kv: scales must be > 0");
    a.k_buf = const_cast<void*>(lay->k_buf); a.v_buf = const_cast<void*>(lay->v_buf);
    a.page_size = lay->page_size; a.kv_fp8 = lay->kv_fp8;
    a.kps = lay->k_page_stride; a.kts = lay->k_tok_stride; a.khs = lay->k_head_stride;
    a.vps = lay->v_page_stride; a.vts = lay->v_tok_stride; a.vhs = lay->v_head_stride;
    a.loc = loc; a.loc64 = loc_is_i64; a.size_limit = size_limit; a.skip_index = skip_index;
    a.k_scale = k_scale; a.v_scale = v_scale; a.err_flag = err_flag;
  }
  const int64_t units = n * (num_q_heads + num_kv_heads);
  auto s = static_cast<hipStream_t>(stream);
  // round 4: the 16-byte / lane-shuffle form of the QK-norm kernel below, without the norm (Llama-3-8B heads, 16 Ki tokens
  // with the store: 148.6 us = 2.9 TB/s on the one-element-per-lane kernel above), for power-of-two head dims with
  // contiguous, 16-byte aligned rows; everything else stays on the kernel above
  const bool pow2 = head_dim == 64 || head_dim == 128 || head_dim == 256 || head_dim == 512;
  bool fast = pow2 && (is_neox ? rotary_dim % 16 == 0 : true) &&
              ((q_stride_t | q_stride_h | k_stride_t | k_stride_h) % 8 == 0) && (((uintptr_t)q | (uintptr_t)k) & 15) == 0;
  if (fast && lay) {
    fast = v_head_dim == head_dim && ((v_stride_t | v_stride_h) % 8 == 0) && ((uintptr_t)v & 15) == 0 &&
           (lay->kv_fp8 || (((lay->k_page_stride | lay->k_tok_stride | lay->k_head_stride | lay->v_page_stride |
                              lay->v_tok_stride | lay->v_head_stride) % 8 == 0) &&
                            (((uintptr_t)lay->k_buf | (uintptr_t)lay->v_buf) & 15) == 0));
  }
  if (fast) {
    QkNormRopeArgs qa{};
    qa.r = a;
    qa.attention_factor = 1.0f;
    qa.positions = positions;
    qa.pos64 = 1;
    qa.on_the_fly = 0;
    qa.cache_vec = ((uintptr_t)cos_sin_cache & 15) == 0 && cos_sin_stride % 4 == 0 && rotary_dim % 8 == 0;
    const int lph = head_dim / 8, hpw = 64 / lph;
    const dim3 grid(static_cast<unsigned>((units + 4 * hpw - 1) / (4 * hpw))), block(256);
#define RX_ROPE_FAST(TT)                                                                               \
  do {                                                                                                 \
    if (lph == 8) hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 8, false>), grid, block, 0, s, qa);        \
    else if (lph == 16) hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 16, false>), grid, block, 0, s, qa); \
    else if (lph == 32) hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 32, false>), grid, block, 0, s, qa); \
    else hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 64, false>), grid, block, 0, s, qa);                \
  } while (0)
    if (dtype == RX_BF16) RX_ROPE_FAST(BF16);
    else RX_ROPE_FAST(F16);
#undef RX_ROPE_FAST
    return check_launch("rx_rope_store_kv");
  }
  const dim3 grid(static_cast<unsigned>((units + 3) / 4)), block(256);
  if (dtype == RX_BF16) hipLaunchKernelGGL(rope_store_kernel<BF16>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(rope_store_kernel<F16>, grid, block, 0, s, a);
  return check_launch("rx_rope_store_kv");
}

extern "C" int rx_qknorm_rope_store_kv(void* q, void* k, const void* v, int64_t q_stride_t, int64_t q_stride_h,
                                       int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                       int64_t n, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim,
                                       int rotary_dim, const void* q_weight, const void* k_weight, float eps,
                                       const void* positions, int positions_is_i64, float base, float factor, float low,
                                       float high, float attention_factor, const float* cos_sin_cache,
                                       int64_t cos_sin_stride, int is_neox, const rx_kv_layout* lay, const void* loc,
                                       int loc_is_i64, int64_t size_limit, int64_t skip_index, float k_scale,
                                       float v_scale, int dtype, int32_t* err_flag, void* stream) {
  RX_RANGE("rx_qknorm_rope_store_kv");
  RX_REQUIRE(n >= 0, "rx_qknorm_rope_store_kv: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(q && k && positions && q_weight && k_weight, "rx_qknorm_rope_store_kv: null pointer");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_qknorm_rope_store_kv: dtype %d", dtype);
  RX_REQUIRE(num_q_heads > 0 && num_kv_heads > 0 && head_dim > 0 && head_dim <= 512 && head_dim % 2 == 0,
             "rx_qknorm_rope_store_kv: head_dim %d must be even and <= 512 (heads %d / %d)", head_dim, num_q_heads, num_kv_heads);
  RX_REQUIRE(rotary_dim > 0 && rotary_dim % 2 == 0 && rotary_dim <= head_dim,
             "rx_qknorm_rope_store_kv: rotary_dim %d must be even and <= head_dim %d", rotary_dim, head_dim);
  RX_REQUIRE(eps >= 0.f, "rx_qknorm_rope_store_kv: eps < 0");
  RX_REQUIRE(cos_sin_cache || (base > 0.f && factor > 0.f), "rx_qknorm_rope_store_kv: no cos_sin_cache and base / factor not > 0");
  QkNormRopeArgs qa{};
  RopeArgs& a = qa.r;
  a.q = static_cast<uint16_t*>(q);
  a.k = static_cast<uint16_t*>(k);
  a.v = static_cast<const uint16_t*>(v);
  a.q_stride_t = q_stride_t; a.q_stride_h = q_stride_h;
  a.k_stride_t = k_stride_t; a.k_stride_h = k_stride_h;
  a.v_stride_t = v_stride_t; a.v_stride_h = v_stride_h;
  a.n = n; a.hq = num_q_heads; a.hkv = num_kv_heads; a.d = head_dim; a.dv = v_head_dim; a.rot = rotary_dim;
  a.cos_sin = cos_sin_cache; a.cos_sin_stride = cos_sin_stride; a.is_neox = is_neox;
  qa.q_weight = static_cast<const uint16_t*>(q_weight);
  qa.k_weight = static_cast<const uint16_t*>(k_weight);
  qa.kf = static_cast<float>(-2.0 / static_cast<double>(rotary_dim) * std::log2(static_cast<double>(base)));
  qa.eps = eps; qa.base = base; qa.factor = factor; qa.low = low; qa.high = high; qa.attention_factor = attention_factor;
  qa.positions = positions; qa.pos64 = positions_is_i64; qa.on_the_fly = cos_sin_cache == nullptr;
  qa.cache_vec = cos_sin_cache && ((uintptr_t)cos_sin_cache & 15) == 0 && cos_sin_stride % 4 == 0 && rotary_dim % 8 == 0;
  if (lay) {
    RX_REQUIRE(v && loc && lay->k_buf && lay->v_buf && v_head_dim > 0, "rx_qknorm_rope_store_kv: pool store needs v, loc and the layout's buffers");
    RX_REQUIRE(lay->page_size >= 1 && size_limit > 0, "rx_qknorm_rope_store_kv: bad page_size / size_limit");
    RX_REQUIRE(k_scale > 0.f && v_scale > 0.f, "rx_qknorm_rope_store_kv: scales must be > 0");
    a.k_buf = const_cast<void*>(lay->k_buf); a.v_buf = const_cast<void*>(lay->v_buf);
    a.page_size = lay->page_size; a.kv_fp8 = lay->kv_fp8;
    a.kps = lay->k_page_stride; a.kts = lay->k_tok_stride; a.khs = lay->k_head_stride;
    a.vps = lay->v_page_stride; a.vts = lay->v_tok_stride; a.vhs = lay->v_head_stride;
    a.loc = loc; a.loc64 = loc_is_i64; a.size_limit = size_limit; a.skip_index = skip_index;
    a.k_scale = k_scale; a.v_scale = v_scale; a.err_flag = err_flag;
  }
  const int64_t units = n * (num_q_heads + num_kv_heads);
  auto s = static_cast<hipStream_t>(stream);
  // the fast form: power-of-two head dims with contiguous, 16-byte aligned rows everywhere it loads or stores 16 bytes
  const bool pow2 = head_dim == 64 || head_dim == 128 || head_dim == 256 || head_dim == 512;
  bool fast = pow2 && (is_neox ? rotary_dim % 16 == 0 : true) &&
              ((q_stride_t | q_stride_h | k_stride_t | k_stride_h) % 8 == 0) &&
              (((uintptr_t)q | (uintptr_t)k | (uintptr_t)q_weight | (uintptr_t)k_weight) & 15) == 0;
  if (fast && lay) {
    fast = v_head_dim == head_dim && ((v_stride_t | v_stride_h) % 8 == 0) && ((uintptr_t)v & 15) == 0 &&
           (lay->kv_fp8 || (((lay->k_page_stride | lay->k_tok_stride | lay->k_head_stride | lay->v_page_stride |
                              lay->v_tok_stride | lay->v_head_stride) % 8 == 0) &&
                            (((uintptr_t)lay->k_buf | (uintptr_t)lay->v_buf) & 15) == 0));
  }
  if (fast) {
    const int lph = head_dim / 8, hpw = 64 / lph;
    const dim3 grid(static_cast<unsigned>((units + 4 * hpw - 1) / (4 * hpw))), block(256);  // 4 waves x UNR passes x hpw heads
#define RX_QKN(TT)                                                                            \
  do {                                                                                        \
    if (lph == 8) hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 8>), grid, block, 0, s, qa);        \
    else if (lph == 16) hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 16>), grid, block, 0, s, qa); \
    else if (lph == 32) hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 32>), grid, block, 0, s, qa); \
    else hipLaunchKernelGGL((qknorm_rope_fast_kernel<TT, 64>), grid, block, 0, s, qa);                \
  } while (0)
    if (dtype == RX_BF16) RX_QKN(BF16);
    else RX_QKN(F16);
#undef RX_QKN
    return check_launch("rx_qknorm_rope_store_kv");
  }
  const dim3 grid(static_cast<unsigned>((units + 3) / 4)), block(256);
  if (dtype == RX_BF16) hipLaunchKernelGGL(qknorm_rope_store_kernel<BF16>, grid, block, 0, s, qa);
  else hipLaunchKernelGGL(qknorm_rope_store_kernel<F16>, grid, block, 0, s, qa);
  return check_launch("rx_qknorm_rope_store_kv");
}
