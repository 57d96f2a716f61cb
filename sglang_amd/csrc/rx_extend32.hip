// K7, head_dim 128: extend attention on v_mfma_f32_32x32x16 (the MFMA-bound fast path).
//
// Same contract as rx_extend.hip (extend_attention_fwd, kernels/ops/attention/extend_attention.py:
// 664-812; _fwd_kernel :241-661).  Why a second kernel: at D=128 the 16x16x32 formulation is
// ISSUE-bound (measured: 5.5 VALU per MFMA, MFMA 25 % busy).  A 32x32x16 MFMA carries twice the
// FLOPs per issue slot, puts ONE query on lane&31 with 16 of its scores per 32-token block in the
// lane's registers (row max = 31 local max + ONE half swap), and halves the per-FLOP count of LDS
// fragment instructions.
//
// Workgroup = NW waves (8 -> 256 queries, 4 -> 128) of one (request, q head); wave = 32 queries.
// KV tiles of 64 tokens are staged once per workgroup (global -> registers one tile ahead -> swizzled
// LDS, double buffered, one barrier per tile):
//   S^T[tok][q] = K Q^T : A = K fragment (lane = (token&31, 8-wide d half)) by ds_read_b128,
//                         B = Q^T kept in 32 VGPRs.
//   softmax on the lane (exp2(fma(s, c, -m)), exact lazy rescale), P packed to bf16 IS the B operand
//   of the next product (registers 8s..8s+7 of an S block = k-step s, cdna_hip_programming.md §3).
//   O^T[d][q] += V^T P^T : A = V^T fragment by two ds_read_b64_tr_b16 per k-step in the matching
//                         permuted token order (16s + 8(j>>2) + 4h + (j&3)).
// LDS image: 256-B rows, 16-byte chunk index XOR ((row&3)<<2 | (row>>2)&3): conflict-free for the
// staging ds_write_b128, the K ds_read_b128 and the V transposed reads.
#include "rx_common.h"

namespace rx {

struct Ext32Args {
  const uint16_t* q;
  const uint16_t* k_ext;
  const uint16_t* v_ext;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t, o_stride_h;
  const uint16_t* k_buf;
  const uint16_t* v_buf;
  int32_t page_size;
  int64_t k_page_stride, k_tok_stride, k_head_stride;
  int64_t v_page_stride, v_tok_stride, v_head_stride;
  const void* qo_indptr;
  int32_t qo64;
  const int32_t* kv_indptr;
  const void* kv_indices;
  float* lse;
  int64_t lse_stride_t, lse_stride_h;
  int32_t bs, hq, hkv, group, mblocks;
  float sm_scale, k_scale, v_scale, logit_cap;
  int32_t causal, skip_prefix, skip_extend, window;
  const float* sinks;
};

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kD = 128, kRow = 256, kTok = 64;  // head dim, bytes per row, tokens per tile

__device__ __forceinline__ int swz32(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <bool LINEAR>
__device__ __forceinline__ int64_t slot_off32(int64_t slot, int32_t page_size, int64_t page_stride,
                                              int64_t tok_stride) {
  if constexpr (LINEAR) return mul_u32(slot, tok_stride);
  if (page_size < 0) {
    const int sh = -page_size - 1;
    return mul_u32(slot >> sh, page_stride) + mul_u32(slot & ((1 << sh) - 1), tok_stride);
  }
  return (slot / page_size) * page_stride + (slot % page_size) * tok_stride;
}

template <typename T>
__device__ __forceinline__ f32x16 mfma32(typename T::vec8 a, typename T::vec8 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma32<BF16>(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<F16>(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// max over lanes l and l^32 (one query's two register halves)
__device__ __forceinline__ float half_swap_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1"
               : "+v"(a), "+v"(b));
  return a;
}

template <typename T, typename IdxT, bool LINEAR, bool VSCALE, int NW>
__global__ __launch_bounds__(64 * NW, 2) void extend_mfma32_kernel(const Ext32Args a) {
  using vec8 = typename T::vec8;
  constexpr int KS = kD / 16;                  // 8 k-steps of the QK^T product
  constexpr int DB = kD / 32;                  // 4 output d blocks of 32
  constexpr int THREADS = 64 * NW;
  constexpr int RPP = THREADS / 16;            // rows staged per pass
  constexpr int NPASS = kTok / RPP;            // 2 (NW=8) or 4 (NW=4)
  constexpr int TILE_BYTES = kTok * kRow;      // 16 KiB
  constexpr int QPW = 32;                      // queries per wave
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][K|V]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;

  int bid = blockIdx.x;
  const int mb = bid % a.mblocks;
  bid /= a.mblocks;
  const int head = bid % a.hq;
  const int req = bid / a.hq;
  const int kvh = head / a.group;

  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t qb0 = mb * NW * QPW;
  if (qb0 >= E) return;
  const int32_t qbase = qb0 + w * QPW;
  const bool active = qbase < E;
  const IdxT* idx = reinterpret_cast<const IdxT*>(a.kv_indices) + kv0;
  const int m = qbase + ql;  // this lane's query (index inside the extend part)

  // ---- Q^T fragments: lane (q, h) holds Q[q][16 ks + 8 h .. +8] ------------------------------------
  vec8 qf[KS];
  {
    const bool ok = m < E;
    const uint16_t* qp = a.q + (qo0 + (ok ? m : 0)) * a.q_stride_t + head * a.q_stride_h + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 16 * ks) : u32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(vec8, raw);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // Q landed before the loop (see rx_extend.hip)

  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? min(E, qb0 + NW * QPW) : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? min(E, qbase + QPW) : E);
  const int nt1 = (p_len + kTok - 1) / kTok;
  const int nt2 = (n_end_wg + kTok - 1) / kTok;
  const int nt = nt1 + nt2;

  // ---- cooperative staging ------------------------------------------------------------------------
  const int st_row = tid >> 4, st_chunk = tid & 15;
  const uint16_t* kbuf_h = a.k_buf + kvh * a.k_head_stride + 8 * st_chunk;
  const uint16_t* vbuf_h = a.v_buf + kvh * a.v_head_stride + 8 * st_chunk;
  const uint16_t* kext_h = a.k_ext + qo0 * a.k_stride_t + kvh * a.k_stride_h + 8 * st_chunk;
  const uint16_t* vext_h = a.v_ext + qo0 * a.v_stride_t + kvh * a.v_stride_h + 8 * st_chunk;
  int32_t slot[NPASS];
  auto load_idx_tile = [&](int t) {
    if (t < nt1) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i)
        slot[i] = static_cast<int32_t>(idx[min(t * kTok + i * RPP + st_row, p_len - 1)]);
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) slot[i] = min((t - nt1) * kTok + i * RPP + st_row, n_end_wg - 1);
    }
  };
  u32x4 stg_k[NPASS], stg_v[NPASS];
  auto issue_loads = [&](int t) {
    if (t < nt1) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        stg_k[i] = *reinterpret_cast<const u32x4*>(
            kbuf_h + slot_off32<LINEAR>(slot[i], a.page_size, a.k_page_stride, a.k_tok_stride));
        stg_v[i] = *reinterpret_cast<const u32x4*>(
            vbuf_h + slot_off32<LINEAR>(slot[i], a.page_size, a.v_page_stride, a.v_tok_stride));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        stg_k[i] = *reinterpret_cast<const u32x4*>(kext_h + mul_u32(slot[i], a.k_stride_t));
        stg_v[i] = *reinterpret_cast<const u32x4*>(vext_h + mul_u32(slot[i], a.v_stride_t));
      }
    }
  };
  auto write_lds = [&](int buf) {
    char* kt = smem + buf * 2 * TILE_BYTES;
    char* vt = kt + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = i * RPP + st_row;
      const int off = row * kRow + ((st_chunk ^ swz32(row)) << 4);
      *reinterpret_cast<u32x4*>(kt + off) = stg_k[i];
      *reinterpret_cast<u32x4*>(vt + off) = stg_v[i];
    }
  };

  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  if (nt > 0) {
    load_idx_tile(0);
    issue_loads(0);
    if (nt > 1) load_idx_tile(1);
    write_lds(0);
    if (nt > 1) {
      issue_loads(1);
      if (nt > 2) load_idx_tile(2);
    }
  }
  __syncthreads();

  // per-lane LDS addressing constants
  const int kswz = swz32(ql);                       // K rows 32 b + ql
  const int tq = lane & 15, qd = tq >> 2, pp = tq & 3, dg = (lane >> 4) & 1;
  // V^T reads: rows 16 s + 4 h + qd (+8); chunk = 4 db + 2 dg + (pp >> 1); swizzle uses row & 15
  const int vrow_lo = 4 * h + qd, vrow_hi = 8 + 4 * h + qd;
  const int vsw_lo = swz32(vrow_lo), vsw_hi = swz32(vrow_hi);
  const int vcol = 2 * dg + (pp >> 1), vbyte = 8 * (pp & 1);
  const bool capped = a.logit_cap > 0.f;

  for (int t = 0; t < nt; ++t) {
    const char* kt = smem + (t & 1) * 2 * TILE_BYTES;
    const char* vt = kt + TILE_BYTES;
    const bool prefix = t < nt1;
    const int tile_n0 = (prefix ? t : t - nt1) * kTok;
    const int32_t lim = prefix ? p_len : n_end_w;
    if (active && tile_n0 < lim) {
      const int nblk = (tile_n0 + 32 < lim) ? 2 : 1;  // visible 32-token blocks of this tile
      const float cs = prefix ? a.sm_scale * a.k_scale : a.sm_scale;
      const float c2 = capped ? kLog2e : cs * kLog2e;
      // ---- S^T blocks -----------------------------------------------------------------------------
      f32x16 sacc[2];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b < nblk) {
          const char* krow = kt + (32 * b + ql) * kRow;
          // K fragments are fetched two k-steps ahead of the MFMA that consumes them: a ds_read_b128
          // round trip is ~100+ cycles, an MFMA 32; without the explicit distance hipcc issues each
          // read right in front of its MFMA and the matrix pipe idles on lgkmcnt(0)
          u32x4 kfr[KS];
          kfr[0] = *reinterpret_cast<const u32x4*>(krow + (((0 + h) ^ kswz) << 4));
          kfr[1] = *reinterpret_cast<const u32x4*>(krow + (((2 + h) ^ kswz) << 4));
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks + 2 < KS)
              kfr[ks + 2] = *reinterpret_cast<const u32x4*>(krow + (((2 * (ks + 2) + h) ^ kswz) << 4));
            // the first MFMA takes the literal zero as C: no 16-register clear per block
            sacc[b] = mfma32<T>(__builtin_bit_cast(vec8, kfr[ks]), qf[ks], ks == 0 ? zero16 : sacc[b]);
          }
          // pin the interleave (2 reads up front, then read / MFMA alternating) so the distance survives
          // instruction scheduling: masks 0x100 = DS read, 0x008 = MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks + 2 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
      }
      // ---- masks + online softmax: lane = one query, 16 tokens per block ------------------------------
      bool full;
      if (prefix) full = (tile_n0 + 32 * nblk <= p_len) && a.window <= 0;
      else full = (tile_n0 + 32 * nblk <= E) && (!a.causal || tile_n0 + 32 * nblk - 1 <= qbase) && a.window <= 0;
      float mt = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b < nblk) {
          if (capped) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[b][i] = a.logit_cap * tanhf(sacc[b][i] * cs / a.logit_cap);
          }
          if (!full) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int n = tile_n0 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * h;
              bool keep;
              if (prefix) {
                keep = n < p_len;
                if (a.window > 0) keep = keep && (P + m <= n + a.window);
              } else {
                keep = n < n_end_w && (!a.causal || n <= m);
                if (a.window > 0) keep = keep && (m <= n + a.window);
              }
              sacc[b][i] = keep ? sacc[b][i] : -INFINITY;
            }
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) mt = fmaxf(mt, sacc[b][i]);
        }
      }
      mt = half_swap_max(mt);
      mt *= c2;
      const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;  // extend_attention.py:474-475
      const float m_new = fmaxf(m_run, mt_fixed);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
      u32x4 pk[2][2];  // [block][k-step within block]: 8 bf16 = registers 8s..8s+7
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b < nblk) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            sacc[b][i] = fast_exp2(__builtin_fmaf(sacc[b][i], c2, -m_new));
            psum += sacc[b][i];
          }
          if constexpr (VSCALE) {
            const float vs = prefix ? a.v_scale : 1.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[b][i] *= vs;
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) pk[b][s][j] = pack2<T>(sacc[b][8 * s + 2 * j], sacc[b][8 * s + 2 * j + 1]);
        }
      }
      l_run = l_run * alpha + psum;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[db] *= alpha;
      }
      // ---- O^T += V^T P^T: the V^T fragments of k-step i+1 are in flight under the MFMAs of step i ----
      {
        const int nsteps = 2 * nblk;  // k-steps of 16 tokens
        u32x4 vfr[2][DB];
        auto load_v = [&](int step, u32x4 (&dst)[DB]) {
          const int r0 = 16 * step;
          const char* rlo = vt + (r0 + vrow_lo) * kRow + vbyte;
          const char* rhi = vt + (r0 + vrow_hi) * kRow + vbyte;
#pragma unroll
          for (int db = 0; db < DB; ++db) {
            const u32x2 lo2 = T::ds_read_tr(rlo + (((4 * db + vcol) ^ vsw_lo) << 4));
            const u32x2 hi2 = T::ds_read_tr(rhi + (((4 * db + vcol) ^ vsw_hi) << 4));
            dst[db] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
          }
        };
        load_v(0, vfr[0]);
#pragma unroll
        for (int step = 0; step < 4; ++step) {
          if (step < nsteps) {
            if (step + 1 < nsteps) load_v(step + 1, vfr[(step + 1) & 1]);
            const vec8 pb = __builtin_bit_cast(vec8, pk[step >> 1][step & 1]);
#pragma unroll
            for (int db = 0; db < DB; ++db)
              oacc[db] = mfma32<T>(__builtin_bit_cast(vec8, vfr[step & 1][db]), pb, oacc[db]);
          }
        }
      }
    }
    if (t + 1 < nt) {
      write_lds((t + 1) & 1);
      if (t + 2 < nt) {
        issue_loads(t + 2);
        if (t + 3 < nt) load_idx_tile(t + 3);
      }
    }
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  if (!active) return;
  float l = l_run;
  {
    float a2 = l, b2 = l;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(a2), "+v"(b2));
    l = a2;
  }
  if (m >= E) return;
  float den = l;
  if (a.sinks) den += fast_exp2(a.sinks[head] * kLog2e - m_run);
  const float inv = 1.0f / den;
  uint16_t* op = a.o + (qo0 + m) * a.o_stride_t + head * a.o_stride_h + 4 * h;
#pragma unroll
  for (int db = 0; db < DB; ++db) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {  // registers 4 gq .. 4 gq + 3 = d 32 db + 8 gq + 4 h + 0..3
      u32x2 pk2;
      pk2[0] = pack2<T>(oacc[db][4 * gq] * inv, oacc[db][4 * gq + 1] * inv);
      pk2[1] = pack2<T>(oacc[db][4 * gq + 2] * inv, oacc[db][4 * gq + 3] * inv);
      *reinterpret_cast<u32x2*>(op + 32 * db + 8 * gq) = pk2;
    }
  }
  if (a.lse && h == 0) a.lse[(qo0 + m) * a.lse_stride_t + head * a.lse_stride_h] = m_run * kLn2 + __logf(l);
}

// launcher used by rx_extend.hip for head_dim == v_head_dim == 128
template <int NW>
static void launch32_nw(const Ext32Args& a, bool bf16, bool idx64, bool linear, bool vs, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
#define RX_E32(TT, IT, LIN, VS) \
  hipLaunchKernelGGL((extend_mfma32_kernel<TT, IT, LIN, VS, NW>), dim3(grid), dim3(64 * NW), 0, s, a)
#define RX_E32_VS(TT, IT, LIN) \
  do { if (vs) RX_E32(TT, IT, LIN, true); else RX_E32(TT, IT, LIN, false); } while (0)
#define RX_E32_LIN(TT, IT) \
  do { if (linear) RX_E32_VS(TT, IT, true); else RX_E32_VS(TT, IT, false); } while (0)
#define RX_E32_IDX(TT) \
  do { if (idx64) RX_E32_LIN(TT, int64_t); else RX_E32_LIN(TT, int32_t); } while (0)
  if (bf16) RX_E32_IDX(BF16);
  else RX_E32_IDX(F16);
#undef RX_E32_IDX
#undef RX_E32_LIN
#undef RX_E32_VS
#undef RX_E32
}

#ifndef RX_EXT32_NW
#define RX_EXT32_NW 8
#endif

int launch_extend32(const rx_extend_params* p, hipStream_t s) {
  Ext32Args a;
  a.q = (const uint16_t*)p->q;
  a.k_ext = (const uint16_t*)p->k_extend;
  a.v_ext = (const uint16_t*)p->v_extend;
  a.o = (uint16_t*)p->o;
  a.q_stride_t = p->q_stride_t; a.q_stride_h = p->q_stride_h;
  a.k_stride_t = p->k_stride_t; a.k_stride_h = p->k_stride_h;
  a.v_stride_t = p->v_stride_t; a.v_stride_h = p->v_stride_h;
  a.o_stride_t = p->o_stride_t; a.o_stride_h = p->o_stride_h;
  a.k_buf = (const uint16_t*)p->kv.k_buf;
  a.v_buf = (const uint16_t*)p->kv.v_buf;
  a.page_size = p->kv.page_size;
  if ((a.page_size & (a.page_size - 1)) == 0) a.page_size = -(__builtin_ctz(a.page_size) + 1);
  a.k_page_stride = p->kv.k_page_stride; a.k_tok_stride = p->kv.k_tok_stride; a.k_head_stride = p->kv.k_head_stride;
  a.v_page_stride = p->kv.v_page_stride; a.v_tok_stride = p->kv.v_tok_stride; a.v_head_stride = p->kv.v_head_stride;
  a.qo_indptr = p->qo_indptr; a.qo64 = p->qo_indptr_is_i64;
  a.kv_indptr = p->kv_indptr; a.kv_indices = p->kv_indices;
  a.lse = p->lse; a.lse_stride_t = p->lse_stride_t; a.lse_stride_h = p->lse_stride_h;
  a.bs = p->bs; a.hq = p->num_q_heads; a.hkv = p->num_kv_heads;
  a.group = p->num_q_heads / p->num_kv_heads;
  a.sm_scale = p->sm_scale; a.k_scale = p->k_scale; a.v_scale = p->v_scale; a.logit_cap = p->logit_cap;
  a.causal = p->is_causal; a.skip_prefix = p->skip_prefix; a.skip_extend = p->skip_extend;
  a.window = p->sliding_window_size; a.sinks = p->sinks;
  const bool linear = p->kv.page_size == 1 ||
                      (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                       p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  constexpr int NW = RX_EXT32_NW;
  a.mblocks = (p->max_extend_len + NW * 32 - 1) / (NW * 32);
  launch32_nw<NW>(a, p->dtype == RX_BF16, p->kv_indices_is_i64 != 0, linear, p->v_scale != 1.0f, s);
  return RX_OK;
}

}  // namespace rx
