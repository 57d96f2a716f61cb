// K7 for the head dims the 32x32x16 kernel (rx_extend32.hip, D = 128) does not serve: an MFMA 16x16x32 extend
// kernel templated on (Dk, Dv) -- 256/256 (Gemma-class heads; the reference retunes its Triton kernel for
// 128 < D <= 256 on gfx950, kernels/ops/attention/extend_attention.py:66-77), 192/128 (the MLA prefill shape:
// qk_nope 128 + rope 64 against v 128), 192/192, 96/96.  Same contract as rx_extend.hip (extend_attention_fwd,
// extend_attention.py:664-812; _fwd_kernel :241-661): causal / non-causal, skip_prefix / skip_extend, sliding
// window, logit cap, sinks, LSE, k/v scales.  Tree masks, the unified form, the Grok temperature and fp8 pools
// stay with the D = 128 kernel and the generic kernel.
//
// Why 16x16x32 here: at Dv = 256 a 32-query block's O^T accumulator alone is 128 registers per lane on the
// 32x32 shape; on 16x16 tiles a wave carries 16 (CB = 1) or 32 (CB = 2) queries with 64 accumulator registers and
// two waves per SIMD still fit.  One workgroup = 4 or 8 waves (NdGeom::NW) of 16 * CB queries of one (request,
// q head); K/V tiles of TT tokens are staged once per workgroup (global -> registers, in flight for a whole tile -> padded LDS rows,
// double buffered, one barrier per tile); S^T = K Q^T puts one query on lane & 15 with 8 scores of a 32-token
// half in the lane, so the softmax is lane-local up to one quad reduction, and packed P is the B operand of
// O^T += V^T P^T with V^T read by ds_read_b64_tr_b16.
#include "rx_common.h"

namespace rx {

struct ExtNdArgs {
  const uint16_t* q;
  const uint16_t* k_ext;
  const uint16_t* v_ext;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t, o_stride_h;
  const uint16_t* k_buf;
  const uint16_t* v_buf;
  int32_t page_shift;  // log2(page_size) of a power-of-two page, or -1: divide by page_size
  int32_t page_size;
  int64_t k_page_stride, k_tok_stride, k_head_stride;
  int64_t v_page_stride, v_tok_stride, v_head_stride;
  const void* qo_indptr;
  int32_t qo64;
  const int32_t* kv_indptr;
  const void* kv_indices;
  int32_t idx64;
  float* lse;
  int64_t lse_stride_t, lse_stride_h;
  int32_t bs, hq, hkv, group, mblocks;
  float sm_scale, k_scale, v_scale, logit_cap;
  int32_t causal, skip_prefix, skip_extend, window;
  const float* sinks;
};

template <int DK, int DV, bool BIG>
struct NdGeom {
#ifndef RX_ND_WIDE256
#define RX_ND_WIDE256 0
#endif
  static constexpr int CB = (DV > 128 && !RX_ND_WIDE256) ? 1 : 2;  // 16-query blocks per wave
#ifndef RX_ND_MINW_WIDE
#define RX_ND_MINW_WIDE 1
#endif
  static constexpr int MINW = (DV > 128 && RX_ND_WIDE256) ? RX_ND_MINW_WIDE : 2;  // waves per SIMD to allocate registers for
  // tokens per staged tile: 64, or 32 for the long rows of the four-wave form (two workgroups per CU keep their LDS)
  static constexpr int TT = (DK > 128 && !BIG) ? 32 : 64;
  static constexpr int KROW = DK * 2, VROW = DV * 2;   // bytes per row
  // padded LDS rows: K rows step an odd number of 16-B chunks (the 16 rows of one ds_read_b128 pass land on 16
  // different chunk positions), V rows step 64 B past a multiple of 256 (the 4 rows of a transposed read land on
  // 4 different 64-B bank groups)
  // (round 3: TWO pad chunks for both.  ds_read_b128 is served in four NON-contiguous 16-lane groups -- {0-3, 12-15,
  // 20-27}, ... (MI355X_MICROARCH.md, LDS) -- so one pad chunk leaves two rows of a group on one bank, and the old
  // V stride did the same to the 32-lane halves of the transposed reads; 32 B is conflict-free for both at every D)
  static constexpr int KSTRIDE = KROW + 32;
  static constexpr int VSTRIDE = VROW + 32;
  static constexpr int KTILE = TT * KSTRIDE, VTILE = TT * VSTRIDE, BUF = KTILE + VTILE;
  // waves per workgroup: a staged tile serves NW * QPW queries.  With four waves of 16 queries (Dv > 128) every tile
  // is re-staged for 64 queries only and the kernel is bound by L2 -> LDS staging and its one barrier per 32 tokens,
  // not by the matrix pipe: the BIG form (eight waves, 64-token tiles, one workgroup per CU) serves Dk > 128 whenever
  // the extends are long enough to fill it (D = 256: 448 -> 558 TFLOP/s, 192/128: 575 -> 715 at config 3's chunk)
  static constexpr int NW = BIG ? 8 : 4;
  static constexpr int NT = 64 * NW;
  static constexpr int QPW = 16 * CB, QPWG = NW * QPW;
};

__device__ __forceinline__ int64_t nd_slot_off(int64_t slot, int32_t shift, int32_t page_size, int64_t page_stride,
                                               int64_t tok_stride) {
  if (shift >= 0) return (slot >> shift) * page_stride + (slot & ((1 << shift) - 1)) * tok_stride;
  return (slot / page_size) * page_stride + (slot % page_size) * tok_stride;
}

// PLAIN: no sliding window and no logit cap in this instance (their scalars and branches cost the plain call several
// per cent even when both are off: rx_extend_d256.hip measured 8-9 %); the launcher picks it when the call uses neither
template <typename T, int DK, int DV, bool BIG, bool PLAIN>
__global__ __launch_bounds__((NdGeom<DK, DV, BIG>::NT), (NdGeom<DK, DV, BIG>::MINW * 4 / NdGeom<DK, DV, BIG>::NW)) void extend_nd_kernel(const ExtNdArgs a0) {
  ExtNdArgs a = a0;
  if constexpr (PLAIN) {
    a.window = 0;
    a.logit_cap = 0.f;
  }
  using G = NdGeom<DK, DV, BIG>;
  using vec8 = typename T::vec8;
  constexpr int CB = G::CB, TT = G::TT;
  constexpr int KS = DK / 32, NB = DV / 16;
  constexpr int CPRK = G::KROW / 16, CPRV = G::VROW / 16;          // 16-byte chunks per row
  constexpr int NCHK = TT * CPRK, NCHV = TT * CPRV;                // chunks per tile
  constexpr int NT = G::NT;
  constexpr int NPK = (NCHK + NT - 1) / NT, NPV = (NCHV + NT - 1) / NT;  // chunks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][K tile | V tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // kv head = block mod Hkv: a kv head's prefix rows stay in one XCD's L2 (blocks go to the XCDs round robin)
  int bid = blockIdx.x;
  const int kvh = bid % a.hkv;
  bid /= a.hkv;
  const int mb = a.mblocks - 1 - bid % a.mblocks;  // heaviest query blocks first under the causal mask
  bid /= a.mblocks;
  const int h = kvh * a.group + bid % a.group;
  const int req = bid / a.group;

  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t qb0 = mb * G::QPWG;
  if (qb0 >= E) return;  // workgroup-uniform
  const int32_t qbase = qb0 + w * G::QPW;
  const bool active = qbase < E;  // inactive waves still stage tiles and hit the barriers

  // ---- Q^T fragments: block c, lane (r, g) holds Q[qbase + 16 c + r][32 s + 8 g .. +8]
  vec8 qf[CB][KS];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int m = qbase + 16 * c + r;
    const bool ok = m < E;
    const uint16_t* qp = a.q + (qo0 + (ok ? m : 0)) * a.q_stride_t + h * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[c][s] = __builtin_bit_cast(vec8, raw);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // Q landed before the loop (rx_extend.hip explains why)

  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? min(E, qb0 + G::QPWG) : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? min(E, qbase + G::QPW) : E);
  const int nt1 = (p_len + TT - 1) / TT;
  const int nt2 = (n_end_wg + TT - 1) / TT;
  const int nt = nt1 + nt2;

  // ---- cooperative staging: chunk id = tid + 256 i -> (row = id / chunks-per-row, chunk = id % ...)
  const uint16_t* kbuf_h = a.k_buf + kvh * a.k_head_stride;
  const uint16_t* vbuf_h = a.v_buf + kvh * a.v_head_stride;
  const uint16_t* kext_h = a.k_ext + qo0 * a.k_stride_t + kvh * a.k_stride_h;
  const uint16_t* vext_h = a.v_ext + qo0 * a.v_stride_t + kvh * a.v_stride_h;
  int32_t slot_k[NPK], slot_v[NPV];
  auto tile_row = [&](int t, int row) -> int32_t {  // KV slot (prefix tile) or new-token row (extend tile)
    if (t < nt1) return static_cast<int32_t>(load_idx(a.kv_indices, kv0 + min(t * TT + row, p_len - 1), a.idx64));
    return min((t - nt1) * TT + row, n_end_wg - 1);
  };
  auto load_idx_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < NPK; ++i) slot_k[i] = tile_row(t, min((tid + NT * i) / CPRK, TT - 1));
#pragma unroll
    for (int i = 0; i < NPV; ++i) slot_v[i] = tile_row(t, min((tid + NT * i) / CPRV, TT - 1));
  };
  u32x4 stg_k[NPK], stg_v[NPV];
  auto issue_loads = [&](int t) {
    const bool pre = t < nt1;
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int id = tid + NT * i, ch = id % CPRK;
      if (NCHK % NT == 0 || id < NCHK) {
        const uint16_t* p = pre ? kbuf_h + nd_slot_off(slot_k[i], a.page_shift, a.page_size, a.k_page_stride, a.k_tok_stride)
                                : kext_h + static_cast<int64_t>(slot_k[i]) * a.k_stride_t;
        stg_k[i] = *reinterpret_cast<const u32x4*>(p + 8 * ch);
      }
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int id = tid + NT * i, ch = id % CPRV;
      if (NCHV % NT == 0 || id < NCHV) {
        const uint16_t* p = pre ? vbuf_h + nd_slot_off(slot_v[i], a.page_shift, a.page_size, a.v_page_stride, a.v_tok_stride)
                                : vext_h + static_cast<int64_t>(slot_v[i]) * a.v_stride_t;
        stg_v[i] = *reinterpret_cast<const u32x4*>(p + 8 * ch);
      }
    }
  };
  auto write_lds = [&](int buf) {
    char* kt = smem + buf * G::BUF;
    char* vt = kt + G::KTILE;
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
      const int id = tid + NT * i;
      if (NCHK % NT == 0 || id < NCHK)
        *reinterpret_cast<u32x4*>(kt + (id / CPRK) * G::KSTRIDE + (id % CPRK) * 16) = stg_k[i];
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int id = tid + NT * i;
      if (NCHV % NT == 0 || id < NCHV)
        *reinterpret_cast<u32x4*>(vt + (id / CPRV) * G::VSTRIDE + (id % CPRV) * 16) = stg_v[i];
    }
  };

  f32x4 oacc[CB][NB];
  float m_run[CB], l_run[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    m_run[c] = -INFINITY;
    l_run[c] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) oacc[c][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // sliding window: tiles wholly below the workgroup's first query's bound are never staged (rx_extend.hip)
  int t0 = 0;
  if (a.window > 0) {
    t0 = min(nt1, max(0, P + qb0 - a.window) / TT);
    if (t0 == nt1) t0 += min(nt2, max(0, qb0 - a.window) / TT);
  }
  if (nt > t0) {
    load_idx_tile(t0);
    issue_loads(t0);
    if (nt > t0 + 1) load_idx_tile(t0 + 1);
    write_lds(t0 & 1);
    if (nt > t0 + 1) {
      issue_loads(t0 + 1);
      if (nt > t0 + 2) load_idx_tile(t0 + 2);
    }
  }
  __syncthreads();

  const int qd = r >> 2, pp = r & 3;
  const int vrow0 = 4 * g + qd;  // V^T read: row inside a 16-token block
  const bool capped = a.logit_cap > 0.f;

  for (int t = t0; t < nt; ++t) {
    const char* kt = smem + (t & 1) * G::BUF;
    const char* vt = kt + G::KTILE;
    const bool prefix = t < nt1;
    const int tile_n0 = (prefix ? t : t - nt1) * TT;
    const int32_t lim = prefix ? p_len : n_end_w;
    const int32_t win_lo = a.window > 0 ? (prefix ? P : 0) + qbase - a.window : INT32_MIN;  // below it: hidden from the whole wave
    if (active) {
#pragma unroll
      for (int hh = 0; hh < TT / 32; ++hh) {
        const int n0 = tile_n0 + 32 * hh;  // first token of this 32-token half
        if (n0 >= lim || n0 + 32 <= win_lo) continue;  // nothing visible to this wave (wave-uniform)
        const float cs = prefix ? a.sm_scale * a.k_scale : a.sm_scale;
        // ---- S^T = K Q^T: tokens 16 bb + 4 g + i of the half on the lane, query r
        f32x4 sacc[CB][2];
#pragma unroll
        for (int c = 0; c < CB; ++c) sacc[c][0] = sacc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const char* krow = kt + (32 * hh + 16 * bb + r) * G::KSTRIDE + g * 16;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const vec8 ka = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(krow + s * 64));
#pragma unroll
            for (int c = 0; c < CB; ++c) sacc[c][bb] = T::mfma(ka, qf[c][s], sacc[c][bb]);
          }
        }
        // V^T fragments of the first d blocks go out before the softmax math (they depend on the staged tile only)
        const char* rp0 = vt + (32 * hh + vrow0) * G::VSTRIDE + 8 * (pp & 1) + (pp >> 1) * 16;
        const char* rp1 = rp0 + 16 * G::VSTRIDE;
        constexpr int NPRE = NB < 4 ? NB : 4;
        u32x2 vlo[NPRE], vhi[NPRE];
#pragma unroll
        for (int nb = 0; nb < NPRE; ++nb) {
          vlo[nb] = T::ds_read_tr(rp0 + nb * 32);
          vhi[nb] = T::ds_read_tr(rp1 + nb * 32);
        }
        bool full;  // every (query, token) pair of this half is visible: no mask code
        // (a window: also every row of the wave within W of the half's first token -- the interior of the band needs no mask)
        if (prefix) full = (n0 + 32 <= p_len) && (a.window <= 0 || P + qbase + G::QPW - 1 <= n0 + a.window);
        else full = (n0 + 32 <= E) && (!a.causal || n0 + 31 <= qbase) && (a.window <= 0 || qbase + G::QPW - 1 <= n0 + a.window);
        const float c2 = capped ? kLog2e : cs * kLog2e;
        vec8 pf[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          const int m = qbase + 16 * c + r;
          float sv[8];
          auto load_sv = [&]() {  // the block's 8 scores per lane as the softmax takes them: capped, masked
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[c][bb][i];
            if (capped) {
#pragma unroll
              for (int j = 0; j < 8; ++j) sv[j] = a.logit_cap * tanhf(sv[j] * cs / a.logit_cap);
            }
            if (!full) {
#pragma unroll
              for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int n = n0 + 16 * bb + 4 * g + i;
                  bool keep;
                  if (prefix) {
                    keep = n < p_len;
                    if (a.window > 0) keep = keep && (P + m <= n + a.window);
                  } else {
                    keep = n < n_end_w && (!a.causal || n <= m);
                    if (a.window > 0) keep = keep && (m <= n + a.window);
                  }
                  sv[bb * 4 + i] = keep ? sv[bb * 4 + i] : -INFINITY;
                }
            }
          };
          load_sv();
          // no row maximum on the common path (round 4; rx_extend.hip has the full comment): exponentials against the
          // standing running max, the lane's partial row sum as the check (compared as bits: NaN-proof under
          // -fno-honor-nans); the max step only when a lane's sum runs away
          float alpha = 1.0f, psum = 0.f;
          {
            const float m_old = m_run[c];
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              e[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_old));
              psum += e[j];
            }
            if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, 4096.0f)) != 0) {
              load_sv();
              float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
              mt = quad_row_max(mt) * c2;  // c2 > 0: max commutes with the scale
              const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;  // extend_attention.py:474-475
              const float m_new = fmaxf(m_old, mt_fixed);
              alpha = fast_exp2(m_old - m_new);
              m_run[c] = m_new;
              psum = 0.f;
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                sv[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_new));
                psum += sv[j];
              }
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) sv[j] = e[j];
            }
          }
          l_run[c] = l_run[c] * alpha + psum;
          if (prefix && a.v_scale != 1.0f) {  // per-tensor V scale of the cached part (wave-uniform branch)
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] *= a.v_scale;
          }
          u32x4 praw;
          praw[0] = pack2<T>(sv[0], sv[1]);
          praw[1] = pack2<T>(sv[2], sv[3]);
          praw[2] = pack2<T>(sv[4], sv[5]);
          praw[3] = pack2<T>(sv[6], sv[7]);
          pf[c] = __builtin_bit_cast(vec8, praw);
          if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) oacc[c][nb] *= alpha;
          }
        }
        // ---- O^T += V^T P^T
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          u32x2 lo, hi;
          if (nb < NPRE) {
            lo = vlo[nb];
            hi = vhi[nb];
          } else {
            lo = T::ds_read_tr(rp0 + nb * 32);
            hi = T::ds_read_tr(rp1 + nb * 32);
          }
          const vec8 av = __builtin_bit_cast(vec8, u32x4{lo[0], lo[1], hi[0], hi[1]});
#pragma unroll
          for (int c = 0; c < CB; ++c) oacc[c][nb] = T::mfma(av, pf[c], oacc[c][nb]);
        }
      }
    }
    // next tile: registers (loaded one tile ago) -> the other LDS buffer; its last readers finished a barrier ago
    if (t + 1 < nt) {
      write_lds((t + 1) & 1);
      if (t + 2 < nt) {
        issue_loads(t + 2);
        if (t + 3 < nt) load_idx_tile(t + 3);
      }
    }
    __syncthreads();
  }

  if (!active) return;
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    float l = l_run[c];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int m = qbase + 16 * c + r;
    if (m >= E) continue;
    float den = l;
    if (a.sinks) den += fast_exp2(a.sinks[h] * kLog2e - m_run[c]);
    const float inv = 1.0f / den;
    uint16_t* op = a.o + (qo0 + m) * a.o_stride_t + h * a.o_stride_h + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x2 pk;
      pk[0] = pack2<T>(oacc[c][nb][0] * inv, oacc[c][nb][1] * inv);
      pk[1] = pack2<T>(oacc[c][nb][2] * inv, oacc[c][nb][3] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
    }
    if (a.lse && g == 0) a.lse[(qo0 + m) * a.lse_stride_t + h * a.lse_stride_h] = m_run[c] * kLn2 + __logf(l);
  }
}

template <typename T, int DK, int DV, bool BIG, bool PLAIN>
static void launch_nd_one_(const ExtNdArgs& a0, int max_extend_len, hipStream_t s) {
  using G = NdGeom<DK, DV, BIG>;
  ExtNdArgs a = a0;
  a.mblocks = (max_extend_len + G::QPWG - 1) / G::QPWG;
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  constexpr unsigned kLds = 2 * G::BUF;
  auto kern = extend_nd_kernel<T, DK, DV, BIG, PLAIN>;
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  (void)attr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), kLds, s, a);
}

template <typename T, int DK, int DV, bool BIG>
static void launch_nd_one(const ExtNdArgs& a0, int max_extend_len, hipStream_t s) {
  if (a0.window > 0 || a0.logit_cap > 0.f) launch_nd_one_<T, DK, DV, BIG, false>(a0, max_extend_len, s);
  else launch_nd_one_<T, DK, DV, BIG, true>(a0, max_extend_len, s);
}

bool extend_nd_supports(int dk, int dv) {
  return (dk == 256 && dv == 256) || (dk == 192 && dv == 128) || (dk == 192 && dv == 192) || (dk == 96 && dv == 96);
}

// caller (rx_extend_attn) has validated pointers, alignment (16 B q / k / v / pools, 8 B o) and strides
int launch_extend_nd(const rx_extend_params* p, hipStream_t s) {
  ExtNdArgs a;
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
  a.page_shift = (a.page_size & (a.page_size - 1)) == 0 ? __builtin_ctz(a.page_size) : -1;
  a.k_page_stride = p->kv.k_page_stride; a.k_tok_stride = p->kv.k_tok_stride; a.k_head_stride = p->kv.k_head_stride;
  a.v_page_stride = p->kv.v_page_stride; a.v_tok_stride = p->kv.v_tok_stride; a.v_head_stride = p->kv.v_head_stride;
  a.qo_indptr = p->qo_indptr; a.qo64 = p->qo_indptr_is_i64;
  a.kv_indptr = p->kv_indptr; a.kv_indices = p->kv_indices; a.idx64 = p->kv_indices_is_i64;
  a.lse = p->lse; a.lse_stride_t = p->lse_stride_t; a.lse_stride_h = p->lse_stride_h;
  a.bs = p->bs; a.hq = p->num_q_heads; a.hkv = p->num_kv_heads;
  a.group = p->num_q_heads / p->num_kv_heads;
  a.mblocks = 1;
  a.sm_scale = p->sm_scale; a.k_scale = p->k_scale; a.v_scale = p->v_scale; a.logit_cap = p->logit_cap;
  a.causal = p->is_causal; a.skip_prefix = p->skip_prefix; a.skip_extend = p->skip_extend;
  a.window = p->sliding_window_size; a.sinks = p->sinks;
  const int dk = p->head_dim, dv = p->v_head_dim, mel = p->max_extend_len;
  const bool bf = p->dtype == RX_BF16;
  const bool no_big = !options().extend_nd_big;  // (A/B switch: the four-wave form only)
  // the eight-wave form when the longest extend fills more than a four-wave workgroup's query rows
#define RX_ND(DK_, DV_)                                                           \
  if (dk == DK_ && dv == DV_) {                                                   \
    const bool big = DK_ > 128 && mel > NdGeom<DK_, DV_, false>::QPWG && !no_big; \
    note_dispatch("extend_nd_kernel<%s, %d, %d, %s, %s>", bf ? "rx::BF16" : "rx::F16", DK_, DV_, tbool(big), \
                  tbool(!(a.window > 0 || a.logit_cap > 0.f)));                     \
    if (bf) {                                                                     \
      if (big) launch_nd_one<BF16, DK_, DV_, (DK_ > 128)>(a, mel, s);             \
      else launch_nd_one<BF16, DK_, DV_, false>(a, mel, s);                       \
    } else {                                                                      \
      if (big) launch_nd_one<F16, DK_, DV_, (DK_ > 128)>(a, mel, s);              \
      else launch_nd_one<F16, DK_, DV_, false>(a, mel, s);                        \
    }                                                                             \
    return RX_OK;                                                                 \
  }
  RX_ND(256, 256)
  RX_ND(192, 128)
  RX_ND(192, 192)
  RX_ND(96, 96)
#undef RX_ND
  return fail(RX_ERR_UNSUPPORTED, "launch_extend_nd: head dims %d / %d", dk, dv);
}

}  // namespace rx
