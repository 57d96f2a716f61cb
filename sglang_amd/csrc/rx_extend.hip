// K7: ragged extend (prefill-with-cached-prefix) attention for gfx950.
//
// Reference: extend_attention_fwd (kernels/ops/attention/extend_attention.py:664-812),
// kernel _fwd_kernel (:241-661): stage 1 over the cached prefix gathered through kv_indices
// (:372-510), stage 2 over the new, contiguous K/V with the causal triangle (:512-631).
//
// MI355X design: the same transposed formulation as the decode kernel.  A wave owns 32 query
// rows of one (request, q head) -- two 16-query N blocks of v_mfma_f32_16x16x32 -- and streams
// 32-token KV tiles: S^T = K Q^T (K straight from HBM/L2 in A-operand shape), softmax on the
// lane, O^T += V^T P^T with V^T read by ds_read_b64_tr_b16 from a wave-private swizzled LDS
// tile.  The four waves of a workgroup take four consecutive query blocks of the same
// (request, head): they walk the same KV rows at the same time, so three of the four K/V reads
// are served by the CU's L1.  Prefix and extend stages share one tile body; the next tile's
// K/V (and the slot indices of the tile after it) are in flight under the current tile's math.
#include <cstdlib>

#include "rx_common.h"

namespace rx {

struct ExtendArgs {
  const uint16_t* q;
  const uint16_t* k_ext;
  const uint16_t* v_ext;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t,
      o_stride_h;
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
  const uint8_t* custom_mask;  // generic kernel only (the D = 128 MFMA kernel has its own copy)
  const int64_t* mask_indptr;
  const int32_t* window_kv_offsets;
  int32_t skip_prefix_mask, xai_len;
  const int32_t* unified_prefix;  // K8 unified form (see radix_hip.h), or null
  const float* sinks;
  const void* bias;  // relative-position score bias [T, Hq, bias_len] (generic kernel; the D = 128 kernel has its own copy), or null
  int32_t bias_f32, bias_len;
  int64_t bias_stride_t, bias_stride_h;
};

#ifndef RX_EXT_CB
#define RX_EXT_CB 2
#endif
constexpr int kQPerWaveDefault = 16 * RX_EXT_CB;

template <int D>
__device__ __forceinline__ int v_swz(int row) {
  if constexpr (D == 64) return row & 7;
  return ((row & 7) << 1) | ((row >> 2) & 1);
}

template <bool LINEAR>
__device__ __forceinline__ int64_t slot_off(int64_t slot, int32_t page_size, int64_t page_stride,
                                            int64_t tok_stride) {
  if constexpr (LINEAR) return mul_u32(slot, tok_stride);
  if (page_size < 0) {  // power-of-two page: -(log2(page) + 1)
    const int sh = -page_size - 1;
    return mul_u32(slot >> sh, page_stride) + mul_u32(slot & ((1 << sh) - 1), tok_stride);
  }
  return (slot / page_size) * page_stride + (slot % page_size) * tok_stride;
}

// One 256-thread workgroup = 128 query rows of one (request, q head): wave w owns rows
// [32w, 32w+32) as two 16-query N blocks.  KV tiles of 64 tokens are staged ONCE per workgroup:
// global -> registers (full 256-B rows, 16 lanes per row, issued one tile ahead) -> XOR-swizzled
// LDS image (double buffered, one barrier per tile).  Each wave then reads its K A-fragments with
// ds_read_b128 and its V^T fragments with ds_read_b64_tr_b16 from the same image layout.
#ifndef RX_EXT_TT
#define RX_EXT_TT 64
#endif
constexpr int kTT = RX_EXT_TT;  // tokens per LDS tile
#ifndef RX_EXT_MAX_SLACK
#define RX_EXT_MAX_SLACK 8.0f  // log2 units (0: the plain running max)
#endif
constexpr float kExtMaxSlack = RX_EXT_MAX_SLACK;
constexpr float kExtSumLimit = 4096.0f;  // a lane's partial row sum above this sends the wave to the max-based step

#ifndef RX_EXT_CB
#define RX_EXT_CB 2  // 16-query N blocks per wave (2 -> 32 queries / wave, 128 / workgroup)
#endif
#ifndef RX_EXT_TT
#define RX_EXT_TT 64  // tokens per staged LDS tile
#endif
#ifndef RX_EXT_MINW
#define RX_EXT_MINW 2
#endif

// PLAIN: no sliding window and no logit cap in this instance (their scalars and branches cost a plain call several per
// cent even when both are off); picked by the launcher
// CB: 16-query blocks per wave.  4 (64 queries per wave, 256 per workgroup) for long D = 64 extends: a staged tile and every
// K / V^T fragment read then feed twice the MFMAs, and at 256 FLOP per score this kernel's time is its instruction count
// (DESIGN 4.2, round 3); the 64 accumulator registers of D = 64 leave room for it.
template <typename T, int D, typename IdxT, bool LINEAR, bool VSCALE, bool PLAIN = false, int CB = RX_EXT_CB>
__global__ __launch_bounds__(256, RX_EXT_MINW) void extend_mfma_kernel(const ExtendArgs a0) {
  constexpr int kCB = CB, kQPerWave = 16 * CB;
  ExtendArgs a = a0;
  if constexpr (PLAIN) {
    a.window = 0;
    a.logit_cap = 0.f;
  }
  using vec8 = typename T::vec8;
  constexpr int KS = D / 32;
  constexpr int NB = D / 16;
  constexpr int ROW_BYTES = D * 2;
  constexpr int CPR = ROW_BYTES / 16;      // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;           // rows staged per pass of the 256 threads
  constexpr int NPASS = kTT / RPP;         // passes per tile (4 at D=128, 2 at D=64)
  constexpr int TILE_BYTES = kTT * ROW_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][K|V]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  int bid = blockIdx.x;
  const int mb = bid % a.mblocks;
  bid /= a.mblocks;
  const int h = bid % a.hq;
  const int req = bid / a.hq;
  const int kvh = h / a.group;

  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;  // prefix length
  const int32_t qb0 = mb * 4 * kQPerWave;
  if (qb0 >= E) return;  // workgroup-uniform
  const int32_t qbase = qb0 + w * kQPerWave;
  const bool active = qbase < E;  // inactive waves still stage tiles and hit the barriers
  const IdxT* idx = reinterpret_cast<const IdxT*>(a.kv_indices) + kv0;

  // ---- Q^T fragments: block c, lane (r,g) holds Q[qbase+16c+r][h][32s+8g..] -------------------
  vec8 qf[kCB][KS];
#pragma unroll
  for (int c = 0; c < kCB; ++c) {
    const int m = qbase + 16 * c + r;
    const bool ok = m < E;
    const uint16_t* qp = a.q + (qo0 + (ok ? m : 0)) * a.q_stride_t + h * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[c][s] = __builtin_bit_cast(vec8, raw);
    }
  }

  // The Q fragments must have LANDED before the tile loop: hipcc's waitcnt pass merges the loop-entry
  // state (Q loads possibly pending) into the loop header and would otherwise emit vmcnt(0) in front
  // of the first MFMA of EVERY iteration, i.e. wait for the next tile's prefetch before computing.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt/expcnt untouched
  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? min(E, qb0 + 4 * kQPerWave) : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? min(E, qbase + kQPerWave) : E);
  const int nt1 = (p_len + kTT - 1) / kTT;
  const int nt2 = (n_end_wg + kTT - 1) / kTT;
  const int nt = nt1 + nt2;

  // ---- cooperative staging: thread -> (row = pass*RPP + tid/CPR, chunk = tid%CPR) ---------------
  const int st_row = tid / CPR, st_chunk = tid % CPR;
  const uint16_t* kbuf_h = a.k_buf + kvh * a.k_head_stride + 8 * st_chunk;
  const uint16_t* vbuf_h = a.v_buf + kvh * a.v_head_stride + 8 * st_chunk;
  const uint16_t* kext_h = a.k_ext + qo0 * a.k_stride_t + kvh * a.k_stride_h + 8 * st_chunk;
  const uint16_t* vext_h = a.v_ext + qo0 * a.v_stride_t + kvh * a.v_stride_h + 8 * st_chunk;

  int64_t slot[NPASS];  // prefix tiles: KV slot; extend tiles: row index inside the extend part
  auto load_idx_tile = [&](int t) {
    if (t < nt1) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i)
        slot[i] = static_cast<int64_t>(idx[min(t * kTT + i * RPP + st_row, p_len - 1)]);
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i)
        slot[i] = min((t - nt1) * kTT + i * RPP + st_row, n_end_wg - 1);
    }
  };
  u32x4 stg_k[NPASS], stg_v[NPASS];
  auto issue_loads = [&](int t) {
    if (t < nt1) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        stg_k[i] = *reinterpret_cast<const u32x4*>(
            kbuf_h + slot_off<LINEAR>(slot[i], a.page_size, a.k_page_stride, a.k_tok_stride));
        stg_v[i] = *reinterpret_cast<const u32x4*>(
            vbuf_h + slot_off<LINEAR>(slot[i], a.page_size, a.v_page_stride, a.v_tok_stride));
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
      const int off = row * ROW_BYTES + ((st_chunk ^ v_swz<D>(row)) & (CPR - 1)) * 16;
      *reinterpret_cast<u32x4*>(kt + off) = stg_k[i];
      *reinterpret_cast<u32x4*>(vt + off) = stg_v[i];
    }
  };

  f32x4 oacc[kCB][NB];
#pragma unroll
  for (int c = 0; c < kCB; ++c)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) oacc[c][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[kCB], l_run[kCB];  // running max (log2 domain) / this lane's partial row sum
#pragma unroll
  for (int c = 0; c < kCB; ++c) {
    m_run[c] = -INFINITY;
    l_run[c] = 0.f;
  }

  // sliding window: tiles wholly below the workgroup's first query's bound (cached token n is visible to query m iff
  // P + m <= n + W, new token n iff m <= n + W) are never staged -- a long chunk under a short window (gpt-oss: W = 128)
  // otherwise computes and masks its whole causal triangle
  int t0 = 0;
  if (a.window > 0) {
    t0 = min(nt1, max(0, P + qb0 - a.window) / kTT);
    if (t0 == nt1) t0 += min(nt2, max(0, qb0 - a.window) / kTT);
  }
  // ---- prologue -----------------------------------------------------------------------------------
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

  const int swr = v_swz<D>(r);  // swizzle of rows 16k + r
  const int qd = r >> 2, pp = r & 3;
  const int vrow0 = 4 * g + qd;  // V^T read: row inside a 16-token block
  const int swv = v_swz<D>(vrow0);

  for (int t = t0; t < nt; ++t) {
    const char* kt = smem + (t & 1) * 2 * TILE_BYTES;
    const char* vt = kt + TILE_BYTES;
    const bool prefix = t < nt1;
    const int tile_n0 = (prefix ? t : t - nt1) * kTT;
    const int32_t lim = prefix ? p_len : n_end_w;
    const int32_t win_lo = a.window > 0 ? (prefix ? P : 0) + qbase - a.window : INT32_MIN;  // below it: hidden from the whole wave
    if (active) {
#pragma unroll
      for (int hh = 0; hh < kTT / 32; ++hh) {
        const int n0 = tile_n0 + 32 * hh;  // first token of this 32-token half
        if (n0 >= lim || n0 + 32 <= win_lo) continue;  // nothing visible to this wave (wave-uniform)
        const float cs = (prefix ? a.sm_scale * a.k_scale : a.sm_scale);
        // ---- S^T = K Q^T ---------------------------------------------------------------------------
        f32x4 sacc[kCB][2];
#pragma unroll
        for (int c = 0; c < kCB; ++c)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) sacc[c][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const char* krow = kt + (32 * hh + 16 * bb + r) * ROW_BYTES;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const int chunk = ((4 * s + g) ^ swr) & (CPR - 1);
            const vec8 ka = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(krow + chunk * 16));
#pragma unroll
            for (int c = 0; c < kCB; ++c) sacc[c][bb] = T::mfma(ka, qf[c][s], sacc[c][bb]);
          }
        }
        // ---- V^T fragments of the first half of the d blocks: issued BEFORE the softmax math so the
        // LDS round trip hides under it (they depend on the staged tile only, not on P) ------------------
        const char* rp0 = vt + (32 * hh + vrow0) * ROW_BYTES + 8 * (pp & 1);
        const char* rp1 = rp0 + 16 * ROW_BYTES;
        u32x2 vlo[NB], vhi[NB];
#pragma unroll
        for (int nb = 0; nb < NB / 2; ++nb) {
          const int chunk = ((2 * nb + (pp >> 1)) ^ swv) & (CPR - 1);
          vlo[nb] = T::ds_read_tr(rp0 + chunk * 16);
          vhi[nb] = T::ds_read_tr(rp1 + chunk * 16);
        }
        // ---- masks + online softmax (lane = one query per block c, 8 tokens) -------------------------
        bool full;  // every (query, token) pair of this half is visible: skip the mask code
        // (a window: also every row of the wave within W of the half's first token -- the interior of the band needs no mask)
        if (prefix) full = (n0 + 32 <= p_len) && (a.window <= 0 || P + qbase + kQPerWave - 1 <= n0 + a.window);
        else full = (n0 + 32 <= E) && (!a.causal || n0 + 31 <= qbase) && (a.window <= 0 || qbase + kQPerWave - 1 <= n0 + a.window);
        const bool capped = a.logit_cap > 0.f;
        const float c2 = capped ? kLog2e : cs * kLog2e;
        vec8 pf[kCB];
#pragma unroll
        for (int c = 0; c < kCB; ++c) {
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
          // Round 4 (rx_extend32_kernel.inc / rx_extend_d256.hip): no row maximum on the common path.  The scores are
          // exponentiated against the STANDING running max and the lane's partial row sum is the check (every p <= it;
          // compared as bits -- sums are never negative, so the unsigned order is the float order with inf and NaN on
          // top, and -fno-honor-nans cannot fold it away); only when a lane's sum runs away the wave takes the
          // thresholded max step (exact algebra: l uses the same m) and redoes the block.  The step's rule as before: the
          // reference max moves only when the tile's exceeds it by more than 2^8 (extend_attention.py:474-475 keeps a
          // fully masked row's max finite).
          float alpha = 1.0f, psum = 0.f;
          {
            const float m_old = m_run[c];
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              e[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_old));
              psum += e[j];
            }
            if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, kExtSumLimit)) != 0) {
              load_sv();
              float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])),
                               fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
              mt = quad_row_max(mt);
              mt *= c2;  // c2 > 0: max commutes with the scale
              const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;
              const float m_new = (mt_fixed > m_old + kExtMaxSlack) ? mt_fixed : m_old;
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
          if constexpr (VSCALE) {  // fp8-style per-tensor V scale: prefix (cached) part only
            const float vs = prefix ? a.v_scale : 1.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] *= vs;
          }
          u32x4 praw;
          praw[0] = pack2<T>(sv[0], sv[1]);
          praw[1] = pack2<T>(sv[2], sv[3]);
          praw[2] = pack2<T>(sv[4], sv[5]);
          praw[3] = pack2<T>(sv[6], sv[7]);
          pf[c] = __builtin_bit_cast(vec8, praw);
          // rescale O only when some row's max moved (alpha == 1 exactly otherwise)
          if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) oacc[c][nb] *= alpha;
          }
        }
        // ---- O^T += V^T P^T --------------------------------------------------------------------------
#pragma unroll
        for (int nb = NB / 2; nb < NB; ++nb) {  // second half of the fragments: in flight under the MFMAs
          const int chunk = ((2 * nb + (pp >> 1)) ^ swv) & (CPR - 1);
          vlo[nb] = T::ds_read_tr(rp0 + chunk * 16);
          vhi[nb] = T::ds_read_tr(rp1 + chunk * 16);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const vec8 av = __builtin_bit_cast(vec8, u32x4{vlo[nb][0], vlo[nb][1], vhi[nb][0], vhi[nb][1]});
#pragma unroll
          for (int c = 0; c < kCB; ++c) oacc[c][nb] = T::mfma(av, pf[c], oacc[c][nb]);
        }
      }
    }
    // ---- stage the next tile: registers (loaded one tile ago) -> the other LDS buffer ---------------
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
#pragma unroll
  for (int c = 0; c < kCB; ++c) {
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
    if (a.lse && g == 0)
      a.lse[(qo0 + m) * a.lse_stride_t + h * a.lse_stride_h] = m_run[c] * kLn2 + __logf(l);
  }
}

// ---- generic fallback: one wave per (query token, q head); any head dims <= 512 -------------------
template <typename T, typename IdxT, bool LINEAR>
__global__ __launch_bounds__(64) void extend_generic_kernel(const ExtendArgs a, int dk, int dv,
                                                            const int32_t* tok2req) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* qs = reinterpret_cast<float*>(dyn_smem);
  const int lane = threadIdx.x;
  const int h = blockIdx.x % a.hq;
  const int64_t tq = blockIdx.x / a.hq;  // global query token
  // find the request owning tq (bs is small; linear scan by every lane)
  int req = 0;
  for (int i = 0; i < a.bs; ++i)
    if (load_idx(a.qo_indptr, i + 1, a.qo64) <= tq) req = i + 1;
  if (req >= a.bs) return;
  (void)tok2req;
  const int kvh = h / a.group;
  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t m = static_cast<int32_t>(tq - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const IdxT* idx = reinterpret_cast<const IdxT*>(a.kv_indices) + kv0;
  const int32_t p_len = a.skip_prefix ? 0 : P;
  const bool masked = a.custom_mask != nullptr;
  const bool unified = a.unified_prefix != nullptr;
  const int32_t q_off = unified ? a.unified_prefix[req] : P;
  const int32_t n_end = (a.skip_extend || unified) ? 0 : ((a.causal && !masked) ? m + 1 : E);
  const int32_t mask_woff = (masked && !unified && a.window_kv_offsets) ? a.window_kv_offsets[req] : 0;
  const int64_t mask_row = unified ? static_cast<int64_t>(P) : static_cast<int64_t>(mask_woff) + P + E;
  const uint8_t* mask_row_p = masked ? a.custom_mask + a.mask_indptr[req] + m * mask_row + mask_woff : nullptr;
  const bool mask_prefix = masked && (unified || !a.skip_prefix_mask);
  const bool causal_in_list = unified && a.causal && !masked;
  float xai = 1.0f;
  if (a.xai_len > 0) {
    if (unified) {
      if (q_off + m >= a.xai_len) xai = static_cast<float>(a.xai_len) / (static_cast<float>(q_off + m) + 1.0f);
    } else if (P + m > a.xai_len) {
      xai = __log2f(static_cast<float>(P + m)) / __log2f(static_cast<float>(a.xai_len));
    }
  }
  for (int d = lane; d < dk; d += 64) qs[d] = T::to_f32(a.q[tq * a.q_stride_t + h * a.q_stride_h + d]);
  __syncthreads();
  constexpr int MAXV = 8;
  float acc[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) acc[j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int total = p_len + n_end;
  for (int base = 0; base < total; base += 64) {
    const int n = base + lane;
    const bool inb = n < total;
    const int nn = inb ? n : total - 1;
    const bool prefix = nn < p_len;
    const uint16_t *kp, *vp;
    if (prefix) {
      const int64_t slot = static_cast<int64_t>(idx[nn]);
      kp = a.k_buf + kvh * a.k_head_stride +
           slot_off<LINEAR>(slot, a.page_size, a.k_page_stride, a.k_tok_stride);
      vp = a.v_buf + kvh * a.v_head_stride +
           slot_off<LINEAR>(slot, a.page_size, a.v_page_stride, a.v_tok_stride);
    } else {
      kp = a.k_ext + (qo0 + nn - p_len) * a.k_stride_t + kvh * a.k_stride_h;
      vp = a.v_ext + (qo0 + nn - p_len) * a.v_stride_t + kvh * a.v_stride_h;
    }
    float s = 0.f;
    for (int d = 0; d < dk; ++d) s += qs[d] * T::to_f32(kp[d]);
    s *= prefix ? a.sm_scale * a.k_scale : a.sm_scale;
    if (a.logit_cap > 0.f) s = a.logit_cap * tanhf(s / a.logit_cap);
    s *= xai;
    bool keep = inb;
    if (a.bias) {  // score_mod.py:44-56: + bias[q token, head, q_pos - kv_pos] inside [0, bias_len)
      const int32_t rel = prefix ? q_off + m - nn : m - (nn - p_len);
      if (inb && rel >= 0 && rel < a.bias_len)
        s += load_bias<T>(static_cast<const char*>(a.bias) + (tq * a.bias_stride_t + h * a.bias_stride_h) * (a.bias_f32 ? 4 : 2), a.bias_f32, rel);
    }
    if (prefix && causal_in_list) keep = keep && (nn <= q_off + m);
    if (a.window > 0) {
      if (prefix) keep = keep && (q_off + m <= nn + a.window);
      else keep = keep && (m <= (nn - p_len) + a.window);
    }
    if (masked && keep && (!prefix || mask_prefix))
      keep = mask_row_p[prefix ? nn : P + (nn - p_len)] != 0;
    s = keep ? s * kLog2e : -INFINITY;
    float mt = s;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) mt = fmaxf(mt, __shfl_xor(mt, dd));
    const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;
    const float m_new = fmaxf(m_run, mt_fixed);
    const float alpha = fast_exp2(m_run - m_new);
    const float p = fast_exp2(s - m_new);
    float ps = p;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) ps += __shfl_xor(ps, dd);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) acc[j] *= alpha;
    const float pv = p * (prefix ? a.v_scale : 1.0f);
    const int nvalid = min(64, total - base);
    for (int j = 0; j < nvalid; ++j) {
      const float pj = __shfl(pv, j);
      const uint16_t* vpj = reinterpret_cast<const uint16_t*>(
          __shfl(reinterpret_cast<uint64_t>(vp), j));
#pragma unroll
      for (int c = 0; c < MAXV; ++c) {
        const int d = lane + 64 * c;
        if (d < dv) acc[c] += pj * T::to_f32(vpj[d]);
      }
    }
  }
  float den = l_run;
  if (a.sinks) den += fast_exp2(a.sinks[h] * kLog2e - m_run);
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {
    const int d = lane + 64 * c;
    if (d < dv) a.o[tq * a.o_stride_t + h * a.o_stride_h + d] = T::from_f32(acc[c] / den);
  }
  if (a.lse && lane == 0)
    a.lse[tq * a.lse_stride_t + h * a.lse_stride_h] = m_run * kLn2 + __logf(l_run);
}

template <typename T, typename IdxT, bool LINEAR>
static int launch_extend(const ExtendArgs& a, int dk, int dv, int64_t total_q, hipStream_t s) {
  // tree masks and the xai temperature live in the D = 128 kernel (rx_extend32.hip) and in the generic
  // kernel; the 16x16x32 kernel below does not carry them
  const bool mfma_ok = (dk == dv) && (dk == 64 || dk == 128) && !a.custom_mask && a.xai_len <= 0 && !a.unified_prefix && !a.bias;
  if (mfma_ok) {
    const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
    const bool vs = a.v_scale != 1.0f;
    const bool plain = a.window <= 0 && a.logit_cap <= 0.f;
    if (dk == 64) {
      {
        const bool cb4 = plain && !vs && a.mblocks >= 4 && RX_EXT_CB == 2;
        note_dispatch("extend_mfma_kernel<%s, 64, %s, %s, %s, %s, %d>", tname<T>(), tname<IdxT>(), tbool(LINEAR),
                      tbool(vs), tbool(plain), cb4 ? 4 : 2);
      }
      if (plain && !vs && a.mblocks >= 4 && RX_EXT_CB == 2) {  // (>= 4 blocks of 128 rows: extends past 384 tokens)
        // long extends: 64 queries per wave (a.mblocks counts 128-row blocks: two of them per workgroup)
        ExtendArgs b = a;
        b.mblocks = (a.mblocks + 1) / 2;
        const unsigned grid4 = static_cast<unsigned>(b.bs) * b.hq * b.mblocks;
        hipLaunchKernelGGL((extend_mfma_kernel<T, 64, IdxT, LINEAR, false, true, 4>), dim3(grid4), dim3(256), 0, s, b);
      } else if (plain) {
        if (vs) hipLaunchKernelGGL((extend_mfma_kernel<T, 64, IdxT, LINEAR, true, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((extend_mfma_kernel<T, 64, IdxT, LINEAR, false, true>), dim3(grid), dim3(256), 0, s, a);
      } else {
        if (vs) hipLaunchKernelGGL((extend_mfma_kernel<T, 64, IdxT, LINEAR, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((extend_mfma_kernel<T, 64, IdxT, LINEAR, false>), dim3(grid), dim3(256), 0, s, a);
      }
    } else {
      note_dispatch("extend_mfma_kernel<%s, 128, %s, %s, %s, false, 2>", tname<T>(), tname<IdxT>(), tbool(LINEAR), tbool(vs));
      if (vs) hipLaunchKernelGGL((extend_mfma_kernel<T, 128, IdxT, LINEAR, true>), dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((extend_mfma_kernel<T, 128, IdxT, LINEAR, false>), dim3(grid), dim3(256), 0, s, a);
    }
  } else {
    if (dv > 512) return fail(RX_ERR_UNSUPPORTED, "rx_extend_attn: v_head_dim %d > 512", dv);
    const unsigned grid = static_cast<unsigned>(total_q * a.hq);
    note_dispatch("extend_generic_kernel<%s, %s, %s>|dk%d,dv%d", tname<T>(), tname<IdxT>(), tbool(LINEAR), dk, dv);
    hipLaunchKernelGGL((extend_generic_kernel<T, IdxT, LINEAR>), dim3(grid), dim3(64),
                       dk * sizeof(float), s, a, dk, dv, (const int32_t*)nullptr);
  }
  return check_launch("rx_extend_attn");
}

int launch_extend32(const rx_extend_params* p, hipStream_t s);  // rx_extend32.hip (D = 128)
bool extend_nd_supports(int dk, int dv);                          // rx_extend_nd.hip (256/256, 192/128, ...)
int launch_extend_nd(const rx_extend_params* p, hipStream_t s);
bool extend_d256_supports(const rx_extend_params* p);              // rx_extend_d256.hip (256 / 256, AGPR accumulators, LDS-DMA tiles)
int launch_extend_d256(const rx_extend_params* p, hipStream_t s);
bool extend_mla_supports(const rx_extend_params* p);               // rx_extend_mla.hip (576 / 512 over one latent kv head)
int launch_extend_mla(const rx_extend_params* p, hipStream_t s);

}  // namespace rx

using namespace rx;

// total_q for the generic path is bs * max_extend_len (an upper bound); blocks past the real
// token count exit early.
static int extend_attn_impl(const rx_extend_params* p, void* stream);
extern "C" int rx_extend_attn(const rx_extend_params* p, void* stream) {
  RX_RANGE("rx_extend_attn");
  return rx::dump_on_error("extend_attn", extend_attn_impl(p, stream), p, p ? sizeof(*p) : 0);
}
static int extend_attn_impl(const rx_extend_params* p, void* stream) {
  RX_REQUIRE(p, "rx_extend_attn: params is null");
  RX_REQUIRE(p->bs >= 0 && p->max_extend_len >= 0, "rx_extend_attn: negative sizes");
  if (p->bs == 0 || p->max_extend_len == 0) return RX_OK;
  RX_REQUIRE(p->q && p->o, "rx_extend_attn: null q/o");
  RX_REQUIRE(p->unified_prefix_lens || (p->k_extend && p->v_extend), "rx_extend_attn: null k_extend/v_extend");
  if (p->unified_prefix_lens)
    RX_REQUIRE(!p->skip_prefix && p->kv.k_buf && p->kv.v_buf && p->kv_indices,
               "rx_extend_attn: the unified form reads every key from the pool (k_buf, v_buf, kv_indices)");
  RX_REQUIRE(p->qo_indptr && p->kv_indptr, "rx_extend_attn: null indptr");
  RX_REQUIRE(p->num_q_heads > 0 && p->num_kv_heads > 0 && p->num_q_heads % p->num_kv_heads == 0,
             "rx_extend_attn: Hq=%d must be a positive multiple of Hkv=%d", p->num_q_heads,
             p->num_kv_heads);
  RX_REQUIRE(p->dtype == RX_BF16 || p->dtype == RX_F16, "rx_extend_attn: dtype %d", p->dtype);
  RX_REQUIRE(p->kv.page_size >= 1, "rx_extend_attn: page_size < 1");
  if (!p->skip_prefix)
    RX_REQUIRE(p->kv.k_buf && p->kv.v_buf && p->kv_indices,
               "rx_extend_attn: prefix stage needs k_buf, v_buf and kv_indices");
  const int dk = p->head_dim, dv = p->v_head_dim;
  const bool mfma_ok = (dk == dv) && (dk == 64 || dk == 128);
  if (mfma_ok) {
    const int64_t all = p->q_stride_t | p->q_stride_h | p->k_stride_t | p->k_stride_h |
                        p->v_stride_t | p->v_stride_h | p->kv.k_page_stride | p->kv.k_tok_stride |
                        p->kv.k_head_stride | p->kv.v_page_stride | p->kv.v_tok_stride |
                        p->kv.v_head_stride;
    RX_REQUIRE(all % 8 == 0, "rx_extend_attn: q/k/v strides must be multiples of 8 elements");
    RX_REQUIRE((p->o_stride_t | p->o_stride_h) % 4 == 0,
               "rx_extend_attn: o strides must be multiples of 4 elements");
    RX_REQUIRE((((uintptr_t)p->q | (uintptr_t)p->k_extend | (uintptr_t)p->v_extend) & 15) == 0 &&
                   (((uintptr_t)p->kv.k_buf | (uintptr_t)p->kv.v_buf) & (p->kv.kv_fp8 ? 7 : 15)) == 0 &&
                   ((uintptr_t)p->o & 7) == 0,
               "rx_extend_attn: q/k/v/k_buf/v_buf must be 16-byte (fp8 pools: 8-byte) and o 8-byte aligned");
  }
  RX_REQUIRE(p->kv.kv_fp8 == 0 || p->kv.kv_fp8 == 1, "rx_extend_attn: kv_fp8 = %d", p->kv.kv_fp8);
  if (p->kv.kv_fp8 && !p->skip_prefix && !(mfma_ok && dk == 128))
    return fail(RX_ERR_UNSUPPORTED, "rx_extend_attn: an fp8 prefix pool needs head_dim 128, got %d/%d", dk, dv);
  if (p->custom_mask) RX_REQUIRE(p->mask_indptr, "rx_extend_attn: custom_mask given without mask_indptr");
  const bool extras = p->custom_mask || p->xai_temperature_len > 0 || p->unified_prefix_lens || p->score_bias;
  if (p->score_bias) {
    RX_REQUIRE(p->score_bias_len > 0, "rx_extend_attn: score_bias_len = %d", p->score_bias_len);
    RX_REQUIRE(((uintptr_t)p->score_bias & (p->score_bias_is_f32 ? 3 : 1)) == 0, "rx_extend_attn: misaligned score_bias");
    if (p->q_pack > 1) return fail(RX_ERR_UNSUPPORTED, "rx_extend_attn: score_bias does not combine with q_pack");
  }
  // the D = 128 kernel stores whole 16-byte row chunks of o
  const bool o16 = ((p->o_stride_t | p->o_stride_h) % 8 == 0) && ((uintptr_t)p->o & 15) == 0;
  if (mfma_ok && dk == 128 && !o16 && (p->kv.kv_fp8 || extras))
    return fail(RX_ERR_INVALID_ARG, "rx_extend_attn: o must be 16-byte aligned with strides that are multiples of 8");
  if (p->q_pack > 1) {
    if (!(mfma_ok && dk == 128 && o16) || p->unified_prefix_lens)
      return fail(RX_ERR_UNSUPPORTED, "rx_extend_attn: q_pack needs head_dim 128, 16-byte aligned o and no unified_prefix_lens");
    RX_REQUIRE(p->num_q_heads == p->num_kv_heads * p->q_pack, "rx_extend_attn: q_pack = %d must be Hq / Hkv = %d / %d",
               p->q_pack, p->num_q_heads, p->num_kv_heads);
  }
  // Dispatch (switches: rx_set_option, read here as plain ints).  D = 64 and, as an A/B switch, D = 128 on the
  // 16x16x32 AGPR / LDS-DMA template; D = 128 on the 32x32x16 kernel; the other MFMA head dims; the latent MLA shape;
  // everything else on the generic kernel below.
  const Options& opt = options();
  if (((dk == 128 && dv == 128 && opt.extend_d256_at128) || (dk == 64 && dv == 64)) && opt.extend_d256 && !p->score_bias &&
      extend_d256_supports(p)) {
    const int rc = launch_extend_d256(p, static_cast<hipStream_t>(stream));
    return rc != RX_OK ? rc : check_launch("rx_extend_attn");
  }
  if (mfma_ok && dk == 128 && o16 && (p->q_pack > 1 || p->kv.kv_fp8 || extras || !opt.extend_16x16_d128)) {  // 32x32x16 fast path
    const int rc = launch_extend32(p, static_cast<hipStream_t>(stream));
    return rc != RX_OK ? rc : check_launch("rx_extend_attn");
  }
  if (opt.extend_d256 && !p->score_bias && extend_d256_supports(p)) {
    const int rc = launch_extend_d256(p, static_cast<hipStream_t>(stream));
    return rc != RX_OK ? rc : check_launch("rx_extend_attn");
  }
  if (!extras && !p->kv.kv_fp8 && p->q_pack <= 1 && extend_nd_supports(dk, dv) && opt.extend_nd) {
    // MFMA 16x16x32 kernel for the other head dims, when the tensors allow 16-byte row chunks
    const int64_t all = p->q_stride_t | p->q_stride_h | p->k_stride_t | p->k_stride_h | p->v_stride_t | p->v_stride_h |
                        p->kv.k_page_stride | p->kv.k_tok_stride | p->kv.k_head_stride | p->kv.v_page_stride |
                        p->kv.v_tok_stride | p->kv.v_head_stride;
    const bool aligned = all % 8 == 0 && (p->o_stride_t | p->o_stride_h) % 4 == 0 &&
                         (((uintptr_t)p->q | (uintptr_t)p->k_extend | (uintptr_t)p->v_extend | (uintptr_t)p->kv.k_buf |
                           (uintptr_t)p->kv.v_buf) & 15) == 0 && ((uintptr_t)p->o & 7) == 0;
    if (aligned) {
      const int rc = launch_extend_nd(p, static_cast<hipStream_t>(stream));
      return rc != RX_OK ? rc : check_launch("rx_extend_attn");
    }
  }
  if (opt.extend_mla && !p->score_bias && extend_mla_supports(p)) {
    const int rc = launch_extend_mla(p, static_cast<hipStream_t>(stream));
    return rc != RX_OK ? rc : check_launch("rx_extend_attn");
  }
  ExtendArgs a;
  a.q = (const uint16_t*)p->q;
  a.k_ext = (const uint16_t*)p->k_extend;
  a.v_ext = (const uint16_t*)p->v_extend;
  a.o = (uint16_t*)p->o;
  a.q_stride_t = p->q_stride_t;
  a.q_stride_h = p->q_stride_h;
  a.k_stride_t = p->k_stride_t;
  a.k_stride_h = p->k_stride_h;
  a.v_stride_t = p->v_stride_t;
  a.v_stride_h = p->v_stride_h;
  a.o_stride_t = p->o_stride_t;
  a.o_stride_h = p->o_stride_h;
  a.k_buf = (const uint16_t*)p->kv.k_buf;
  a.v_buf = (const uint16_t*)p->kv.v_buf;
  a.page_size = p->kv.page_size;
  if ((a.page_size & (a.page_size - 1)) == 0) a.page_size = -(__builtin_ctz(a.page_size) + 1);
  a.k_page_stride = p->kv.k_page_stride;
  a.k_tok_stride = p->kv.k_tok_stride;
  a.k_head_stride = p->kv.k_head_stride;
  a.v_page_stride = p->kv.v_page_stride;
  a.v_tok_stride = p->kv.v_tok_stride;
  a.v_head_stride = p->kv.v_head_stride;
  a.qo_indptr = p->qo_indptr;
  a.qo64 = p->qo_indptr_is_i64;
  a.kv_indptr = p->kv_indptr;
  a.kv_indices = p->kv_indices;
  a.lse = p->lse;
  a.lse_stride_t = p->lse_stride_t;
  a.lse_stride_h = p->lse_stride_h;
  a.bs = p->bs;
  a.hq = p->num_q_heads;
  a.hkv = p->num_kv_heads;
  a.group = p->num_q_heads / p->num_kv_heads;
  a.mblocks = (p->max_extend_len + 4 * kQPerWaveDefault - 1) / (4 * kQPerWaveDefault);
  a.sm_scale = p->sm_scale;
  a.k_scale = p->k_scale;
  a.v_scale = p->v_scale;
  a.logit_cap = p->logit_cap;
  a.causal = p->is_causal;
  a.skip_prefix = p->skip_prefix;
  a.skip_extend = p->skip_extend;
  a.window = p->sliding_window_size;
  a.custom_mask = p->custom_mask;
  a.mask_indptr = p->mask_indptr;
  a.window_kv_offsets = p->window_kv_offsets;
  a.skip_prefix_mask = p->skip_prefix_custom_mask;
  a.xai_len = p->xai_temperature_len;
  a.unified_prefix = p->unified_prefix_lens;
  a.sinks = p->sinks;
  a.bias = p->score_bias;
  a.bias_f32 = p->score_bias_is_f32;
  a.bias_len = p->score_bias_len;
  a.bias_stride_t = p->score_bias_stride_t;
  a.bias_stride_h = p->score_bias_stride_h;
  const bool linear = p->kv.page_size == 1 ||
                      (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                       p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  const bool idx64 = p->kv_indices_is_i64 != 0;
  const int64_t total_q = static_cast<int64_t>(p->bs) * p->max_extend_len;
  auto s = static_cast<hipStream_t>(stream);
#define RX_GO(TT)                                                                          \
  (idx64 ? (linear ? launch_extend<TT, int64_t, true>(a, dk, dv, total_q, s)               \
                   : launch_extend<TT, int64_t, false>(a, dk, dv, total_q, s))             \
         : (linear ? launch_extend<TT, int32_t, true>(a, dk, dv, total_q, s)               \
                   : launch_extend<TT, int32_t, false>(a, dk, dv, total_q, s)))
  return p->dtype == RX_BF16 ? RX_GO(BF16) : RX_GO(F16);
#undef RX_GO
}
