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
// KV tiles of 64 tokens are staged once per workgroup (global -> registers, in flight for one whole
// tile -> padded LDS rows, double buffered, one barrier per tile, written AFTER the barrier):
//   S^T[tok][q] = K Q^T : A = K fragment (lane = (token&31, 8-wide d half)) by ds_read_b128,
//                         B = Q^T kept in 32 VGPRs.
//   softmax on the lane (packed-fp32 exp2(fma(s, c, -m)), exact lazy rescale), P packed to bf16 IS
//   the B operand of the next product (registers 8s..8s+7 of an S block = k-step s,
//   cdna_hip_programming.md §3).
//   O^T[d][q] += V^T P^T : A = V^T fragment by two ds_read_b64_tr_b16 per k-step in the matching
//                         permuted token order (16s + 8(j>>2) + 4h + (j&3)).
// Fully visible tiles run as a wave-level software pipeline (32-token online-softmax steps, MFMAs of
// one block under the softmax VALU of the other); boundary tiles take a plain masked path.
#include "rx_common.h"


#ifndef RX_EXT32_SMALL_WG_TILES
#define RX_EXT32_SMALL_WG_TILES 28  // below this many estimated tiles per workgroup: 128-query workgroups
#endif
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
  int32_t kv_fp8;  // prefix pool holds fp8 e4m3fn bytes (strides in bytes)
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
  const uint8_t* custom_mask;   // tree mask (speculative verify) or null
  const int64_t* mask_indptr;
  const int32_t* window_kv_offsets;
  int32_t skip_prefix_mask;     // 1: the prefix part is not masked
  int32_t xai_len;              // Grok temperature length or <= 0
  const int32_t* unified_prefix;
  int32_t q_pack;  // GQA-packed query rows: row m is token m / q_pack (1 = off) // K8 unified form: per-request prefix length, or null
  const float* sinks;
};

typedef __attribute__((ext_vector_type(16))) float f32x16;
#ifndef RX_EXT32_MAX_SLACK
#define RX_EXT32_MAX_SLACK 8.0f
#endif
constexpr float kMaxSlack = RX_EXT32_MAX_SLACK;  // see sm_slice: how far (log2 units) a tile max may exceed a row's reference max

constexpr int kD = 128, kRow = 256, kTok = 64;  // head dim, bytes per row, tokens per tile
// LDS images: padded rows instead of an XOR swizzle, so that fragment addresses are lane constant +
// immediate.  K rows step 17 chunks of 16 B: the 16 rows one ds_read_b128 pass touches land on 16
// different chunk positions.  V rows step 20 chunks: the 4 rows x 64 B of a ds_read_b64_tr_b16
// half-wave land on 4 different 64-B bank groups.  Staging writes whole 256-B rows: conflict-free.
constexpr int kKStride = kRow + 16, kVStride = kRow + 64;
constexpr int kKTile = kTok * kKStride, kVTile = kTok * kVStride, kBufBytes = kKTile + kVTile;


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


template <typename T>
__device__ __forceinline__ void pv_mfma(u32x4 a, u32x4 b, f32x16& c) {
  c = mfma32<T>(__builtin_bit_cast(typename T::vec8, a), __builtin_bit_cast(typename T::vec8, b), c);
}
// QK^T step: S^T (+)= K fragment x Q fragment
template <typename T, bool FIRST>
__device__ __forceinline__ void qk_mfma(u32x4 k, const typename T::vec8& q, f32x16& sc) {
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  sc = mfma32<T>(__builtin_bit_cast(typename T::vec8, k), q, FIRST ? zero16 : sc);
}

// This file is compiled with -fno-honor-nans (sglang_amd/build.py): with NaNs honoured hipcc
// canonicalises every MFMA result before fmaxf (v_max_f32 x, x, x -- one extra VALU per score).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }  // v_max3_f32
__device__ __forceinline__ float max2f(float a, float b) { return fmaxf(a, b); }

// max over lanes l and l^32 (one query's two register halves)
__device__ __forceinline__ float half_swap_max(float x) {
  float a = x, b = x;
  // not volatile: a pure function of its inputs, so the LDS fragment reads may move across it
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}

// KV8: the cached prefix is an fp8 e4m3fn pool; its rows are upcast (exact) on the way into LDS, the
// new tokens' K/V are 16-bit as always.
// PLAIN: the launch uses none of the per-request extras (tree mask, unified list, sliding window, Grok
// temperature, logit cap, GQA packing, window offsets) -- plain prefill / extend over a cached prefix.  Their
// fields are then compile-time constants: the fast loop of the general instance keeps ~30 more scalars alive and
// hipcc spills SGPRs into VGPR lanes, reading 26 of them back with v_readlane EVERY tile (VALU issue slots in a
// VALU-issue-bound loop).
// PKC (PLAIN instances only): the GQA packing factor as a compile-time constant (0 = none).  A long causal extend under
// GQA walks fewer tiles packed -- a 256-row block is 64 tokens x 4 heads instead of 256 tokens of one head, so its
// diagonal is one boundary tile instead of four -- but the general instance that used to serve q_pack lost that to its
// scalars; with the factor a constant (row -> token is a shift) the packed call keeps the PLAIN loop.
template <typename T, typename IdxT, bool LINEAR, bool VSCALE, int NW, bool KV8, bool PLAIN, int PKC = 0>
__global__ __launch_bounds__(64 * NW, 2) void extend_mfma32_kernel(const Ext32Args a_in) {
  static_assert(PKC == 0 || PLAIN, "a constant packing factor goes with the PLAIN instance");
  Ext32Args a = a_in;
  if constexpr (PLAIN) {
    a.q_pack = PKC > 0 ? PKC : 1;
    a.unified_prefix = nullptr;
    a.custom_mask = nullptr;
    a.mask_indptr = nullptr;
    a.window_kv_offsets = nullptr;
    a.skip_prefix_mask = 1;
    a.window = 0;
    a.xai_len = 0;
    a.logit_cap = 0.f;
  }
  using vec8 = typename T::vec8;
  using KvE = std::conditional_t<KV8, uint8_t, uint16_t>;  // prefix pool element
  constexpr int KS = kD / 16;                  // 8 k-steps of the QK^T product
  constexpr int DB = kD / 32;                  // 4 output d blocks of 32
  constexpr int THREADS = 64 * NW;
  constexpr int RPP = THREADS / 16;            // rows staged per pass
  constexpr int NPASS = kTok / RPP;            // 2 (NW=8) or 4 (NW=4)
  constexpr int QPW = 32;                      // queries per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | V tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;

  // XCD-aware decode of the block index: workgroups go to the 8 XCDs round robin, so kv head =
  // block mod Hkv pins each kv head's K/V rows (the shared prefix: 1.8 MB per head at config 3) to
  // one XCD's 4-MiB L2 instead of streaming all heads through every L2.
  int bid = blockIdx.x;
  const int kvh = bid % a.hkv;
  bid /= a.hkv;
  // query blocks are dealt heaviest first: under the causal mask block mb walks mb+1 times as many new-token
  // tiles as block 0, and a late heavy block is the kernel's tail
  const int mb = a.mblocks - 1 - bid % a.mblocks;
  bid /= a.mblocks;
  const int head = kvh * a.group + bid % a.group;
  const int req = bid / a.group;

  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t pack = PLAIN ? (PKC > 0 ? PKC : 1) : a.q_pack;
  // queries of this request as ROWS: one per new token, or (q_pack = G) one per (new token, q head of the group)
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0) * pack;
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t qb0 = mb * NW * QPW;
  // GQA packing (q_pack = G > 1): the workgroup's "head" is a KV head and its query rows are (token, q head of
  // the group) pairs, row = token * G + g, so a request with few new tokens still fills a 32-row block and its K/V
  // tiles are staged once for the whole group.  E counts rows; keys, positions and mask rows go by token = row / G;
  // q, o and lse are addressed as (token, kv head * G + g) of the caller's ordinary [tokens, Hq, D] tensors.
  const int32_t Ek = E / pack;
  if (qb0 >= E) return;
  const int32_t qbase = qb0 + w * QPW;
  const bool active = qbase < E;
  const IdxT* idx = reinterpret_cast<const IdxT*>(a.kv_indices) + kv0;
  const int m = qbase + ql;                      // this lane's query row (index inside the extend part)
  const int mp = pack == 1 ? m : m / pack;       // its token (position inside the extend part)

  // K8 unified form: the kv list holds prefix + new tokens; q_off = the query's distance from list start
  const bool unified = a.unified_prefix != nullptr;
  const int32_t q_off = unified ? a.unified_prefix[req] : P;
  // speculative tree mask: row of query m = mask_base + m * mask_row (+ woff + kv position)
  const bool masked = a.custom_mask != nullptr;
  const int32_t mask_woff = (masked && !unified && a.window_kv_offsets) ? a.window_kv_offsets[req] : 0;
  const int64_t mask_row = unified ? static_cast<int64_t>(P) : static_cast<int64_t>(mask_woff) + P + Ek;
  const uint8_t* mask_base = masked ? a.custom_mask + a.mask_indptr[req] + mask_woff : nullptr;
  const bool mask_prefix = masked && (unified || !a.skip_prefix_mask);
  const bool causal_in_list = unified && a.causal && !masked;  // the causal rule applies inside the kv list
  // Grok temperature: per-query multiplier of the scaled scores (1 when off)
  float xai = 1.0f;
  {
    const int32_t qidx = q_off + mp;
    if (a.xai_len > 0) {
      if (unified) {  // extend_attention.py:940-946
        if (qidx >= a.xai_len) xai = static_cast<float>(a.xai_len) / (static_cast<float>(qidx) + 1.0f);
      } else if (qidx > a.xai_len) {  // :336-343
        xai = __log2f(static_cast<float>(qidx)) / __log2f(static_cast<float>(a.xai_len));
      }
    }
  }

  // ---- Q^T fragments: lane (q, h) holds Q[q][16 ks + 8 h .. +8] ------------------------------------
  vec8 qf[KS];
  {
    const bool ok = m < E;
    const int32_t tk = ok ? mp : 0, gq = ok ? m - mp * pack : 0;
    const uint16_t* qp = a.q + (qo0 + tk) * a.q_stride_t + (head * pack + gq) * a.q_stride_h + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 16 * ks) : u32x4{0, 0, 0, 0};
      qf[ks] = __builtin_bit_cast(vec8, raw);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // Q landed before the loop (see rx_extend.hip)

  const bool no_ext = a.skip_extend || a.unified_prefix != nullptr;  // unified: every key comes from the pool
  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t n_end_wg = no_ext ? 0 : (a.causal ? min(Ek, (qb0 + NW * QPW - 1) / pack + 1) : Ek);
  const int32_t n_end_w = no_ext ? 0 : (a.causal ? min(Ek, (qbase + QPW - 1) / pack + 1) : Ek);
  const int nt1 = (p_len + kTok - 1) / kTok;
  const int nt2 = (n_end_wg + kTok - 1) / kTok;
  const int nt = nt1 + nt2;

  // ---- cooperative staging ------------------------------------------------------------------------
  const int st_row = tid >> 4, st_chunk = tid & 15;
  const KvE* kbuf_h = reinterpret_cast<const KvE*>(a.k_buf) + kvh * a.k_head_stride + 8 * st_chunk;
  const KvE* vbuf_h = reinterpret_cast<const KvE*>(a.v_buf) + kvh * a.v_head_stride + 8 * st_chunk;
  // The fast loop stages tiles t+1 / t+2 and fetches the indices of t+3 WITHOUT asking whether they exist (a guard
  // inside the fenced MFMA groups costs +1.6 k cycles per tile: hipcc's wait-count pass merges pessimistically at
  // every join).  Tiles past the end therefore resolve to rows that are always readable: extend rows are clamped to
  // [0, n_end_wg), and a launch without an extend part reads pool slot 0 (the padding slot) through a zero stride.
  const bool has_ext = nt2 > 0;
  const uint16_t* kext_h = has_ext ? a.k_ext + qo0 * a.k_stride_t + kvh * a.k_stride_h + 8 * st_chunk
                                   : reinterpret_cast<const uint16_t*>(kbuf_h);
  const uint16_t* vext_h = has_ext ? a.v_ext + qo0 * a.v_stride_t + kvh * a.v_stride_h + 8 * st_chunk
                                   : reinterpret_cast<const uint16_t*>(vbuf_h);
  const int64_t k_ext_stride = has_ext ? a.k_stride_t : 0, v_ext_stride = has_ext ? a.v_stride_t : 0;
  int32_t slot[NPASS];
  auto load_idx_tile = [&](int t) {
    if (t < nt1) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i)
        slot[i] = static_cast<int32_t>(idx[min(t * kTok + i * RPP + st_row, p_len - 1)]);
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) slot[i] = max(0, min((t - nt1) * kTok + i * RPP + st_row, n_end_wg - 1));
    }
  };
  u32x4 stg_k[NPASS], stg_v[NPASS];
  // 8 pool elements of one staged chunk: 16 B, or 8 B of an fp8 pool kept raw in the register's low
  // half until write_lds upcasts them (converting here would wait for the load at once)
  auto pool_load = [&](const KvE* p) {
    if constexpr (KV8) {
      const u32x2 raw = *reinterpret_cast<const u32x2*>(p);
      return u32x4{raw[0], raw[1], 0u, 0u};
    } else {
      return *reinterpret_cast<const u32x4*>(p);
    }
  };
  auto issue_loads = [&](int t) {
    if (t < nt1) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        stg_k[i] = pool_load(kbuf_h + slot_off32<LINEAR>(slot[i], a.page_size, a.k_page_stride, a.k_tok_stride));
        stg_v[i] = pool_load(vbuf_h + slot_off32<LINEAR>(slot[i], a.page_size, a.v_page_stride, a.v_tok_stride));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        stg_k[i] = *reinterpret_cast<const u32x4*>(kext_h + mul_u32(slot[i], k_ext_stride));
        stg_v[i] = *reinterpret_cast<const u32x4*>(vext_h + mul_u32(slot[i], v_ext_stride));
      }
    }
  };
  auto write_lds = [&](int buf, bool from_pool) {  // from_pool: the staged tile is a prefix tile
    char* kt = smem + buf * kBufBytes + st_row * kKStride + st_chunk * 16;
    char* vt = smem + buf * kBufBytes + kKTile + st_row * kVStride + st_chunk * 16;
    if (KV8 && from_pool) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        *reinterpret_cast<u32x4*>(kt + i * RPP * kKStride) = fp8x8_to_16<T>(u32x2{stg_k[i][0], stg_k[i][1]});
        *reinterpret_cast<u32x4*>(vt + i * RPP * kVStride) = fp8x8_to_16<T>(u32x2{stg_v[i][0], stg_v[i][1]});
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        *reinterpret_cast<u32x4*>(kt + i * RPP * kKStride) = stg_k[i];
        *reinterpret_cast<u32x4*>(vt + i * RPP * kVStride) = stg_v[i];
      }
    }
  };

  f32x16 oacc[DB];
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;

  // sliding window (plain lists only: no tree mask, unified list or window offsets): tiles wholly below the
  // workgroup's first row's bound are never staged (rx_extend.hip)
  int t0 = 0;
  if (a.window > 0 && !a.custom_mask && !a.unified_prefix && !a.window_kv_offsets) {
    const int32_t tok0 = qb0 / pack;
    t0 = min(nt1, max(0, P + tok0 - a.window) / kTok);
    if (t0 == nt1) t0 += min(nt2, max(0, tok0 - a.window) / kTok);
  }
  if (nt > t0) {
    load_idx_tile(t0);
    issue_loads(t0);
    if (nt > t0 + 1) load_idx_tile(t0 + 1);
    write_lds(t0 % 2, t0 < nt1);
    if (nt > t0 + 1) {
      issue_loads(t0 + 1);
      if (nt > t0 + 2) load_idx_tile(t0 + 2);
    }
  }

  // per-lane LDS read offsets: both images are padded rows, so every fragment address is ONE lane
  // constant plus an immediate (block / k-step / d-block distance)
  const int tq = lane & 15, qd = tq >> 2, pp = tq & 3, dg = (lane >> 4) & 1;
  const int kaddr = ql * kKStride + h * 16;                                        // K row 32 b + ql
  const int vaddr = kKTile + (4 * h + qd) * kVStride + (2 * dg + (pp >> 1)) * 16 + 8 * (pp & 1);
  const bool capped = a.logit_cap > 0.f;

  // K fragment (block b, k-step ks): lane (ql, h) <- K[32 b + ql][16 ks + 8 h .. +8]
  auto load_k = [&](const char* tile, int b, int ks) {
    return *reinterpret_cast<const u32x4*>(tile + kaddr + b * 32 * kKStride + ks * 32);
  };
  // V^T fragments of k-step `step` (16 tokens): rows 16 step + 4 h + qd (+8), d block db
  auto load_v1 = [&](const char* tile, int step, int db) {
    const u32x2 lo2 = T::ds_read_tr(tile + vaddr + step * 16 * kVStride + db * 64);
    const u32x2 hi2 = T::ds_read_tr(tile + vaddr + step * 16 * kVStride + db * 64 + 8 * kVStride);
    return u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
  };
  // exp2(s c2 - m) on one 32-token block, P packed to 16-bit as the next product's B operand; returns
  // the lane's partial row sum.  Scalar fp32 on purpose: v_pk_*_f32 beside MFMAs costs more than the
  // two scalar ops it replaces (MI355X_MICROARCH.md, per-instruction cycle constants).
  auto exp_pack = [&](f32x16& sc, float c2, float m_new, float vs, u32x4 (&pk)[2]) -> float {
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v0 = fast_exp2(__builtin_fmaf(sc[2 * i], c2, -m_new));
      float v1 = fast_exp2(__builtin_fmaf(sc[2 * i + 1], c2, -m_new));
      ps += v0 + v1;
      if constexpr (VSCALE) {
        v0 *= vs;
        v1 *= vs;
      }
      pk[i >> 2][i & 3] = pack2<T>(v0, v1);
    }
    return ps;
  };
  auto row_max16 = [&](const f32x16& sc) -> float {
    float m01 = max3f(sc[0], sc[1], sc[2]), m23 = max3f(sc[3], sc[4], sc[5]);
    m01 = max3f(m01, sc[6], sc[7]);
    m23 = max3f(m23, sc[8], sc[9]);
    m01 = max3f(m01, sc[10], sc[11]);
    m23 = max3f(m23, sc[12], sc[13]);
    m01 = max3f(m01, sc[14], sc[15]);
    return max2f(m01, m23);
  };

  // Per-tile facts.  `fast` = both 32-token blocks fully visible to every query of this wave: no mask.
  struct TileInfo {
    bool prefix, work, full, fast;
    int tile_n0, nblk;
    float cs, c2, vs;
  };
  auto tile_info = [&](int t) {
    TileInfo ti;
    ti.prefix = t < nt1;
    ti.tile_n0 = (ti.prefix ? t : t - nt1) * kTok;
    // list positions this wave can see at all (causal inside the list in the unified form)
    const int32_t lim = ti.prefix ? (causal_in_list ? min(p_len, q_off + qbase + QPW) : p_len) : n_end_w;
    ti.work = active && ti.tile_n0 < lim;
    ti.nblk = (ti.tile_n0 + 32 < lim) ? 2 : 1;  // visible 32-token blocks of this tile
    ti.cs = ti.prefix ? a.sm_scale * a.k_scale : a.sm_scale;
    ti.c2 = capped ? kLog2e : ti.cs * kLog2e;
    ti.vs = ti.prefix ? a.v_scale : 1.0f;
    const int n_hi = ti.tile_n0 + 32 * ti.nblk;
    if (ti.prefix) ti.full = n_hi <= p_len && a.window <= 0 && !mask_prefix && (!causal_in_list || n_hi - 1 <= q_off + qbase);
    else ti.full = n_hi <= Ek && (!a.causal || n_hi - 1 <= qbase / pack) && a.window <= 0 && !masked;
    ti.fast = ti.work && ti.full && ti.nblk == 2 && !capped && (LINEAR || a.page_size < 0);
    return ti;
  };
  // one barrier per tile; tile t+1 is written AFTER it (its readers, tile t-1's products, are done)
  // and tile t+2's global loads are re-issued at once, so they have this whole tile to land
  auto tile_sync_and_stage = [&](int t) {
    __syncthreads();
    if (t + 1 < nt) {
      write_lds((t + 1) % 2, t + 1 < nt1);
      if (t + 2 < nt) {
        issue_loads(t + 2);
        if (t + 3 < nt) load_idx_tile(t + 3);
      }
    }
  };

  // The two tile bodies live in two separate inner loops (runs of fast tiles, runs of boundary
  // tiles): with both bodies inside one loop hipcc's allocator spills 147 registers, each alone fits.
  // Waves of one workgroup may be in different loops at the same t; every tile is one barrier either way.
  int t = t0;
  while (t < nt) {
    // A run of fast tiles [t, fe): found ONCE per run in closed form (tile_info per tile cost ~310 cycles of scalar
    // work between the end of a tile and its barrier), and it may reach the very last tile: staging past the end is
    // harmless (see has_ext above).
    int fe = t;
    {
      const TileInfo ti0 = tile_info(t);
      if (ti0.fast) {  // the tile-independent conditions hold; the rest is "both blocks inside the visible range"
        if (t < nt1) fe = min(nt1, (causal_in_list ? min(p_len, q_off + qbase + 1) : p_len) / kTok);
        else fe = min(nt, nt1 + (a.causal ? min(Ek, qbase / pack + 1) : Ek) / kTok);
        fe = max(fe, t + 1);
      }
    }
    const float c2u = tile_info(t).c2, vs = tile_info(t).vs;  // constant inside a run (prefix or new tokens)
    // JUMPT: the jump test of sm_slice (plain instances only: one scale per run, no per-query temperature)
    constexpr bool JUMPT = PLAIN;
    float thr = (m_run + kMaxSlack) / (c2u * xai);
    for (; t < fe; ++t) {
      __syncthreads();
      const char* tile = smem + (t % 2) * kBufBytes;
      // ===== fast body: a hand-ordered wave-level software pipeline.  Measured before it: the tile's
      // phases (QK^T MFMAs, softmax VALU, PV MFMAs, staging) cost their SUM -- hipcc issues all MFMAs
      // of a phase back to back and the in-order wave then does its VALU with the matrix pipe idle.
      // Here every MFMA (one K or V^T fragment against the wave's 32-query block) is followed
      // by one slice of independent work, fenced so the order survives:
      //   QK^T(b0)            | K fragment reads two k-steps ahead
      //   QK^T(b1)            | softmax of block 0 (its own online-softmax step: no wait for b1's max)
      //   PV(b0), k-steps 0,1 | softmax of block 1, V^T fragment reads
      //   PV(b1), k-steps 2,3 | staging: tile t+1 registers -> LDS, tile t+2 global loads
      // (tile t+1 may be written any time after barrier t: its buffer's last readers were tile t-1's)
      f32x16 s0, s1;
      u32x4 pk0[2], pk1[2];
      u32x4 vfa[DB], vfb[DB];
      float ma, mb_, m0, m1, alpha0, alpha1;
      float ps0[2] = {0.f, 0.f}, ps1[2] = {0.f, 0.f};
      bool jumped0 = false, jumped1 = false;  // wave-uniform: slice j == 2 of block 0 / 1 moved a reference max (JUMPT)
      // one slice of a block's softmax; j = 0..6
      auto sm_slice = [&](int j, f32x16& sc, float m_prev, float& m_new, float& alpha, float (&ps)[2],
                          u32x4 (&pk)[2], bool& jumped) {
        const float c2 = c2u * xai;
        if (j == 0) {
          ma = max3f(sc[0], sc[1], sc[2]);
          mb_ = max3f(sc[3], sc[4], sc[5]);
          ma = max3f(ma, sc[6], sc[7]);
          mb_ = max3f(mb_, sc[8], sc[9]);
          asm volatile("" ::"v"(ma), "v"(mb_));  // anchors: hipcc otherwise sinks a slice to its first use
        } else if (j == 1) {
          ma = max3f(ma, sc[10], sc[11]);
          mb_ = max3f(mb_, sc[12], sc[13]);
          ma = max3f(ma, sc[14], sc[15]);
          ma = max2f(ma, mb_);
          asm volatile("" ::"v"(ma));
        } else if (j == 2) {
          // thresholded running max: a row moves its reference max only when the tile's max exceeds it by more
          // than kMaxSlack (log2 units).  exp2(s - m) then reaches 2^kMaxSlack at most -- exact algebra (l uses
          // the same m), fp32 sums and 16-bit P have the range -- and the O^T rescale, which costs 64
          // instructions per block, runs on the first tile and almost never again; with the plain rule some row
          // of a 32-row block sets a new max in ~70 % of 56 random tiles.
          // JUMPT (round 3): in a VALU-issue-bound loop even the TEST was 12 instructions per block (half swap, scale,
          // compare, select, exp2 of the difference).  Now one compare of the lane's raw maximum against a per-lane
          // threshold thr = (m + slack) / c2 and a wave-uniform branch: no lane above it means m_new = m_prev and
          // alpha = 1 for every row (a half row below the threshold cannot lift the row's maximum above it); only a
          // wave with a jumping row takes the old code, which also moves the threshold.
          if (!JUMPT || __builtin_amdgcn_ballot_w64(ma > thr) != 0) {
            float mt = half_swap_max(ma) * c2;
            mt = (mt == -INFINITY) ? -1e20f : mt;  // extend_attention.py:474-475
            const float m_cand = max2f(m_prev, mt);
            m_new = (m_cand - m_prev > kMaxSlack) ? m_cand : m_prev;
            alpha = fast_exp2(m_prev - m_new);
            if constexpr (JUMPT) {
              thr = (m_new + kMaxSlack) / c2;
              jumped = true;
            }
          } else {
            m_new = m_prev;
            alpha = 1.0f;
          }
          asm volatile("" ::"v"(m_new), "v"(alpha));
        } else {
          const int e = 4 * (j - 3);
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fast_exp2(__builtin_fmaf(sc[e + i], c2, -m_new));
          ps[0] += v[0] + v[2];
          ps[1] += v[1] + v[3];
          if constexpr (VSCALE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= vs;
          }
          const int pi = 2 * (j - 3);
          pk[pi >> 2][pi & 3] = pack2<T>(v[0], v[1]);
          pk[(pi + 1) >> 2][(pi + 1) & 3] = pack2<T>(v[2], v[3]);
          asm volatile("" ::"v"(pk[pi >> 2][pi & 3]), "v"(pk[(pi + 1) >> 2][(pi + 1) & 3]), "v"(ps[0]), "v"(ps[1]));
        }
      };
      {
        u32x4 kf[2 * KS + 8];
        constexpr int KA = 2;  // K fragments in flight ahead of the MFMA that consumes them
#pragma unroll
        for (int i = 0; i < KA; ++i) kf[i] = load_k(tile, i >> 3, i & 7);
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) {
          if (i + KA < 2 * KS) kf[i + KA] = load_k(tile, (i + KA) >> 3, (i + KA) & 7);
          if (i + 2 >= 2 * KS) {  // last two gaps: the first PV k-step's V^T fragments
            vfa[2 * (i + 2 - 2 * KS)] = load_v1(tile, 0, 2 * (i + 2 - 2 * KS));
            vfa[2 * (i + 2 - 2 * KS) + 1] = load_v1(tile, 0, 2 * (i + 2 - 2 * KS) + 1);
          }
          if (i == 0) qk_mfma<T, true>(kf[i], qf[0], s0);
          else if (i < KS) qk_mfma<T, false>(kf[i], qf[i], s0);
          else if (i == KS) qk_mfma<T, true>(kf[i], qf[0], s1);
          else qk_mfma<T, false>(kf[i], qf[i - KS], s1);
          if (i > KS) sm_slice(i - KS - 1, s0, m_run, m0, alpha0, ps0, pk0, jumped0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (JUMPT ? jumped0 : (__builtin_amdgcn_ballot_w64(alpha0 != 1.0f) != 0)) {
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[db] *= alpha0;
      }
      // PV(b0): k-steps 0 (vfa), 1 (vfb) | softmax of block 1
#pragma unroll
      for (int g = 0; g < 2 * DB; ++g) {
        if (g < 2) {
          vfb[2 * g] = load_v1(tile, 1, 2 * g);
          vfb[2 * g + 1] = load_v1(tile, 1, 2 * g + 1);
        }
        if (g < DB) pv_mfma<T>(vfa[g], pk0[0], oacc[g]);
        else pv_mfma<T>(vfb[g - DB], pk0[1], oacc[g - DB]);
        if (g >= DB && g < DB + 2) {
          vfa[2 * (g - DB)] = load_v1(tile, 2, 2 * (g - DB));
          vfa[2 * (g - DB) + 1] = load_v1(tile, 2, 2 * (g - DB) + 1);
        }
        if (g < 7) sm_slice(g, s1, m0, m1, alpha1, ps1, pk1, jumped1);
        __builtin_amdgcn_sched_barrier(0);
      }
      l_run = (l_run * alpha0 + (ps0[0] + ps0[1])) * alpha1 + (ps1[0] + ps1[1]);
      m_run = m1;
      if (JUMPT ? jumped1 : (__builtin_amdgcn_ballot_w64(alpha1 != 1.0f) != 0)) {
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[db] *= alpha1;
      }
      // PV(b1): k-steps 2 (vfa), 3 (vfb) | staging
      {
        const int t2 = t + 2;
        const bool pre = t2 < nt1;
        // one address form for pool rows and new rows: (slot >> sh) * page_stride + (slot & mask) * tok_stride
        // in BYTES (an fp8 pool's elements are bytes, everything else is 16-bit)
        const int esz = (KV8 && pre) ? 1 : 2;
        const char* kb = pre ? reinterpret_cast<const char*>(kbuf_h) : reinterpret_cast<const char*>(kext_h);
        const char* vb = pre ? reinterpret_cast<const char*>(vbuf_h) : reinterpret_cast<const char*>(vext_h);
        const int64_t kts = (pre ? a.k_tok_stride : k_ext_stride) * esz, vts = (pre ? a.v_tok_stride : v_ext_stride) * esz;
        const int64_t kps = a.k_page_stride * esz, vps = a.v_page_stride * esz;
        const int sh = (LINEAR || !pre) ? 31 : -a.page_size - 1;  // extend rows are never paged
        const uint32_t lo_mask = (1u << sh) - 1u;
        auto reissue = [&](int i) {
          const uint32_t sl = static_cast<uint32_t>(slot[i]);
          const char* kp = kb + mul_u32(sl >> sh, kps) + mul_u32(sl & lo_mask, kts);
          const char* vp = vb + mul_u32(sl >> sh, vps) + mul_u32(sl & lo_mask, vts);
          if (KV8 && pre) {
            const u32x2 kr = *reinterpret_cast<const u32x2*>(kp), vr = *reinterpret_cast<const u32x2*>(vp);
            stg_k[i] = u32x4{kr[0], kr[1], 0u, 0u};
            stg_v[i] = u32x4{vr[0], vr[1], 0u, 0u};
          } else {
            stg_k[i] = *reinterpret_cast<const u32x4*>(kp);
            stg_v[i] = *reinterpret_cast<const u32x4*>(vp);
          }
        };
#pragma unroll
        for (int g = 0; g < 2 * DB; ++g) {
          if (g < 2) {
            vfb[2 * g] = load_v1(tile, 3, 2 * g);
            vfb[2 * g + 1] = load_v1(tile, 3, 2 * g + 1);
          }
          if (g < DB) pv_mfma<T>(vfa[g], pk1[0], oacc[g]);
          else pv_mfma<T>(vfb[g - DB], pk1[1], oacc[g - DB]);
          if (g == 2) write_lds((t + 1) % 2, t + 1 < nt1);
          if (g >= 3 && g - 3 < NPASS) reissue(g - 3);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 2 * DB - 3; i < NPASS; ++i) reissue(i);
        load_idx_tile(t + 3);
      }
    }
    for (; t < nt; ++t) {
      const TileInfo ti = tile_info(t);
      if (ti.fast) break;
      tile_sync_and_stage(t);
      if (!ti.work) continue;
      const char* tile = smem + (t % 2) * kBufBytes;
      const bool prefix = ti.prefix, full = ti.full;
      const int tile_n0 = ti.tile_n0, nblk = ti.nblk;
      const float cs = ti.cs, c2b = ti.c2, vsb = ti.vs;
      // ===== boundary tiles: causal diagonal, ragged ends, window, logit cap
      f32x16 sacc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b < nblk) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            if (ks == 0) qk_mfma<T, true>(load_k(tile, b, ks), qf[ks], sacc[b]);
            else qk_mfma<T, false>(load_k(tile, b, ks), qf[ks], sacc[b]);
        }
      }
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
                keep = n < p_len && (!causal_in_list || n <= q_off + mp);
                if (a.window > 0) keep = keep && (q_off + mp <= n + a.window);
                if (mask_prefix && keep && m < E) keep = mask_base[mp * mask_row + n] != 0;
              } else {
                keep = n < n_end_w && (masked || !a.causal || n <= mp);
                if (a.window > 0) keep = keep && (mp <= n + a.window);
                if (masked && keep && m < E) keep = mask_base[mp * mask_row + P + n] != 0;
              }
              sacc[b][i] = keep ? sacc[b][i] : -INFINITY;
            }
          }
          mt = fmaxf(mt, row_max16(sacc[b]));
        }
      }
      const float c2 = c2b * xai;
      mt = half_swap_max(mt);
      mt *= c2;
      const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;  // extend_attention.py:474-475
      const float m_cand = fmaxf(m_run, mt_fixed);
      const float m_new = (m_cand - m_run > kMaxSlack) ? m_cand : m_run;
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
      u32x4 pk[2][2];  // [block][k-step within block]: 8 bf16 = registers 8s..8s+7
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (b < nblk) psum += exp_pack(sacc[b], c2, m_new, vsb, pk[b]);
      l_run = l_run * alpha + psum;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int db = 0; db < DB; ++db) oacc[db] *= alpha;
      }
#pragma unroll
      for (int step = 0; step < 4; ++step) {
        if (step < 2 * nblk) {
#pragma unroll
          for (int db = 0; db < DB; ++db) pv_mfma<T>(load_v1(tile, step, db), pk[step >> 1][step & 1], oacc[db]);
        }
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  // The accumulator has one query ROW per lane: stored as it stands, every store instruction touches 32
  // different rows (8 bytes each, sixteen of them per lane) and the tail is store-issue bound (~4 us per
  // workgroup).  Each wave therefore transposes its 32 x 128 block through a private LDS region (the K/V
  // tiles are dead after one more barrier) and writes whole 256-byte rows, 16 lanes x 16 B per row.
  __syncthreads();
  if (!active) return;
  constexpr int kORow = 272;  // 256 + 16: keeps ds_read_b128 aligned, spreads the row-per-lane writes
  char* obuf = smem + w * (32 * kORow);
  float l = l_run;
  {
    float a2 = l, b2 = l;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(a2), "+v"(b2));
    l = a2;
  }
  float den = l;
  if (a.sinks) den += fast_exp2(a.sinks[head * pack + (m < E ? m - mp * pack : 0)] * kLog2e - m_run);
  const float inv = 1.0f / den;
#pragma unroll
  for (int db = 0; db < DB; ++db) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {  // registers 4 gq .. 4 gq + 3 = d 32 db + 8 gq + 4 h + 0..3
      u32x2 pk2;
      pk2[0] = pack2<T>(oacc[db][4 * gq] * inv, oacc[db][4 * gq + 1] * inv);
      pk2[1] = pack2<T>(oacc[db][4 * gq + 2] * inv, oacc[db][4 * gq + 3] * inv);
      *reinterpret_cast<u32x2*>(obuf + ql * kORow + (32 * db + 8 * gq + 4 * h) * 2) = pk2;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int pss = 0; pss < 8; ++pss) {
    const int row = 4 * pss + (lane >> 4), chunk = lane & 15;
    const u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * kORow + chunk * 16);
    if (qbase + row < E) {
      const int32_t r = qbase + row, tk = pack == 1 ? r : r / pack, gq = r - tk * pack;
      *reinterpret_cast<u32x4*>(a.o + (qo0 + tk) * a.o_stride_t + (head * pack + gq) * a.o_stride_h + 8 * chunk) = v;
    }
  }
  if (a.lse && h == 0 && m < E)
    a.lse[(qo0 + mp) * a.lse_stride_t + (head * pack + (m - mp * pack)) * a.lse_stride_h] = m_run * kLn2 + __logf(l);
}

// launcher used by rx_extend.hip for head_dim == v_head_dim == 128
template <int NW, bool KV8, bool PLAIN, int PKC = 0>
static void launch32_nw(const Ext32Args& a, bool bf16, bool idx64, bool linear, bool vs, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  constexpr unsigned kLds = 2 * kBufBytes;  // 74 KiB: above the 64 KiB static limit, hence dynamic
  note_dispatch("extend_mfma32_kernel<%s, %s, %s, %s, %d, %s, %s, %d>", bf16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int",
                tbool(linear), tbool(!PLAIN && vs), NW, tbool(KV8), tbool(PLAIN), PKC);
#define RX_E32(TT, IT, LIN, VS)                                                                        \
  do {                                                                                                 \
    auto kern = extend_mfma32_kernel<TT, IT, LIN, VS, NW, KV8, PLAIN, PKC>;                            \
    static const hipError_t attr = hipFuncSetAttribute(                                                \
        reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);       \
    (void)attr;                                                                                        \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), kLds, s, a);                                   \
  } while (0)
#define RX_E32_VS(TT, IT, LIN)                                          \
  do {                                                                  \
    if constexpr (PLAIN) RX_E32(TT, IT, LIN, false);                    \
    else if (vs) RX_E32(TT, IT, LIN, true);                             \
    else RX_E32(TT, IT, LIN, false);                                    \
  } while (0)
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

int launch_extend32(const rx_extend_params* p, hipStream_t s) {
  const Options& opt = options();
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
  a.custom_mask = p->custom_mask; a.mask_indptr = p->mask_indptr; a.window_kv_offsets = p->window_kv_offsets;
  a.skip_prefix_mask = p->skip_prefix_custom_mask; a.xai_len = p->xai_temperature_len;
  a.unified_prefix = p->unified_prefix_lens;
  a.q_pack = p->q_pack > 1 ? p->q_pack : 1;
  // Long causal extends of a GQA-4 / GQA-8 model pack by themselves (bit-identical results, +2.6 % at the config-3 chunk: a
  // 256-row block's diagonal is one boundary tile instead of four); option ext32_autopack = 0 turns it off
  const int grp = p->num_kv_heads > 0 ? p->num_q_heads / p->num_kv_heads : 1;
  if (opt.ext32_autopack && a.q_pack == 1 && (grp == 4 || grp == 8) && p->num_q_heads == grp * p->num_kv_heads && p->is_causal &&
      !p->skip_extend && p->max_extend_len >= 256 && !p->kv.kv_fp8 && p->v_scale == 1.0f && !p->unified_prefix_lens &&
      !p->custom_mask && p->sliding_window_size <= 0 && p->xai_temperature_len <= 0 && !(p->logit_cap > 0.f) &&
      (p->avg_kv_len_hint + p->max_extend_len / 2) / kTok >= RX_EXT32_SMALL_WG_TILES)
    a.q_pack = grp;
  if (a.q_pack > 1) {  // the grid's heads are KV heads; their rows carry the q heads of the group
    a.hq = p->num_kv_heads;
    a.group = 1;
  }
  const bool linear = p->kv.page_size == 1 ||
                      (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                       p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  // Workgroup size.  One 256-query workgroup per CU (8 waves) is best when a (request, head, query block)
  // walks many tiles (config 3: 60 tiles, 739 vs 706 TFLOP/s); with few tiles the per-workgroup
  // prologue / epilogue and the launch itself dominate and two 128-query workgroups per CU overlap them
  // (no prefix, 2 Ki new tokens: 452 -> 512; 512 + 128: 356 -> 493).  Estimated tiles per workgroup from
  // the host-side hints: (mean prefix + half the longest extend) / 64.  Option ext32_small_wg: 0 / 1 force a form.
  const int est_tiles = (p->avg_kv_len_hint + p->max_extend_len / 2) / kTok;
  const bool small_wg = opt.ext32_small_wg < 0 ? est_tiles < RX_EXT32_SMALL_WG_TILES : opt.ext32_small_wg != 0;
  const int nw = small_wg ? 4 : 8;
  a.mblocks = (p->max_extend_len * a.q_pack + nw * 32 - 1) / (nw * 32);
  a.kv_fp8 = p->kv.kv_fp8;
  const bool bf = p->dtype == RX_BF16, i64 = p->kv_indices_is_i64 != 0, vsc = p->v_scale != 1.0f;
  const bool plain_any = !a.kv_fp8 && !vsc && !a.unified_prefix && !a.custom_mask && a.window <= 0 &&
                         a.xai_len <= 0 && !(a.logit_cap > 0.f) && opt.ext32_plain;
  const bool plain = plain_any && a.q_pack == 1;
  if (plain_any && (a.q_pack == 4 || a.q_pack == 8) && !small_wg) {  // packed rows on the PLAIN loop (GQA 4 / 8)
    if (a.q_pack == 4) launch32_nw<8, false, true, 4>(a, bf, i64, linear, false, s);
    else launch32_nw<8, false, true, 8>(a, bf, i64, linear, false, s);
    return RX_OK;
  }
  if (small_wg) {
    if (plain) launch32_nw<4, false, true>(a, bf, i64, linear, false, s);
    else if (a.kv_fp8) launch32_nw<4, true, false>(a, bf, i64, linear, vsc, s);
    else launch32_nw<4, false, false>(a, bf, i64, linear, vsc, s);
  } else {
    if (plain) launch32_nw<8, false, true>(a, bf, i64, linear, false, s);
    else if (a.kv_fp8) launch32_nw<8, true, false>(a, bf, i64, linear, vsc, s);
    else launch32_nw<8, false, false>(a, bf, i64, linear, vsc, s);
  }
  return RX_OK;
}

}  // namespace rx
