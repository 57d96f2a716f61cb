// K4/K5/K6: split-KV decode attention for gfx950.
//
// Reference: decode_attention_fwd (kernels/ops/attention/decode_attention.py:968-1044),
// stage 1 _fwd_grouped_kernel_stage1 (:383-608), stage 2 _fwd_kernel_stage2 (:731-805).
//
// MI355X design (not a translation of the Triton tiling):
//   * one 256-thread workgroup = 4 independent waves per (request, kv head, q-block, kv split);
//     wave w streams KV tiles w, w+4, ... of 32 tokens and keeps its own online-softmax state;
//     the four states are merged once through LDS at the end.  No barrier in the main loop.
//   * K goes HBM -> VGPR directly in MFMA A-operand shape (lane = (token&15, 8-element d
//     group)), 16 B per lane, every byte of a 256-B head row is fetched exactly once.
//   * S^T = K Q^T  with v_mfma_f32_16x16x32 (A = K tile, B = Q^T kept in registers, the up-to-16
//     query heads of the GQA group on the N axis).  The accumulator layout puts one q head per
//     (lane & 15) and 4 tokens per lane, so the bf16 P fragment IS the B operand of the next
//     product with no lane movement.
//   * O^T = V^T P^T: V is written once to a wave-private, XOR-swizzled LDS tile (ds_write_b128)
//     and read back transposed with ds_read_b64_tr_b16 as the A operand.
//   * softmax statistics live on the lane: the row max needs two cross-lane steps
//     (ds_bpermute via __shfl_xor 16/32), the row sum is reduced once after the loop.
//   * next tile's K/V loads are issued right after the current tile's operands are consumed,
//     so every wave keeps ~16 KiB of HBM reads in flight under its softmax + PV work.
#include "rx_common.h"

namespace rx {

struct DecodeArgs {
  const uint16_t* q;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, o_stride_t, o_stride_h;
  const uint16_t* k_buf;
  const uint16_t* v_buf;
  int32_t page_size;
  int64_t k_page_stride, k_tok_stride, k_head_stride;
  int64_t v_page_stride, v_tok_stride, v_head_stride;
  const int32_t* kv_indptr;
  const void* kv_indices;
  const int32_t* req_to_token;
  int64_t req_row_stride;
  const void* req_pool_indices;
  int32_t rpi64;
  const void* seq_lens;
  int32_t sl64;
  const int32_t* num_kv_splits;
  int32_t max_kv_splits;
  float* attn_logits;
  float* attn_lse;
  int32_t bs, hq, hkv, group, qblocks;
  float sm_scale;  // sm_scale * k_scale
  float v_scale, logit_cap;
  const float* sinks;
  int32_t kv_fp8;  // pool holds fp8 e4m3fn bytes (strides in bytes)
  int32_t xai_len; // Grok temperature length or <= 0
  const int32_t* kv_start;   // mode (b): request b attends tokens [kv_start[b], seq_len_b) only (or NULL)
  const uint16_t* extra_o;   // [num_extra, bs, hq, dv] partial outputs merged by stage 2 (or NULL)
  const float* extra_lse;    // [num_extra, bs, hq]
  int32_t num_extra;
  const int32_t* extra_index;  // row of request b inside one extra partial (NULL: b; < 0: the request has none)
  int32_t extra_rows;          // rows of one extra partial (bs unless extra_index compacts them)
  int32_t stages;  // 0 both, 1 stage 1 only, 2 stage 2 only
  int32_t* merge_counters;  // in-kernel stage 2 (rx_common.h split_arrive_is_last), or NULL: stage-2 launch
  // fused store of the new token (16-bit pools, one q block per kv head): its K / V rows [bs, Hkv, D], or NULL
  const uint16_t* k_new;
  const uint16_t* v_new;
  int64_t kn_stride_t, kn_stride_h, vn_stride_t, vn_stride_h;
  const int32_t* order;  // launch order of the requests (a permutation of 0..bs-1, longest first), or NULL
  // a request with ONE kv split writes its final output from stage 1 and stage 2 leaves it alone (MFMA kernel, no
  // extra partials): a length-aware schedule then costs the unsplit majority of a batch nothing
  int32_t direct_single;
  // compacted (request, split) pairs of the split schedule (rx_split_items), or NULL: bs x max_kv_splits slots
  const int32_t* items;
  const int32_t* items_count;
  int32_t items_occ3;  // the three-workgroups-per-CU instance asked for (split_items_wgs_per_cu == 3)
  int32_t items_cap;
  // relative-position score bias [bs, Hq, bias_len] (radix_hip.h: score_bias; decode_mfma_bias_kernel and the generic kernel), or NULL
  const void* bias;
  int32_t bias_f32, bias_len;
  int64_t bias_stride_t, bias_stride_h;
  // per-unit descriptors (radix_hip.h: unit_desc / unit_first_slots), req_to_token mode only, or NULL
  const int32_t* desc;
  const int32_t* first;
};

// Grok temperature factor of a request (decode_attention.py:156-160): the single query sits at seq_len-1
__device__ __forceinline__ float xai_factor(int32_t xai_len, int32_t seq_len) {
  const int32_t qidx = seq_len - 1;
  if (xai_len <= 0 || qidx <= xai_len) return 1.0f;
  return __log2f(static_cast<float>(qidx)) / __log2f(static_cast<float>(xai_len));
}

// Single-pass epilogue with extra partials (shared-prefix decode): folds the extras of (request b, head h),
// column d, into the running (max in log2 units, sum, weighted value) of this workgroup's own pass -- what
// stage 2 would do, without the fp32 partial round trip and the second launch.
template <typename T>
__device__ __forceinline__ void fold_extras(const DecodeArgs& a, int b, int h, int d, int dv, float& mx, float& lsum,
                                            float& acc) {
  constexpr int kBatch = 4;  // loads of a batch are issued together: one memory latency per batch, not per partial
  const int xb = a.extra_index ? a.extra_index[b] : b;
  if (xb < 0) return;  // not a member of any shared-prefix group
  for (int x0 = 0; x0 < a.num_extra; x0 += kBatch) {
    float xl[kBatch], xv[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int x = min(x0 + j, a.num_extra - 1);
      const int64_t xrow = (static_cast<int64_t>(x) * a.extra_rows + xb) * a.hq + h;
      xl[j] = a.extra_lse[xrow];
      xv[j] = T::to_f32(a.extra_o[xrow * dv + d]);
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      if (x0 + j >= a.num_extra || !(xl[j] > -INFINITY)) continue;  // past the end / empty partial (row undefined)
      const float m2 = xl[j] * kLog2e;
      const float nm = fmaxf(mx, m2);
      const float so = fast_exp2(mx - nm), w = fast_exp2(m2 - nm);
      acc = acc * so + w * xv[j];
      lsum = lsum * so + w;
      mx = nm;
    }
  }
}

// RX_DEC_TIMELINE (dev builds only: RX_CFLAGS=-DRX_DEC_TIMELINE RX_LIB_NAME=... RX_VARIANT_SOURCES=rx_decode.hip; tools/decode_timeline.py):
// wave 0 of every workgroup stamps the constant-rate clock (s_memrealtime, 100 MHz) at kernel entry, when its first K/V tile
// has landed, when its tile loop ends and at exit, plus the XCC it ran on, into a buffer of its own -- where a short launch's
// fixed cost goes.  No output value depends on a stamp; the product build compiles none of this.
#ifdef RX_DEC_TIMELINE
constexpr int kTlMax = 8192;
__device__ unsigned long long g_dec_timeline[kTlMax * 6];
#define RX_TL(slot)                                                                                         \
  do {                                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < kTlMax) g_dec_timeline[blockIdx.x * 6 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define RX_TL(slot) do {} while (0)
#endif

constexpr int kMinBlockKV = 32;  // decode_attention.py:36 (_MIN_BLOCK_KV)
constexpr int kTile = 32;        // tokens per wave tile (K of the PV MFMA)
constexpr int kWavesPerWG = 4;

// XOR swizzle of the 16-byte chunk index inside a V row of the LDS tile.  Makes both the
// ds_write_b128 of 8 consecutive rows and the ds_read_b64_tr_b16 of an 8-row x 32-byte
// window conflict-free (bank = addr/4 mod 32 for writes, mod 64 for tr reads).
template <int D>
__device__ __forceinline__ int v_swizzle(int row) {
  if constexpr (D == 64) return row & 7;
  return ((row & 7) << 1) | ((row >> 2) & 1);
}

template <bool LINEAR>
__device__ __forceinline__ int64_t slot_offset(int64_t slot, int32_t page_size,
                                               int64_t page_stride, int64_t tok_stride) {
  if constexpr (LINEAR) return mul_u32(slot, tok_stride);
  // page_size < 0 encodes a power-of-two page: -(log2(page) + 1)  (shift/mask instead of div/mod)
  if (page_size < 0) {
    const int sh = -page_size - 1;
    return mul_u32(slot >> sh, page_stride) + mul_u32(slot & ((1 << sh) - 1), tok_stride);
  }
  return (slot / page_size) * page_stride + (slot % page_size) * tok_stride;
}

struct SeqInfo {
  int32_t seq_len;   // tokens this launch attends
  int32_t full_len;  // the request's sequence length (position of the query + 1)
  const void* idx;   // token -> slot list of this request
};

// attended / full length of request b (mode (b) with kv_start: the suffix only)
__device__ __forceinline__ int32_t attended_len(const DecodeArgs& a, int b, int32_t& start) {
  start = 0;
  if (a.kv_indices) return a.kv_indptr[b + 1] - a.kv_indptr[b];
  const int32_t full = static_cast<int32_t>(load_idx(a.seq_lens, b, a.sl64));
  if (a.kv_start) start = min(max(a.kv_start[b], 0), full);
  return full - start;
}

template <typename IdxT>
__device__ __forceinline__ SeqInfo seq_info(const DecodeArgs& a, int b) {
  SeqInfo s;
  if (a.kv_indices) {
    const int32_t beg = a.kv_indptr[b];
    s.seq_len = s.full_len = a.kv_indptr[b + 1] - beg;
    s.idx = reinterpret_cast<const IdxT*>(a.kv_indices) + beg;
  } else {
    const int64_t req = load_idx(a.req_pool_indices, b, a.rpi64);
    int32_t start;
    s.seq_len = attended_len(a, b, start);
    s.full_len = s.seq_len + start;
    s.idx = a.req_to_token + req * a.req_row_stride + start;  // IdxT == int32_t in this mode
  }
  return s;
}

__device__ __forceinline__ void split_range(int32_t seq_len, int32_t splits, int32_t split,
                                            int32_t& lo, int32_t& hi) {
  // decode_attention.py:466-472
  const int32_t per =
      ((seq_len + splits - 1) / splits + kMinBlockKV - 1) / kMinBlockKV * kMinBlockKV;
  lo = per * split;
  hi = min(lo + per, seq_len);
}

#ifndef RX_DEC_MINW
#define RX_DEC_MINW 1  // min waves per SIMD requested from the register allocator
#endif
#ifndef RX_DEC_FP8_DEPTH
#define RX_DEC_FP8_DEPTH 2  // K/V register sets (tiles in flight per wave) of the fp8-pool kernel
#endif
#ifndef RX_DEC_16_DEPTH
#define RX_DEC_16_DEPTH 1  // K/V register sets of the 16-bit kernels (dev A/B: 2 = two tiles in flight per wave)
#endif
#ifndef RX_DEC_NT
#define RX_DEC_NT 0  // 1: non-temporal K/V loads
#endif

// one K/V fragment chunk = 8 elements per lane: 16 B of a 16-bit pool, 8 B of an fp8 pool
template <typename V, typename E>
__device__ __forceinline__ V kv_load8(const E* p) {
#if RX_DEC_NT
  return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
  return *reinterpret_cast<const V*>(p);
#endif
}
template <typename T, bool KV8, typename V>
__device__ __forceinline__ u32x4 kv_frag16(V raw) {
  if constexpr (KV8) return fp8x8_to_16<T>(raw);
  else return raw;
}

// FUSE: the request's NEWEST token (position seq_len - 1, the one this decode step produced) is read from k_new /
// v_new instead of the pool, and the lanes that hold its 16-byte chunks write them to its pool slot on the way -- the
// KV store of the step (K1) without its own launch (a small-batch or TP-shard layer is 35-100 us, the store launch
// ~5).  Only with ONE q block per kv head: then exactly one workgroup ever touches that row.
//
// OCC3: the register budget of THREE workgroups per CU (168 VGPRs: no spills in the plain D = 128 form; the fused-store
// form spills 7-14 dwords there and runs 10 % slower, so it has no such instance).  A split-items grid asks for it
// (rx_decode_params.split_items_wgs_per_cu) when its schedule cut a MIXED batch into ~3 x CUs near-equal workgroups that
// are all resident at once, so nothing waits for a second round (one 32 k request among 63 of 1 k: 83 us per layer at
// two per CU -> 78 at three); a uniform batch of one-pass requests is ~0.5 % faster at two.
//
// BIAS (decode_mfma_bias_kernel, round 5): the reference's relative_bias_score_mod (score_mod.py:44-56 through
// decode_attention.py:539-551) -- the score against list position n gets + bias[b, h, (len - 1) - n] inside [0, bias_len).
// Its own instances: the plain kernels carry none of it.
template <typename T, int D, typename IdxT, bool LINEAR, bool KV8, bool FUSE, bool OCC3, bool BIAS>
__device__ __forceinline__ void decode_mfma_body(const DecodeArgs& a, const int block_id) {
  static_assert(!(FUSE && KV8), "the fused store writes 16-bit rows");
  static_assert(!BIAS || (!FUSE && !KV8 && !OCC3), "the biased instances are the plain 16-bit ones");
  using vec8 = typename T::vec8;
  using KvE = std::conditional_t<KV8, uint8_t, uint16_t>;  // pool element
  using KvV = u32x4;  // 16 B per lane and load: 8 elements of a 16-bit pool, 16 of an fp8 pool
  // fp8 pools: a 16-B load is TWO 8-element k-groups.  The contraction index is free to permute (Q uses
  // the same map), so lane g takes d = 64 j + 16 g + 8 e + (0..7) for k-step 2 j + e: its 16 bytes of
  // load j are the operands of k-steps 2j and 2j+1.  (8-B loads touched 32 B of each 128-B row per
  // instruction and ran at 4.5 TB/s.)
  constexpr int NL = KV8 ? D / 64 : D / 32;  // loads per 16-token row block (KS = D / 32 k-steps)
  constexpr int KS = D / 32;  // k-steps of the QK^T product
  constexpr int NB = D / 16;  // 16-wide d blocks of the output
  // LDS row of the V tile: the row itself, or 256 B for D = 96 (the chunk swizzle permutes 16 chunk positions: a
  // 12-chunk row is stored in a 16-chunk slot, four positions stay empty)
  constexpr int ROW_BYTES = (D == 96) ? 256 : D * 2;
  constexpr int TILE_BYTES = kTile * ROW_BYTES;  // >= 16 * D * 4 (fp32 O^T of one wave)
  __shared__ __attribute__((aligned(16))) char smem[kWavesPerWG * TILE_BYTES + 2 * 4 * 16 * 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  RX_TL(0);
#ifdef RX_DEC_TIMELINE
  if (tid == 0 && blockIdx.x < kTlMax) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_dec_timeline[blockIdx.x * 6 + 4] = xcc & 0xf;
    g_dec_timeline[blockIdx.x * 6 + 5] = 0;
  }
#endif

  // split is the SLOWEST grid dimension: workgroups are dealt round-robin over the 8 XCDs, and
  // with split fastest only `splits` of every `max_kv_splits` consecutive blocks do any work, i.e.
  // the live blocks would pile onto splits/8 of the chip (measured: 2 live splits of 8 -> 2.0 TB/s).
  int bid = block_id;
  const int qb = bid % a.qblocks;
  bid /= a.qblocks;
  const int kvh = bid % a.hkv;
  bid /= a.hkv;
  // a ragged batch is dealt longest request first (a.order): the chip's last round of workgroups is then the
  // short requests, not a 4 k-token one that starts when the others are finishing
  int b, split;
  // With the per-unit tables (round 6) everything the prologue needs sits at an address known from the block index alone:
  // the unit's descriptor and the slot ids of its first tiles are requested together, before anything else has returned.
  const bool use_desc = a.desc != nullptr;  // (the host sets it in req_to_token mode only: IdxT == int32_t)
  u32x4 d0 = {0u, 0u, 0u, 0u}, d1 = {0u, 0u, 0u, 0u};
  int32_t fs0 = 0, fs1 = 0;
  if (use_desc) {
    const int32_t* dp = a.desc + 8 * static_cast<int64_t>(bid);
    d0 = *reinterpret_cast<const u32x4*>(dp);
    d1 = *reinterpret_cast<const u32x4*>(dp + 4);
    const int32_t* fp = a.first + 128 * static_cast<int64_t>(bid) + 32 * w + r;
    fs0 = fp[0];
    fs1 = fp[16];
  }
  SeqInfo si;
  int32_t splits;
  if (a.items) {  // live (request, split) pairs only, longest requests first (rx_decode_params.split_items)
    if (bid >= a.items_count[0]) return;
    if (!use_desc) {
      b = a.items[2 * bid];
      split = a.items[2 * bid + 1];
    }
  } else if (!use_desc) {
    b = a.order ? a.order[bid % a.bs] : bid % a.bs;
    split = bid / a.bs;
  }
  if (use_desc) {
    b = static_cast<int>(d0[0]);
    split = static_cast<int>(d0[1]);
    si.seq_len = si.full_len = static_cast<int32_t>(d0[2]);
    splits = static_cast<int32_t>(d0[3]);
    si.idx = a.req_to_token + static_cast<int64_t>((static_cast<uint64_t>(d1[1]) << 32) | d1[0]);
  } else {
    si = seq_info<IdxT>(a, b);
    splits = (a.num_kv_splits && a.max_kv_splits > 1) ? a.num_kv_splits[b] : 1;
  }
  const IdxT* idx = reinterpret_cast<const IdxT*>(si.idx);
  const bool single = (a.max_kv_splits == 1) || (a.direct_single && splits == 1);
  const int gq = qb * 16 + r;            // q head inside the GQA group handled by this lane
  const bool q_valid = gq < a.group;
  const int h = kvh * a.group + gq;      // global q head

  if (!single && a.merge_counters && si.seq_len == 0) {
    // no workgroup of this request will ever arrive (its split count may even be 0): split 0's workgroup writes
    // what the stage-2 kernel writes for zero live splits
    if (split == 0)
      for (int i = tid; i < 16 * D; i += 256) {
        const int q = i / D, d = i % D;
        if (qb * 16 + q >= a.group) continue;
        const int hh = kvh * a.group + qb * 16 + q;
        const float e_sum = a.sinks ? INFINITY : 0.f;  // exp(sink - (-inf))
        a.o[b * a.o_stride_t + hh * a.o_stride_h + d] = T::from_f32(0.f * (a.v_scale / e_sum));
      }
    return;
  }
  if (split >= splits) return;
  int32_t lo, hi;
  if (use_desc) {  // (the table's builder ran the same split_range)
    lo = static_cast<int32_t>(d1[2]);
    hi = static_cast<int32_t>(d1[3]);
  } else {
    split_range(si.seq_len, splits, split, lo, hi);
  }
  if (hi <= lo) {
    if (single && si.seq_len == 0) {  // empty request: define the output (reference: 0/0)
      for (int i = tid; i < 16 * D; i += 256) {
        const int q = i / D, d = i % D;
        if (qb * 16 + q >= a.group) continue;
        const int hh = kvh * a.group + qb * 16 + q;
        float out = 0.f;
        if (a.num_extra) {  // nothing of its own to attend: the extra partials are the whole result
          float mx = -INFINITY, lsum = 0.f, acc = 0.f;
          fold_extras<T>(a, b, hh, d, D, mx, lsum, acc);
          float den = lsum;
          if (a.sinks) den += fast_exp2(a.sinks[hh] * kLog2e - mx);
          out = acc / den * a.v_scale;
        }
        a.o[b * a.o_stride_t + hh * a.o_stride_h + d] = T::from_f32(out);
      }
    }
    return;
  }
  const int ntiles = (hi - lo + kTile - 1) / kTile;

  const KvE* kbase = reinterpret_cast<const KvE*>(a.k_buf) + kvh * a.k_head_stride + (KV8 ? 16 : 8) * g;
  const KvE* vbase = reinterpret_cast<const KvE*>(a.v_buf) + kvh * a.v_head_stride + (KV8 ? 16 : 8) * g;
  char* vt = smem + w * TILE_BYTES;  // this wave's V tile

  f32x4 oacc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY;  // running max (log2 domain), identical in the 4 lanes of a q head
  float l_run = 0.f;        // this lane's partial row sum

  const float xai = xai_factor(a.xai_len, si.full_len);
  const float scale_log2 = a.sm_scale * kLog2e * xai;

  auto load_slots = [&](int t, int64_t& s0, int64_t& s1) {
    const int32_t t0 = min(lo + t * kTile + r, hi - 1);
    const int32_t t1 = min(lo + t * kTile + 16 + r, hi - 1);
    s0 = static_cast<int64_t>(idx[t0]);
    s1 = static_cast<int64_t>(idx[t1]);
  };
  // FUSE: this workgroup's range ends at the request's newest token, whose rows are not in the pool yet
  const bool has_new = FUSE && hi == si.seq_len;
  // Register sets of K/V tiles in flight per wave.  One 32-token tile is 16 KiB of a 16-bit pool but
  // 8 KiB of an fp8 pool: with a single set the fp8 kernel has half the bytes in flight and ran at
  // 4.55 TB/s; two sets restore the 128 KiB per CU of the 16-bit kernel in the same registers.
  constexpr int DEPTH = KV8 ? RX_DEC_FP8_DEPTH : ((D <= 128 && !OCC3 && !BIAS) ? RX_DEC_16_DEPTH : 1);
  KvV kf[DEPTH][2][NL], vf[DEPTH][2][NL];  // fp8 pools: upcast (exact) where consumed
  constexpr int LSTEP = KV8 ? 64 : 32;  // elements between a lane's consecutive loads
  int64_t new_slot = 0;  // FUSE: pool slot of the newest token, kept from the slot list (no dependent load at the tail)
  auto load_kv = [&](int t, int64_t s0, int64_t s1, KvV (&kfs)[2][NL], KvV (&vfs)[2][NL]) {
    const int64_t ko0 = slot_offset<LINEAR>(s0, a.page_size, a.k_page_stride, a.k_tok_stride);
    const int64_t ko1 = slot_offset<LINEAR>(s1, a.page_size, a.k_page_stride, a.k_tok_stride);
    const int64_t vo0 = slot_offset<LINEAR>(s0, a.page_size, a.v_page_stride, a.v_tok_stride);
    const int64_t vo1 = slot_offset<LINEAR>(s1, a.page_size, a.v_page_stride, a.v_tok_stride);
    const KvE* kp0 = kbase + ko0;
    const KvE* kp1 = kbase + ko1;
    const KvE* vp0 = vbase + vo0;
    const KvE* vp1 = vbase + vo1;
    if constexpr (FUSE) {
      // rows that are (or clamp to) the newest token come from k_new / v_new; only the last tile can hold it
      if (has_new && t == ntiles - 1) {
        // (computed here, not hoisted: the pointers would otherwise sit in registers through the whole loop)
        const uint16_t* knew = a.k_new + b * a.kn_stride_t + kvh * a.kn_stride_h + 8 * g;
        const uint16_t* vnew = a.v_new + b * a.vn_stride_t + kvh * a.vn_stride_h + 8 * g;
        if (lo + t * kTile + r >= hi - 1) {
          kp0 = reinterpret_cast<const KvE*>(knew);
          vp0 = reinterpret_cast<const KvE*>(vnew);
          if constexpr (!OCC3) new_slot = s0;  // (clamped rows carry the same slot)
        }
        if (lo + t * kTile + 16 + r >= hi - 1) {
          kp1 = reinterpret_cast<const KvE*>(knew);
          vp1 = reinterpret_cast<const KvE*>(vnew);
          if constexpr (!OCC3) new_slot = s1;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NL; ++s) {
      kfs[0][s] = kv_load8<KvV>(kp0 + LSTEP * s);
      kfs[1][s] = kv_load8<KvV>(kp1 + LSTEP * s);
    }
#pragma unroll
    for (int s = 0; s < NL; ++s) {
      vfs[0][s] = kv_load8<KvV>(vp0 + LSTEP * s);
      vfs[1][s] = kv_load8<KvV>(vp1 + LSTEP * s);
    }
  };
  // 8 pool elements of k-step s as a 16-bit MFMA operand / LDS chunk
  auto frag16 = [&](const KvV (&fs)[NL], int s) -> u32x4 {
    if constexpr (KV8) return fp8x8_to_16<T>(u32x2{fs[s >> 1][2 * (s & 1)], fs[s >> 1][2 * (s & 1) + 1]});
    else return fs[s];
  };

  // set u holds tile w + 4 (u + DEPTH k); n0/n1[u] = the slots of the tile that will refill it
  constexpr int STEP = kWavesPerWG * DEPTH;
  int64_t n0[DEPTH], n1[DEPTH];
  // The first tiles' slot ids are requested BEFORE the q rows and land with them (one wait): the prologue is a chain of
  // dependent round trips -- kernel arguments -> request row / length -> slot ids -> K / V rows -> first product -- and q used
  // to be a link of its own in it (tools/decode_timeline.py, round 6: entry -> first tile landed 5.0 us on an idle chip).
  int64_t f0[DEPTH], f1[DEPTH];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    n0[u] = n1[u] = f0[u] = f1[u] = 0;
    const int tt = w + kWavesPerWG * u;
    if (tt < ntiles) {
      if (use_desc && u == 0) {  // tile w's slot ids came with the descriptor
        f0[0] = static_cast<int64_t>(fs0);
        f1[0] = static_cast<int64_t>(fs1);
      } else {
        load_slots(tt, f0[u], f1[u]);
      }
    }
  }
  // ---- Q^T fragments (B operand): lane (r,g) holds Q[h][32s + 8g .. +8] ----------------
  vec8 qf[KS];
  {
    const uint16_t* qp = a.q + b * a.q_stride_t + (q_valid ? h : 0) * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // element offset of k-step s for this lane (qp already carries + 8 g)
      const int qoff = KV8 ? 64 * (s >> 1) + 8 * g + 8 * (s & 1) : 32 * s;  // fp8: 64 j + 16 g + 8 e
      u32x4 raw = q_valid ? *reinterpret_cast<const u32x4*>(qp + qoff) : u32x4{0, 0, 0, 0};
      qf[s] = __builtin_bit_cast(vec8, raw);
    }
  }

  // The Q fragments must have LANDED before the tile loop: hipcc's waitcnt pass merges the loop-entry
  // state (Q loads possibly pending) into the loop header and would otherwise emit vmcnt(0) in front
  // of the first MFMA of EVERY iteration, i.e. wait for the next tile's prefetch before computing.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt/expcnt untouched
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
    const int tt = w + kWavesPerWG * u;
    if (tt < ntiles) {
      load_kv(tt, f0[u], f1[u], kf[u], vf[u]);
      if (tt + STEP < ntiles) load_slots(tt + STEP, n0[u], n1[u]);
    }
  }

  for (int t0 = w; t0 < ntiles; t0 += STEP) {
#pragma unroll
   for (int u = 0; u < DEPTH; ++u) {  // unrolled: the register set is a compile-time index
    const int t = t0 + kWavesPerWG * u;
    if (t >= ntiles) break;
#ifdef RX_DEC_TIMELINE
    if (t0 == w && u == 0) {  // the first tile's operands have landed
      __builtin_amdgcn_s_waitcnt(0x0F70);
      RX_TL(1);
    }
#endif
    // ---- S^T[token][q] = K Q^T ---------------------------------------------------------
    f32x4 sacc[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      sacc[bb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        sacc[bb] = T::mfma(__builtin_bit_cast(vec8, frag16(kf[u][bb], s)), qf[s], sacc[bb]);
    }
    // ---- V tile -> LDS (row = token, swizzled 16-B chunks) -------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int row = 16 * bb + r;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // 16-byte chunk of V[row][d .. d+8): d = 32 s + 8 g, or the fp8 map 64 j + 16 g + 8 e
        const int c = KV8 ? 8 * (s >> 1) + 2 * g + (s & 1) : 4 * s + g;
        const int chunk = c ^ v_swizzle<D>(row);
        *reinterpret_cast<u32x4*>(vt + row * ROW_BYTES + chunk * 16) = frag16(vf[u][bb], s);
      }
    }
    // ---- prefetch the next tile (registers of this tile are free again) ------------------
    if constexpr (FUSE) {
      // the newest token's chunks, still in this tile's registers, go to its pool slot (the lane that holds the
      // token itself, not the clamped copies): the step's KV store
      if (has_new && t == ntiles - 1) {
        const int32_t tn = hi - 1 - (lo + t * kTile);  // row of the newest token in this tile
        if (r == (tn & 15)) {
          const int bb = tn >> 4;
          // (three workgroups per CU: no register pair to carry the slot in -- read it again, one dependent load in ONE
          // wave's last tile)
          if constexpr (OCC3) new_slot = static_cast<int64_t>(idx[hi - 1]);
          KvE* kd = const_cast<KvE*>(kbase) + slot_offset<LINEAR>(new_slot, a.page_size, a.k_page_stride, a.k_tok_stride);
          KvE* vd = const_cast<KvE*>(vbase) + slot_offset<LINEAR>(new_slot, a.page_size, a.v_page_stride, a.v_tok_stride);
#pragma unroll
          for (int s = 0; s < NL; ++s) {
            *reinterpret_cast<KvV*>(kd + LSTEP * s) = bb ? kf[u][1][s] : kf[u][0][s];
            *reinterpret_cast<KvV*>(vd + LSTEP * s) = bb ? vf[u][1][s] : vf[u][0][s];
          }
        }
      }
    }
    if (t + STEP < ntiles) {
      load_kv(t + STEP, n0[u], n1[u], kf[u], vf[u]);
      if (t + 2 * STEP < ntiles) load_slots(t + 2 * STEP, n0[u], n1[u]);
    }
    // ---- online softmax on the lane ------------------------------------------------------
    float sv[8];
    const int tok_base = lo + t * kTile + 4 * g;
    float mt = -INFINITY;
    float bv[8];
    if constexpr (BIAS) {
      // the lane's 8 tokens sit rel0, rel0 - 1, ... behind the query (position len - 1); only a request's last
      // bias_len tokens carry a bias (a wave-uniform skip for every tile before them)
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = 0.f;
      const int32_t rel0 = si.seq_len - 1 - tok_base;
      if (si.seq_len - 1 - (lo + t * kTile + kTile - 1) < a.bias_len && q_valid) {
        const char* brow = static_cast<const char*>(a.bias) + (b * a.bias_stride_t + h * a.bias_stride_h) * (a.bias_f32 ? 4 : 2);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int32_t rel = rel0 - 16 * bb - i;
            if (rel >= 0 && rel < a.bias_len) bv[bb * 4 + i] = load_bias<T>(brow, a.bias_f32, rel) * kLog2e;
          }
      }
    }
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = sacc[bb][i];
        if (a.logit_cap > 0.f) {
          x = a.logit_cap * tanhf(x * a.sm_scale / a.logit_cap) * (kLog2e * xai);
        } else {
          x *= scale_log2;
        }
        if constexpr (BIAS) x += bv[bb * 4 + i];
        x = (tok_base + 16 * bb + i < hi) ? x : -INFINITY;
        sv[bb * 4 + i] = x;
        mt = fmaxf(mt, x);
      }
    mt = quad_row_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sv[j] = fast_exp2(sv[j] - m_new);
      psum += sv[j];
    }
    l_run = l_run * alpha + psum;
    u32x4 praw;
    praw[0] = pack2<T>(sv[0], sv[1]);
    praw[1] = pack2<T>(sv[2], sv[3]);
    praw[2] = pack2<T>(sv[4], sv[5]);
    praw[3] = pack2<T>(sv[6], sv[7]);
    const vec8 pf = __builtin_bit_cast(vec8, praw);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) oacc[nb] *= alpha;

    // ---- O^T += V^T P^T, V^T fragments by transposed LDS reads ---------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const int qd = r >> 2, pp = r & 3;
      const int row0 = 4 * g + qd;  // + 16 for the second 16-token block
      const int sw = v_swizzle<D>(row0);  // same for row0 + 16
      const char* rp0 = vt + row0 * ROW_BYTES + 8 * (pp & 1);
      const char* rp1 = rp0 + 16 * ROW_BYTES;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int chunk = (2 * nb + (pp >> 1)) ^ sw;
        const u32x2 lo2 = T::ds_read_tr(rp0 + chunk * 16);
        const u32x2 hi2 = T::ds_read_tr(rp1 + chunk * 16);
        const u32x4 av = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
        oacc[nb] = T::mfma(__builtin_bit_cast(vec8, av), pf, oacc[nb]);
      }
    }
   }
  }

  // ---- merge the four waves through LDS ----------------------------------------------------
  RX_TL(2);
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  float* sm_m = reinterpret_cast<float*>(smem + kWavesPerWG * TILE_BYTES);
  float* sm_l = sm_m + kWavesPerWG * 16;
  __syncthreads();  // every wave is done with its V tile
  {
    float* ot = reinterpret_cast<float*>(vt);  // [16 q][D] fp32
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      *reinterpret_cast<f32x4*>(ot + r * D + 16 * nb + 4 * g) = oacc[nb];
    if (g == 0) {
      sm_m[w * 16 + r] = m_run;
      sm_l[w * 16 + r] = l_run;
    }
  }
  __syncthreads();
  for (int i = tid; i < 16 * D; i += 256) {
    const int q = i / D, d = i % D;
    const int gq2 = qb * 16 + q;
    if (gq2 >= a.group) continue;
    const int hh = kvh * a.group + gq2;
    float mx = sm_m[q];
#pragma unroll
    for (int ww = 1; ww < kWavesPerWG; ++ww) mx = fmaxf(mx, sm_m[ww * 16 + q]);
    float lsum = 0.f, acc = 0.f;
#pragma unroll
    for (int ww = 0; ww < kWavesPerWG; ++ww) {
      const float sc = fast_exp2(sm_m[ww * 16 + q] - mx);
      lsum += sm_l[ww * 16 + q] * sc;
      acc += reinterpret_cast<const float*>(smem + ww * TILE_BYTES)[q * D + d] * sc;
    }
    if (single) {
      if (a.num_extra) fold_extras<T>(a, b, hh, d, D, mx, lsum, acc);
      float den = lsum;
      if (a.sinks) den += fast_exp2(a.sinks[hh] * kLog2e - mx);
      a.o[b * a.o_stride_t + hh * a.o_stride_h + d] = T::from_f32(acc / den * a.v_scale);
    } else {
      const int64_t row = (static_cast<int64_t>(b) * a.hq + hh) * a.max_kv_splits + split;
      if (a.merge_counters) {  // partials that another XCD's workgroup may merge: device-scope write-through stores
        store_dev(a.attn_logits + row * D + d, acc / lsum);
        if (d == 0) store_dev(a.attn_lse + row, mx * kLn2 + __logf(lsum));
      } else {
        a.attn_logits[row * D + d] = acc / lsum;
        if (d == 0) a.attn_lse[row] = mx * kLn2 + __logf(lsum);
      }
    }
  }
  RX_TL(3);
  if (!single && a.merge_counters) {  // stage 2 here: the last of this head block's live splits merges them
    const int32_t per = ((si.seq_len + splits - 1) / splits + kMinBlockKV - 1) / kMinBlockKV * kMinBlockKV;
    const int32_t live = min((si.seq_len + per - 1) / per, min(splits, a.max_kv_splits));
    if (!split_arrive_is_last(a.merge_counters + (b * a.hkv + kvh) * a.qblocks + qb, live)) return;
    const int h0 = kvh * a.group + qb * 16;
    const int64_t row0 = (static_cast<int64_t>(b) * a.hq + h0) * a.max_kv_splits;
    merge_splits_in_kernel<T>(a.attn_logits + row0 * D, a.attn_lse + row0, min(16, a.group - qb * 16), D, live,
                              a.max_kv_splits, a.sinks ? a.sinks + h0 : nullptr, a.v_scale,
                              a.o + b * a.o_stride_t + h0 * a.o_stride_h, a.o_stride_h);
#ifdef RX_DEC_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < kTlMax) g_dec_timeline[blockIdx.x * 6 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
  }
}

template <typename T, int D, typename IdxT, bool LINEAR, bool KV8, bool FUSE = false, bool OCC3 = false>
__global__ __launch_bounds__(256, OCC3 ? 3 : RX_DEC_MINW) void decode_mfma_kernel(const DecodeArgs a) {
  decode_mfma_body<T, D, IdxT, LINEAR, KV8, FUSE, OCC3, false>(a, blockIdx.x);
}
// (A RESIDENT form of this kernel -- at most two workgroups per CU, each walking (request, split) units handed out by a
// device-side queue -- was built and measured in round 6 and removed: without the next unit's first tiles requested under the
// current unit's tail, every unit pays its whole prologue and epilogue, and finer pieces cost more than the balance buys.
// TP = 8 shard, 256 x 4 k, us per launch, queue-fed / hardware-dispatched grid: 1 split 106.7 / 104.0, 4 splits 112.9 / 111.0,
// 8 splits 116.5 / 113.2; configs[1] 8 splits 131.4 / 131.8.  DESIGN 4.1 has the per-workgroup timeline behind it.)
template <typename T, int D, typename IdxT, bool LINEAR>
__global__ __launch_bounds__(256, RX_DEC_MINW) void decode_mfma_bias_kernel(const DecodeArgs a) {
  decode_mfma_body<T, D, IdxT, LINEAR, false, false, false, true>(a, blockIdx.x);
}

// ---- generic fallback: any head dims (Dk != Dv, 13, 80, 96, 576/512 ...) --------------------
// One wave per (request, q head, split); lane = token for QK^T, lane = d for PV.  Correctness
// path for shapes outside the MFMA kernel; same outputs / scratch layout.
template <typename T, typename IdxT, bool LINEAR>
__global__ __launch_bounds__(64) void decode_generic_kernel(const DecodeArgs a, int dk, int dv) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  float* qs = reinterpret_cast<float*>(dyn_smem);  // [dk]
  const int lane = threadIdx.x;
  int bid = blockIdx.x;  // split slowest (see decode_mfma_kernel)
  const int h = bid % a.hq;
  bid /= a.hq;
  const int b = bid % a.bs;
  const int split = bid / a.bs;
  const int kvh = h / a.group;
  const SeqInfo si = seq_info<IdxT>(a, b);
  const IdxT* idx = reinterpret_cast<const IdxT*>(si.idx);
  const int32_t splits = (a.num_kv_splits && a.max_kv_splits > 1) ? a.num_kv_splits[b] : 1;
  const bool single = (a.max_kv_splits == 1);
  if (split >= splits) return;
  int32_t lo, hi;
  split_range(si.seq_len, splits, split, lo, hi);
  if (hi <= lo) {
    if (single && si.seq_len == 0)
      for (int d = lane; d < dv; d += 64) a.o[b * a.o_stride_t + h * a.o_stride_h + d] = 0;
    return;
  }
  for (int d = lane; d < dk; d += 64)
    qs[d] = T::to_f32(a.q[b * a.q_stride_t + h * a.q_stride_h + d]);
  __syncthreads();
  constexpr int MAXV = 8;  // dv <= 512
  float acc[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) acc[j] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int base = lo; base < hi; base += 64) {
    const int tok = base + lane;
    const bool valid = tok < hi;
    const int64_t slot = static_cast<int64_t>(idx[valid ? tok : hi - 1]);
    const uint16_t* kp = a.k_buf + kvh * a.k_head_stride +
                         slot_offset<LINEAR>(slot, a.page_size, a.k_page_stride, a.k_tok_stride);
    float s = 0.f;
    for (int d = 0; d < dk; ++d) s += qs[d] * T::to_f32(kp[d]);
    s *= a.sm_scale;
    if (a.logit_cap > 0.f) s = a.logit_cap * tanhf(s / a.logit_cap);
    s *= xai_factor(a.xai_len, si.full_len);
    if (a.bias) {  // score_mod.py:44-56 through decode_attention.py:215-227
      const int32_t rel = si.seq_len - 1 - tok;
      if (valid && rel < a.bias_len)
        s += load_bias<T>(static_cast<const char*>(a.bias) + (b * a.bias_stride_t + h * a.bias_stride_h) * (a.bias_f32 ? 4 : 2), a.bias_f32, rel);
    }
    s = valid ? s * kLog2e : -INFINITY;
    float mt = s;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) mt = fmaxf(mt, __shfl_xor(mt, dd));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = fast_exp2(m_run - m_new);
    const float p = fast_exp2(s - m_new);
    float ps = p;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) ps += __shfl_xor(ps, dd);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) acc[j] *= alpha;
    const int nvalid = min(64, hi - base);
    const int64_t voff =
        slot_offset<LINEAR>(slot, a.page_size, a.v_page_stride, a.v_tok_stride);
    for (int j = 0; j < nvalid; ++j) {
      const float pj = __shfl(p, j);
      const int64_t vo = __shfl(voff, j);
      const uint16_t* vp = a.v_buf + kvh * a.v_head_stride + vo;
#pragma unroll
      for (int c = 0; c < MAXV; ++c) {
        const int d = lane + 64 * c;
        if (d < dv) acc[c] += pj * T::to_f32(vp[d]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < MAXV; ++c) {
    const int d = lane + 64 * c;
    if (d >= dv) continue;
    if (single) {
      float den = l_run;
      if (a.sinks) den += fast_exp2(a.sinks[h] * kLog2e - m_run);
      a.o[b * a.o_stride_t + h * a.o_stride_h + d] = T::from_f32(acc[c] / den * a.v_scale);
    } else {
      const int64_t row = (static_cast<int64_t>(b) * a.hq + h) * a.max_kv_splits + split;
      a.attn_logits[row * dv + d] = acc[c] / l_run;
      if (d == 0) a.attn_lse[row] = m_run * kLn2 + __logf(l_run);
    }
  }
}

// ---- stage 2: merge the kv splits (decode_attention.py:731-805) --------------------------------
// One thread = four consecutive output columns of one (request, head).  The split weights come from
// the LSEs alone (m = max_s lse_s, w_s = exp(lse_s - m)), so the partial rows are read by
// independent 16-byte loads with nothing but an FMA between them: HBM-bound instead of a dependent
// load -> exp -> rescale chain per split.  Same value as the reference's running rescale up to fp32
// rounding.
template <typename T>
__global__ __launch_bounds__(256) void decode_merge_kernel(const DecodeArgs a, int dv) {
  const int dv4 = dv >> 2;
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t bh = gid / dv4;
  if (bh >= static_cast<int64_t>(a.bs) * a.hq) return;
  const int d = static_cast<int>(gid % dv4) * 4;
  const int h = static_cast<int>(bh % a.hq);
  const int b = static_cast<int>(bh / a.hq);
  int32_t kv_begin;
  const int32_t seq_len = attended_len(a, b, kv_begin);
  const int32_t splits = a.num_kv_splits ? a.num_kv_splits[b] : 1;
  if (a.direct_single && splits == 1) return;  // stage 1 wrote this request's final output
  // live splits are a prefix: split s covers [per s, min(per (s+1), seq_len))
  const int32_t per = ((seq_len + splits - 1) / splits + kMinBlockKV - 1) / kMinBlockKV * kMinBlockKV;
  int32_t live = per > 0 ? (seq_len + per - 1) / per : 0;
  live = min(live, min(splits, a.max_kv_splits));
  const int64_t row0 = bh * a.max_kv_splits;
  const float* lse = a.attn_lse + row0;
  float e_max = -INFINITY;
  for (int s = 0; s < live; ++s) e_max = fmaxf(e_max, lse[s]);
  const int64_t xstride = static_cast<int64_t>(a.extra_rows) * a.hq;  // one extra partial = [extra_rows, hq] rows
  const int xb = a.extra_index ? a.extra_index[b] : b;
  const int64_t xbh = static_cast<int64_t>(xb) * a.hq + h;
  const int nx = xb < 0 ? 0 : a.num_extra;  // not in any shared-prefix group: no extras
  for (int x = 0; x < nx; ++x) e_max = fmaxf(e_max, a.extra_lse[x * xstride + xbh]);
  float e_sum = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* lp = a.attn_logits + row0 * dv + d;
#pragma unroll 4
  for (int s = 0; s < live; ++s) {
    const float w = __expf(lse[s] - e_max);
    const f32x4 tv = *reinterpret_cast<const f32x4*>(lp + static_cast<int64_t>(s) * dv);
    acc += w * tv;
    e_sum += w;
  }
  for (int x = 0; x < nx; ++x) {  // 16-bit partials of another pass (shared-prefix phase)
    const float xl = a.extra_lse[x * xstride + xbh];
    if (!(xl > -INFINITY)) continue;  // empty partial: its row is undefined
    const float w = __expf(xl - e_max);
    const u32x2 raw = *reinterpret_cast<const u32x2*>(a.extra_o + (x * xstride + xbh) * dv + d);
    acc += w * f32x4{T::to_f32(static_cast<uint16_t>(raw[0] & 0xffff)), T::to_f32(static_cast<uint16_t>(raw[0] >> 16)),
                     T::to_f32(static_cast<uint16_t>(raw[1] & 0xffff)), T::to_f32(static_cast<uint16_t>(raw[1] >> 16))};
    e_sum += w;
  }
  if (a.sinks) e_sum += __expf(a.sinks[h] - e_max);
  const float inv = a.v_scale / e_sum;
  u32x2 pk;
  pk[0] = pack2<T>(acc[0] * inv, acc[1] * inv);
  pk[1] = pack2<T>(acc[2] * inv, acc[3] * inv);
  *reinterpret_cast<u32x2*>(a.o + b * a.o_stride_t + h * a.o_stride_h + d) = pk;
}

// any v_head_dim / output alignment: one thread per column, the reference's running rescale as written
template <typename T>
__global__ __launch_bounds__(128) void decode_merge_scalar_kernel(const DecodeArgs a, int dv) {
  const int h = blockIdx.x % a.hq;
  const int b = blockIdx.x / a.hq;
  int32_t kv_begin;
  const int32_t seq_len = attended_len(a, b, kv_begin);
  const int32_t splits = a.num_kv_splits ? a.num_kv_splits[b] : 1;
  if (a.direct_single && splits == 1) return;  // stage 1 wrote this request's final output
  const int64_t row0 = (static_cast<int64_t>(b) * a.hq + h) * a.max_kv_splits;
  for (int d = threadIdx.x; d < dv; d += 128) {
    float e_sum = 0.f, e_max = -INFINITY, acc = 0.f;
    for (int s = 0; s < a.max_kv_splits; ++s) {
      int32_t lo, hi;
      split_range(seq_len, splits, s, lo, hi);
      if (s < splits && hi > lo) {
        const float tv = a.attn_logits[(row0 + s) * dv + d];
        const float tl = a.attn_lse[row0 + s];
        const float n_max = fmaxf(tl, e_max);
        const float old_scale = __expf(e_max - n_max);
        const float w = __expf(tl - n_max);
        acc = acc * old_scale + w * tv;
        e_sum = e_sum * old_scale + w;
        e_max = n_max;
      }
    }
    const int xb = a.extra_index ? a.extra_index[b] : b;
    for (int x = 0; x < (xb < 0 ? 0 : a.num_extra); ++x) {
      const int64_t xrow = (static_cast<int64_t>(x) * a.extra_rows + xb) * a.hq + h;
      const float tl = a.extra_lse[xrow];
      if (!(tl > -INFINITY)) continue;
      const float tv = T::to_f32(a.extra_o[xrow * dv + d]);
      const float n_max = fmaxf(tl, e_max);
      const float old_scale = __expf(e_max - n_max);
      const float w = __expf(tl - n_max);
      acc = acc * old_scale + w * tv;
      e_sum = e_sum * old_scale + w;
      e_max = n_max;
    }
    if (a.sinks) e_sum += __expf(a.sinks[h] - e_max);
    a.o[b * a.o_stride_t + h * a.o_stride_h + d] = T::from_f32(acc / e_sum * a.v_scale);
  }
}

// Eight split slots, no extra partials (the MLA config-5 launch: 17 MB of partials, 6 us): everything the thread needs --
// the request's length and split count, its 8 LSEs (two 16-byte loads) and its 8 partial rows -- is loaded AT ONCE, dead
// slots included (selected away, never multiplied: their memory is allocated but undefined), so the kernel is one round
// trip plus the store instead of four dependent ones (length -> LSEs -> rows in batches of four -> store).
template <typename T>
__global__ __launch_bounds__(256) void decode_merge8_kernel(const DecodeArgs a, int dv) {
  const int dv4 = dv >> 2;
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t bh = gid / dv4;
  if (bh >= static_cast<int64_t>(a.bs) * a.hq) return;
  const int d = static_cast<int>(gid % dv4) * 4;
  const int h = static_cast<int>(bh % a.hq);
  const int b = static_cast<int>(bh / a.hq);
  const int64_t row0 = bh * 8;
  const f32x4 l0 = *reinterpret_cast<const f32x4*>(a.attn_lse + row0), l1 = *reinterpret_cast<const f32x4*>(a.attn_lse + row0 + 4);
  const float* lp = a.attn_logits + row0 * dv + d;
  f32x4 tv[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) tv[s] = *reinterpret_cast<const f32x4*>(lp + static_cast<int64_t>(s) * dv);
  int32_t kv_begin;
  const int32_t seq_len = attended_len(a, b, kv_begin);
  const int32_t splits = a.num_kv_splits ? a.num_kv_splits[b] : 1;
  if (a.direct_single && splits == 1) return;  // stage 1 wrote this request's final output
  const int32_t per = ((seq_len + splits - 1) / splits + kMinBlockKV - 1) / kMinBlockKV * kMinBlockKV;
  int32_t live = per > 0 ? (seq_len + per - 1) / per : 0;
  live = min(live, min(splits, 8));
  const float lse[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  float e_max = -INFINITY;
#pragma unroll
  for (int s = 0; s < 8; ++s) e_max = s < live ? fmaxf(e_max, lse[s]) : e_max;
  float e_sum = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float w = s < live ? __expf(lse[s] - e_max) : 0.f;
    const f32x4 row = s < live ? tv[s] : f32x4{0.f, 0.f, 0.f, 0.f};
    acc += w * row;
    e_sum += w;
  }
  if (a.sinks) e_sum += __expf(a.sinks[h] - e_max);
  const float inv = a.v_scale / e_sum;
  u32x2 pk;
  pk[0] = pack2<T>(acc[0] * inv, acc[1] * inv);
  pk[1] = pack2<T>(acc[2] * inv, acc[3] * inv);
  *reinterpret_cast<u32x2*>(a.o + b * a.o_stride_t + h * a.o_stride_h + d) = pk;
}

template <typename T>
static void launch_merge(const DecodeArgs& a, int dv, hipStream_t s) {
  const bool vec = dv % 4 == 0 && ((a.o_stride_t | a.o_stride_h) & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.o) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.extra_o) & 7) == 0;
  if (vec) {
    const unsigned grid = static_cast<unsigned>((static_cast<int64_t>(a.bs) * a.hq * (dv >> 2) + 255) / 256);
    if (a.max_kv_splits == 8 && a.num_extra == 0 && (reinterpret_cast<uintptr_t>(a.attn_lse) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(a.attn_logits) & 15) == 0) {
      hipLaunchKernelGGL((decode_merge8_kernel<T>), dim3(grid), dim3(256), 0, s, a, dv);
      return;
    }
    hipLaunchKernelGGL((decode_merge_kernel<T>), dim3(grid), dim3(256), 0, s, a, dv);
  } else {
    hipLaunchKernelGGL((decode_merge_scalar_kernel<T>), dim3(a.bs * a.hq), dim3(128), 0, s, a, dv);
  }
}

template <typename T, typename IdxT, bool LINEAR>
static int launch_decode(const DecodeArgs& a, int dk, int dv, hipStream_t s) {
  const bool mfma_ok = (dk == dv) && (dk == 64 || dk == 128 || ((dk == 256 || dk == 96) && !a.kv_fp8 && !a.bias));
  if (a.stages == 2) {
    launch_merge<T>(a, dv, s);
    return check_launch("rx_decode_attn");
  }
  if (mfma_ok) {
    const unsigned grid = a.items ? static_cast<unsigned>(a.items_cap) * a.hkv * a.qblocks
                                  : static_cast<unsigned>(a.bs) * a.hkv * a.qblocks * a.max_kv_splits;
    if (grid == 0) return RX_OK;
    {
      const bool occ3 = a.items && a.items_occ3 && dk == 128 && !a.kv_fp8;
      note_dispatch("decode_mfma_kernel<%s, %d, %s, %s, %s, %s, %s>|%s,%s", tname<T>(), dk, tname<IdxT>(), tbool(LINEAR),
                    tbool(a.kv_fp8 != 0), tbool(a.k_new != nullptr && !a.kv_fp8), tbool(occ3), a.items ? "pairs" : "slots",
                    a.kv_indices ? "indices" : "req_to_token");
    }
#define RX_DEC(DD, K8, FU) \
  hipLaunchKernelGGL((decode_mfma_kernel<T, DD, IdxT, LINEAR, K8, FU>), dim3(grid), dim3(256), 0, s, a)
    if (a.bias) {  // (mfma_ok: D = 64 / 128 on a 16-bit pool without the fused store)
      note_dispatch("decode_mfma_bias_kernel<%s, %d, %s, %s>|%s,%s", tname<T>(), dk, tname<IdxT>(), tbool(LINEAR),
                    a.items ? "pairs" : "slots", a.kv_indices ? "indices" : "req_to_token");
      if (dk == 64) hipLaunchKernelGGL((decode_mfma_bias_kernel<T, 64, IdxT, LINEAR>), dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((decode_mfma_bias_kernel<T, 128, IdxT, LINEAR>), dim3(grid), dim3(256), 0, s, a);
    } else if (a.items && a.items_occ3 && dk == 128 && !a.kv_fp8) {  // a mixed batch's schedule: three workgroups per CU
      if (a.k_new) hipLaunchKernelGGL((decode_mfma_kernel<T, 128, IdxT, LINEAR, false, true, true>), dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((decode_mfma_kernel<T, 128, IdxT, LINEAR, false, false, true>), dim3(grid), dim3(256), 0, s, a);
    } else if (a.kv_fp8) {
      if (dk == 64) RX_DEC(64, true, false);
      else RX_DEC(128, true, false);
    } else if (a.k_new) {  // fused store of the new token
      if (dk == 64) RX_DEC(64, false, true);
      else RX_DEC(128, false, true);
    } else {
      if (dk == 64) RX_DEC(64, false, false);
      else if (dk == 128) RX_DEC(128, false, false);
      else if (dk == 96) RX_DEC(96, false, false);  // Phi-3-class heads
      else RX_DEC(256, false, false);  // Gemma-class heads: the same kernel, 64 accumulator registers per wave
    }
#undef RX_DEC
  } else {
    if (a.kv_fp8)
      return fail(RX_ERR_UNSUPPORTED, "rx_decode_attn: fp8 KV pools need head_dim 64/128 (or the MLA shape), got %d/%d",
                  dk, dv);
    if (dv > 512) return fail(RX_ERR_UNSUPPORTED, "rx_decode_attn: v_head_dim %d > 512", dv);
    const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.max_kv_splits;
    note_dispatch("decode_generic_kernel<%s, %s, %s>|dk%d,dv%d", tname<T>(), tname<IdxT>(), tbool(LINEAR), dk, dv);
    hipLaunchKernelGGL((decode_generic_kernel<T, IdxT, LINEAR>), dim3(grid), dim3(64),
                       dk * sizeof(float), s, a, dk, dv);
  }
  if (a.max_kv_splits > 1 && a.stages != 1 && !(a.merge_counters && mfma_ok))
    launch_merge<T>(a, dv, s);
  return check_launch("rx_decode_attn");
}

template <typename T>
static int dispatch_decode(const DecodeArgs& a, int dk, int dv, bool idx64, bool linear,
                           hipStream_t s) {
  if (idx64) {
    return linear ? launch_decode<T, int64_t, true>(a, dk, dv, s)
                  : launch_decode<T, int64_t, false>(a, dk, dv, s);
  }
  return linear ? launch_decode<T, int32_t, true>(a, dk, dv, s)
                : launch_decode<T, int32_t, false>(a, dk, dv, s);
}

int launch_decode_mla(const rx_decode_params* p, int32_t* merge_counters, int direct_single, hipStream_t s);  // rx_decode_mla.hip

template <typename T>
static int run_mla(const rx_decode_params* p, const DecodeArgs& a, hipStream_t s) {
  if (a.stages != 2) {
    const int rc = launch_decode_mla(p, a.merge_counters, a.direct_single, s);
    if (rc != RX_OK) return rc;
  }
  if (a.max_kv_splits > 1 && a.stages != 1 && !a.merge_counters)
    launch_merge<T>(a, p->v_head_dim, s);
  return check_launch("rx_decode_attn(mla)");
}

}  // namespace rx

namespace rx {
// rx_decode_units: one workgroup of 128 threads per unit; thread j resolves the slot of token min(lo + j, hi - 1), thread 0
// writes the descriptor
__global__ __launch_bounds__(128) void decode_units_kernel(const int32_t* __restrict__ r2t, int64_t row_stride, const void* rpi, int rpi64,
                                                          const void* seq_lens, int sl64, const int32_t* __restrict__ nsplits,
                                                          int max_kv_splits, const int32_t* __restrict__ items,
                                                          const int32_t* __restrict__ count, const int32_t* __restrict__ order, int bs,
                                                          int32_t* __restrict__ desc, int32_t* __restrict__ first) {
  const int u = blockIdx.x, j = threadIdx.x;
  int b, split;
  if (items) {
    if (u >= count[0]) return;
    b = items[2 * u];
    split = items[2 * u + 1];
  } else {
    if (u >= bs) return;
    b = order ? order[u] : u;
    split = 0;
  }
  const int64_t req = load_idx(rpi, b, rpi64);
  const int32_t seq = static_cast<int32_t>(load_idx(seq_lens, b, sl64));
  const int32_t sp = (nsplits && max_kv_splits > 1) ? nsplits[b] : 1;
  int32_t lo = 0, hi = 0;
  if (sp > 0 && split < sp) split_range(seq, sp, split, lo, hi);
  const int64_t row = req * row_stride;
  first[128 * static_cast<int64_t>(u) + j] = hi > lo ? r2t[row + min(lo + j, hi - 1)] : 0;
  if (j == 0) {
    int32_t* d = desc + 8 * static_cast<int64_t>(u);
    d[0] = b;
    d[1] = split;
    d[2] = seq;
    d[3] = sp;
    d[4] = static_cast<int32_t>(static_cast<uint64_t>(row) & 0xffffffffu);
    d[5] = static_cast<int32_t>(static_cast<uint64_t>(row) >> 32);
    d[6] = lo;
    d[7] = hi;
  }
}
}  // namespace rx

using namespace rx;

extern "C" int rx_decode_units(const int32_t* req_to_token, int64_t req_row_stride, const void* req_pool_indices,
                               int req_pool_indices_is_i64, const void* seq_lens, int seq_lens_is_i64, const int32_t* num_kv_splits,
                               int max_kv_splits, const int32_t* split_items, const int32_t* split_items_count, int cap,
                               const int32_t* request_order, int bs, int32_t* unit_desc, int32_t* unit_first_slots, void* stream) {
  RX_RANGE("rx_decode_units");
  RX_REQUIRE(bs >= 0 && cap >= 0, "rx_decode_units: negative sizes");
  if (bs == 0 || cap == 0) return RX_OK;
  RX_REQUIRE(req_to_token && req_pool_indices && seq_lens && unit_desc && unit_first_slots, "rx_decode_units: null pointer");
  RX_REQUIRE((split_items == nullptr) == (split_items_count == nullptr), "rx_decode_units: split_items and its count come together");
  RX_REQUIRE(split_items || cap <= bs, "rx_decode_units: an unsplit step has bs units (cap %d > bs %d)", cap, bs);
  RX_REQUIRE((((uintptr_t)unit_desc | (uintptr_t)unit_first_slots) & 15) == 0, "rx_decode_units: tables must be 16-byte aligned");
  hipLaunchKernelGGL(rx::decode_units_kernel, dim3(static_cast<unsigned>(cap)), dim3(128), 0, static_cast<hipStream_t>(stream),
                     req_to_token, req_row_stride, req_pool_indices, req_pool_indices_is_i64, seq_lens, seq_lens_is_i64, num_kv_splits,
                     max_kv_splits, split_items, split_items_count, request_order, bs, unit_desc, unit_first_slots);
  return check_launch("rx_decode_units");
}

#ifdef RX_DEC_TIMELINE
// dev builds only (not in include/radix_hip.h): copies the first n workgroups' stamps to host memory, [n][6] =
// entry, first tile landed, loop end, epilogue end, xcc id, merge end (0 when the workgroup did not merge)
extern "C" int rx_dev_decode_timeline(unsigned long long* out_host, int n) {
  if (n > kTlMax) n = kTlMax;
  hipError_t e = hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_dec_timeline), sizeof(unsigned long long) * 6 * n);
  return e == hipSuccess ? 0 : -3;
}
#endif

static int decode_attn_impl(const rx_decode_params* p, void* stream);
extern "C" int rx_decode_attn(const rx_decode_params* p, void* stream) {
  RX_RANGE("rx_decode_attn");
  return rx::dump_on_error("decode_attn", decode_attn_impl(p, stream), p, p ? sizeof(*p) : 0);
}
static int decode_attn_impl(const rx_decode_params* p, void* stream) {
  RX_REQUIRE(p, "rx_decode_attn: params is null");
  RX_REQUIRE(p->bs >= 0, "rx_decode_attn: bs < 0");
  if (p->bs == 0) return RX_OK;
  RX_REQUIRE(p->q && p->o && p->kv.k_buf && p->kv.v_buf, "rx_decode_attn: null q/o/k_buf/v_buf");
  RX_REQUIRE(p->num_q_heads > 0 && p->num_kv_heads > 0 && p->num_q_heads % p->num_kv_heads == 0,
             "rx_decode_attn: Hq=%d must be a positive multiple of Hkv=%d", p->num_q_heads,
             p->num_kv_heads);
  RX_REQUIRE(p->head_dim > 0 && p->v_head_dim > 0, "rx_decode_attn: bad head dims");
  RX_REQUIRE(p->kv.page_size >= 1, "rx_decode_attn: page_size < 1");
  RX_REQUIRE(p->dtype == RX_BF16 || p->dtype == RX_F16, "rx_decode_attn: dtype %d", p->dtype);
  RX_REQUIRE(p->kv.kv_fp8 == 0 || p->kv.kv_fp8 == 1, "rx_decode_attn: kv_fp8 = %d", p->kv.kv_fp8);
  const bool mode_a = p->kv_indices != nullptr;
  if (mode_a) {
    RX_REQUIRE(p->kv_indptr, "rx_decode_attn: kv_indices given without kv_indptr");
  } else {
    RX_REQUIRE(p->req_to_token && p->req_pool_indices && p->seq_lens,
               "rx_decode_attn: neither kv_indices nor req_to_token/req_pool_indices/seq_lens");
  }
  const int max_splits = p->max_kv_splits < 1 ? 1 : p->max_kv_splits;
  if (max_splits > 1)
    RX_REQUIRE(p->attn_logits && p->attn_lse && p->num_kv_splits,
               "rx_decode_attn: max_kv_splits=%d needs attn_logits, attn_lse and num_kv_splits",
               max_splits);
  const int dk = p->head_dim, dv = p->v_head_dim;
  if (p->score_bias) {
    RX_REQUIRE(p->score_bias_len > 0, "rx_decode_attn: score_bias_len = %d", p->score_bias_len);
    RX_REQUIRE(((uintptr_t)p->score_bias & (p->score_bias_is_f32 ? 3 : 1)) == 0, "rx_decode_attn: misaligned score_bias");
    if (p->kv.kv_fp8 || p->k_new || p->v_new || p->rope_cos_sin || (p->head_dim == 576 && p->v_head_dim == 512))
      return fail(RX_ERR_UNSUPPORTED, "rx_decode_attn: score_bias needs a 16-bit non-latent pool and no fused store / RoPE");
  }
  const bool mfma_ok = (dk == dv) && (dk == 64 || dk == 128 || ((dk == 256 || dk == 96) && !p->kv.kv_fp8 && !p->score_bias));
  if (mfma_ok) {
    // 16-byte vector loads: every stride a multiple of 8 elements, bases 16-byte aligned
    const int64_t all = p->q_stride_t | p->q_stride_h | p->kv.k_page_stride | p->kv.k_tok_stride |
                        p->kv.k_head_stride | p->kv.v_page_stride | p->kv.v_tok_stride |
                        p->kv.v_head_stride;
    RX_REQUIRE(all % 8 == 0, "rx_decode_attn: q/kv strides must be multiples of 8 elements");
    RX_REQUIRE((((uintptr_t)p->q | (uintptr_t)p->kv.k_buf | (uintptr_t)p->kv.v_buf) & (p->kv.kv_fp8 ? 7 : 15)) == 0 &&
                   ((uintptr_t)p->q & 15) == 0,
               "rx_decode_attn: q must be 16-byte aligned, k_buf/v_buf 16-byte (fp8 pools: 8-byte)");
  }
  DecodeArgs a;
  a.q = (const uint16_t*)p->q;
  a.o = (uint16_t*)p->o;
  a.q_stride_t = p->q_stride_t;
  a.q_stride_h = p->q_stride_h;
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
  a.kv_indptr = p->kv_indptr;
  a.kv_indices = p->kv_indices;
  a.req_to_token = p->req_to_token;
  a.req_row_stride = p->req_row_stride;
  a.req_pool_indices = p->req_pool_indices;
  a.rpi64 = p->req_pool_indices_is_i64;
  a.seq_lens = p->seq_lens;
  a.sl64 = p->seq_lens_is_i64;
  a.num_kv_splits = max_splits > 1 ? p->num_kv_splits : nullptr;
  a.max_kv_splits = max_splits;
  a.attn_logits = p->attn_logits;
  a.attn_lse = p->attn_lse;
  a.bs = p->bs;
  a.hq = p->num_q_heads;
  a.hkv = p->num_kv_heads;
  a.group = p->num_q_heads / p->num_kv_heads;
  a.qblocks = (a.group + 15) / 16;
  a.sm_scale = p->sm_scale * p->k_scale;
  a.v_scale = p->v_scale;
  a.logit_cap = p->logit_cap;
  a.sinks = p->sinks;
  a.kv_fp8 = p->kv.kv_fp8;
  a.xai_len = p->xai_temperature_len;
  a.bias = p->score_bias;
  a.bias_f32 = p->score_bias_is_f32;
  a.bias_len = p->score_bias_len;
  a.bias_stride_t = p->score_bias_stride_t;
  a.bias_stride_h = p->score_bias_stride_h;
  a.desc = a.first = nullptr;
  RX_REQUIRE(p->stages >= 0 && p->stages <= 2 && (p->stages == 0 || max_splits > 1),
             "rx_decode_attn: stages = %d (1 / 2 need max_kv_splits > 1)", p->stages);
  a.stages = p->stages;
  a.kv_start = mode_a ? nullptr : p->kv_start;
  a.num_extra = p->num_extra_partials > 0 ? p->num_extra_partials : 0;
  a.extra_o = a.num_extra ? (const uint16_t*)p->extra_o : nullptr;
  a.extra_lse = a.num_extra ? p->extra_lse : nullptr;
  a.extra_index = a.num_extra ? p->extra_index : nullptr;
  a.extra_rows = (a.num_extra && p->extra_index && p->extra_rows > 0) ? p->extra_rows : p->bs;
  if (a.num_extra)
    RX_REQUIRE(p->extra_o && p->extra_lse && (max_splits > 1 || mfma_ok),
               "rx_decode_attn: extra partials need extra_o, extra_lse and either max_kv_splits > 1 (stage 2 merges "
               "them) or the D = 64 / 128 kernel (its single-pass epilogue folds them in)");
  RX_REQUIRE(!(mode_a && p->kv_start), "rx_decode_attn: kv_start applies to the req_to_token lookup only "
             "(with kv_indices, build the list with rx_build_kv_indices' kv_start)");
  const bool linear = p->kv.page_size == 1 ||
                      (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                       p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  const bool idx64 = mode_a && p->kv_indices_is_i64;
  auto s = static_cast<hipStream_t>(stream);
  // MLA latent rows: Dk 576 / Dv 512, one kv head, V = the first 512 columns of the K rows
  const bool mla = dk == 576 && dv == 512 && p->num_kv_heads == 1 && p->kv.v_buf == p->kv.k_buf &&
                   p->kv.v_tok_stride == p->kv.k_tok_stride && p->kv.v_page_stride == p->kv.k_page_stride &&
                   ((p->q_stride_t | p->q_stride_h | p->kv.k_tok_stride | p->kv.k_page_stride) % 8 == 0) &&
                   ((uintptr_t)p->q & 15) == 0 && ((uintptr_t)p->kv.k_buf & (p->kv.kv_fp8 ? 7 : 15)) == 0 &&
                   ((p->o_stride_t | p->o_stride_h) % 4 == 0) && ((uintptr_t)p->o & 7) == 0;
  // stage 2 inside the stage-1 kernel: the kernels that have the epilogue, both stages wanted, nothing else to merge,
  // and the vector stores of the merge possible
  const bool merge_in_kernel = p->merge_counters && p->stages == 0 && a.num_extra == 0 && max_splits > 1 &&
                               (mfma_ok || mla) && p->o_stride_t % 4 == 0 && p->o_stride_h % 4 == 0 &&
                               ((uintptr_t)p->o & 7) == 0 && ((uintptr_t)p->attn_logits & 15) == 0 &&
                               max_splits % 8 == 0 && ((uintptr_t)p->attn_lse & 15) == 0 &&  // 8 LSEs = two 16-B loads
                               // the partials take device-scope (write-through) stores in this form: a win while they
                               // are small (bs 1 x 32 k: 39 -> 36 us per layer, 16 x 4 k: 56 -> 53), a loss once they
                               // are many MB (MLA 64 x 16 heads x 8 splits = 17 MB: 67.6 -> 69.5 us)
                               // (partial_pairs_hint: how many (request, split) pairs really write a partial, when
                               // the caller's schedule knows better than bs * max_kv_splits)
                               // (a split-items table bounds them the same way: its capacity >= the live pairs)
                               static_cast<int64_t>((p->partial_pairs_hint > 0 || p->split_items)
                                                        ? min(static_cast<int64_t>(p->partial_pairs_hint > 0 ? p->partial_pairs_hint
                                                                                                           : p->split_items_cap),
                                                              static_cast<int64_t>(p->bs) * max_splits)
                                                        : static_cast<int64_t>(p->bs) * max_splits) *
                                       p->num_q_heads * dv * 4 <=
                                   (static_cast<int64_t>(mla ? options().merge_in_kernel_max_mb_mla : options().merge_in_kernel_max_mb) << 20);
  a.merge_counters = merge_in_kernel ? p->merge_counters : nullptr;
  // fused store of the new token: one q block per kv head (one workgroup touches the row), a 16-bit pool, the whole
  // request attended in one pass of the MFMA kernel, 16-byte chunks
  a.order = p->request_order;
  a.items = p->split_items;
  a.items_count = p->split_items_count;
  a.items_cap = p->split_items_cap;
  a.items_occ3 = p->split_items_wgs_per_cu == 3 ? 1 : 0;
  if (a.items) {
    RX_REQUIRE(a.items_count && a.items_cap >= 0, "rx_decode_attn: split_items without its count / cap");
    RX_REQUIRE(p->num_kv_splits && p->max_kv_splits > 1, "rx_decode_attn: split_items go with a split schedule (num_kv_splits, max_kv_splits > 1)");
  }
  a.direct_single = ((mfma_ok || mla) && a.num_extra == 0 && max_splits > 1 && p->stages == 0) ? 1 : 0;
  a.k_new = a.v_new = nullptr;
  if (mla) {
    // the latent kernel takes k_new (the whole new row, v is its prefix) only with fused RoPE: checked in launch_decode_mla
  } else if (p->k_new || p->v_new) {
    RX_REQUIRE(p->k_new && p->v_new, "rx_decode_attn: k_new and v_new come together");
    RX_REQUIRE(mfma_ok && dk <= 128 && dk != 96 && !mla && !p->kv.kv_fp8 && a.qblocks == 1 && p->stages != 2,
               "rx_decode_attn: the fused store needs the D = 64 / 128 kernel on a 16-bit pool with at most 16 q heads "
               "per kv head, in a call that runs stage 1 (store with rx_store_kv* instead)");
    RX_REQUIRE((((uintptr_t)p->k_new | (uintptr_t)p->v_new) & 15) == 0 &&
                   (p->k_new_stride_t | p->k_new_stride_h | p->v_new_stride_t | p->v_new_stride_h) % 8 == 0,
               "rx_decode_attn: k_new / v_new need 16-byte aligned rows");
    a.k_new = (const uint16_t*)p->k_new;
    a.v_new = (const uint16_t*)p->v_new;
    a.kn_stride_t = p->k_new_stride_t; a.kn_stride_h = p->k_new_stride_h;
    a.vn_stride_t = p->v_new_stride_t; a.vn_stride_h = p->v_new_stride_h;
  }
  // per-unit tables: req_to_token mode of the MFMA kernel, whole requests or the live-pairs grid, no kv_start
  if (p->unit_desc || p->unit_first_slots) {
    RX_REQUIRE(p->unit_desc && p->unit_first_slots, "rx_decode_attn: unit_desc and unit_first_slots come together");
    RX_REQUIRE((((uintptr_t)p->unit_desc | (uintptr_t)p->unit_first_slots) & 15) == 0, "rx_decode_attn: unit tables must be 16-byte aligned");
    if (!mode_a && mfma_ok && !mla && !p->kv_start && p->stages != 2 && (a.items || max_splits == 1)) {
      a.desc = p->unit_desc;
      a.first = p->unit_first_slots;
    }
  }
  if (mla) return p->dtype == RX_BF16 ? run_mla<BF16>(p, a, s) : run_mla<F16>(p, a, s);
  return p->dtype == RX_BF16 ? dispatch_decode<BF16>(a, dk, dv, idx64, linear, s)
                             : dispatch_decode<F16>(a, dk, dv, idx64, linear, s);
}
