// MLA (absorbed) decode attention for gfx950: Dk = 576 (512 latent + 64 rope), Dv = 512, Hkv = 1,
// V = the first 512 columns of the SAME latent rows.
//
// Reference: the HAS_MLA branch of _fwd_grouped_kernel_stage1
// (kernels/ops/attention/decode_attention.py:383-608, v = trans(k) at :556-557, BLOCK_DMODEL=512 /
// BLOCK_DPE=64 at :638-646), driven by TritonAttnBackend.forward_decode
// (srt/layers/attention/triton_backend.py:1739-1757,1858) over MLATokenToKVPool
// (srt/mem_cache/memory_pool.py:3906-4179).  Same split-KV scratch contract and stage 2 as the
// dense path (rx_decode.hip).
//
// MI355X design.  A latent row is 1152 B (bf16) and is needed twice: all 576 columns for q.k and
// the first 512 for p.v.  One workgroup (4 waves) = one (request, 16-head q block, kv split):
//   * 32-token tiles are staged ONCE per workgroup: whole 1152-B rows, coalesced 16 B/lane,
//     HBM -> registers (issued one tile ahead) -> LDS (double buffered, one barrier per tile).
//     Every HBM byte is read once and serves both products.
//   * the 16 q heads sit on the N axis of v_mfma_f32_16x16x32 -- MLA's 16 heads per GPU at TP=8
//     fill the tile exactly.  Each wave computes the full S^T = K Q^T (18 k-steps x 2 token blocks;
//     redundant across the 4 waves, but MFMA is ~10 % busy in this HBM-bound kernel and it makes
//     the softmax statistics identical in every wave: no cross-wave exchange at all).
//   * each wave owns one 128-column slice of Dv: O^T[128w..128w+128) += V^T P^T with V^T fragments
//     read transposed (ds_read_b64_tr_b16) straight out of the staged K rows.
// Rows are padded to 1168 B in LDS (stride = 73 x 16 B, odd in 16-B units) to spread the banks.
#include "rx_common.h"

namespace rx {

struct MlaArgs {
  const uint16_t* q;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, o_stride_t, o_stride_h;
  const uint16_t* kv_buf;
  int32_t page_size;  // encoded like DecodeArgs (negative = power of two)
  int64_t page_stride, tok_stride;
  const int32_t* kv_indptr;
  const void* kv_indices;
  const int32_t* req_to_token;
  const int32_t* kv_start;  // req_to_token lookup only: request b attends [kv_start[b], seq_len_b) (or NULL)
  int64_t req_row_stride;
  const void* req_pool_indices;
  int32_t rpi64;
  const void* seq_lens;
  int32_t sl64;
  const int32_t* num_kv_splits;
  int32_t max_kv_splits;
  float* attn_logits;
  float* attn_lse;
  int32_t bs, hq, qblocks;
  float sm_scale, v_scale, logit_cap;
  const float* sinks;
  int32_t xai_len;
  int32_t* merge_counters;  // in-kernel stage 2 (rx_common.h split_arrive_is_last), or NULL
  int32_t direct_single;    // a request with one kv split writes its final output from stage 1 (see rx_decode.hip)
  // fused RoPE (rx_decode_params.rope_*; decode_mla_kernel on 16-bit rows): q_pe rotated in registers, the newest
  // token's k_pe rotated on its way into the staged tile
  const void* rope_cache;   // [max_pos, 64]: cos (32) | sin (32) per position; NULL = off
  int32_t rope_f32;         // the table is fp32 (else the call's 16-bit dtype)
  int64_t rope_stride;      // elements per position row
  const void* rope_pos;     // [bs]
  int32_t rope_pos64, rope_neox;
  uint16_t* kpe_out;        // [bs, 64] rotated k_pe of every request's newest token, or NULL
  int64_t kpe_out_stride;
  const uint16_t* k_new;    // [bs, 576] the step's new latent rows (k_pe NOT rotated), or NULL: the pool holds them
  int64_t k_new_stride;
};

#ifndef RX_MLA_PD
#define RX_MLA_PD 4  // K-fragment prefetch distance (LDS reads in flight ahead of the MFMA)
#endif
typedef int v2i_t __attribute__((ext_vector_type(2)));
constexpr int kMlaDk = 576, kMlaDv = 512;
constexpr int kMlaTile = 32;
constexpr float kMlaSumLimit = 4096.0f;  // a lane's partial row sum above this sends the wave to the max-based softmax step

#ifndef RX_MLA_MERGE_R
#define RX_MLA_MERGE_R 4  // output chunks per thread and device-scope round trip of the in-kernel stage 2 (rx_common.h)
#endif
// stage 2 inside the kernel: called by every thread of a workgroup that wrote a partial of (request b, q block qb)
template <typename T>
__device__ __forceinline__ void mla_merge_if_last(const MlaArgs& a, int b, int qb, int32_t seq_len, int32_t splits) {
  const int32_t per = ((seq_len + splits - 1) / splits + 31) / 32 * 32;
  const int32_t live = min((seq_len + per - 1) / per, min(splits, a.max_kv_splits));
  if (!split_arrive_is_last(a.merge_counters + b * a.qblocks + qb, live)) return;
  const int h0 = qb * 16;
  const int64_t row0 = (static_cast<int64_t>(b) * a.hq + h0) * a.max_kv_splits;
  merge_splits_in_kernel<T, RX_MLA_MERGE_R>(a.attn_logits + row0 * kMlaDv, a.attn_lse + row0, min(16, a.hq - h0), kMlaDv, live,
                            a.max_kv_splits, a.sinks ? a.sinks + h0 : nullptr, a.v_scale,
                            a.o + b * a.o_stride_t + h0 * a.o_stride_h, a.o_stride_h);
}

constexpr int kMlaRowBytes = kMlaDk * 2;          // 1152
// padded LDS row stride: TWO pad chunks.  ds_read_b128 is served in four non-contiguous 16-lane groups (MI355X_MICROARCH.md,
// LDS) that mix rows 0-3 / 12-15 of k-group g with rows 4-11 of g + 1: with one pad chunk (1168 B, rounds 1-2) two of
// them share a bank, and so do two rows of a transposed V read -- both took twice their cycles (SQ_LDS_BANK_CONFLICT
// 0.29 of the LDS cycles); 1184 B is conflict-free for both.
constexpr int kMlaLdsRow = kMlaRowBytes + 32;
constexpr int kMlaChunks = kMlaRowBytes / 16;     // 72 16-byte chunks per row

template <bool LINEAR>
__device__ __forceinline__ int64_t mla_slot_off(int64_t slot, int32_t page_size, int64_t page_stride,
                                                int64_t tok_stride) {
  if constexpr (LINEAR) return mul_u32(slot, tok_stride);
  if (page_size < 0) {
    const int sh = -page_size - 1;
    return mul_u32(slot >> sh, page_stride) + mul_u32(slot & ((1 << sh) - 1), tok_stride);
  }
  return (slot / page_size) * page_stride + (slot % page_size) * tok_stride;
}

// KV8: the latent rows are fp8 e4m3fn (576 B); they are upcast (exact) while being staged, so the LDS
// image and everything after it is the 16-bit kernel's.
// fp8 rows carry half the bytes per tile through the same per-tile compute path, so that path -- a chain of
// latency-bound steps (LDS round trips, the score exchange, the softmax dependency chain) at two waves per SIMD --
// sets the rate, not HBM (round 1: 4.0-4.3 TB/s against 5.1 for 16-bit rows).  The fp8 instance therefore keeps ONE
// staged tile in LDS instead of two (41 KB per workgroup) and runs THREE workgroups per CU: 50 % more rows in
// flight and a third wave per SIMD to fill the bubbles, for one more barrier per tile.
template <bool KV8>
struct MlaBuf {
  static constexpr int NBUF = KV8 ? 1 : 2;
  static constexpr int WGS = NBUF == 1 ? 3 : 2;  // workgroups per CU (= waves per SIMD) to allocate registers for
};

template <typename T, typename IdxT, bool LINEAR, bool KV8>
__global__ __launch_bounds__(256, MlaBuf<KV8>::WGS) void decode_mla_kernel(const MlaArgs a) {
  constexpr int NBUF = MlaBuf<KV8>::NBUF;
  using vec8 = typename T::vec8;
  using KvE = std::conditional_t<KV8, uint8_t, uint16_t>;
  using KvV = u32x4;  // 16 B per lane and load: 8 elements of a 16-bit row, 16 of an fp8 row
  constexpr int KS = kMlaDk / 32;       // 18 k-steps
  constexpr int NBW = kMlaDv / 16 / 4;  // 8 d-blocks of 16 per wave
  // two staged tiles + (split-S form) the 4 x 1 KiB exchange of partial score blocks
  __shared__ __attribute__((aligned(16))) char smem[NBUF * kMlaTile * kMlaLdsRow + 4 * 64 * 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // Block index -> (request, split, 16-head q block).  The q blocks of one (request, split) read the SAME
  // latent rows; workgroups are dealt round-robin to the 8 XCDs, so with the q block as the fastest index
  // (the plain order) the up-to-8 blocks that share rows land on 8 different L2s and the rows are fetched
  // from HBM up to 8 times (TP=1 DeepSeek: 128 heads = 8 q blocks).  Here the XCD is bid % 8, the
  // (request, split) pair is bound to that XCD and its q blocks follow each other on it.  Split slowest
  // (see rx_decode.hip).
  int qb, b, split;
  {
    const int G = a.qblocks;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qb = j % G;
    const int r = (j / G) * 8 + xcd;  // (request, split) pair
    if (r >= a.bs * a.max_kv_splits) return;
    b = r % a.bs;
    split = r / a.bs;
  }

  int32_t seq_len;
  const IdxT* idx;
  if (a.kv_indices) {
    const int32_t beg = a.kv_indptr[b];
    seq_len = a.kv_indptr[b + 1] - beg;
    idx = reinterpret_cast<const IdxT*>(a.kv_indices) + beg;
  } else {
    const int64_t req = load_idx(a.req_pool_indices, b, a.rpi64);
    seq_len = static_cast<int32_t>(load_idx(a.seq_lens, b, a.sl64));
    idx = reinterpret_cast<const IdxT*>(a.req_to_token + req * a.req_row_stride);
  }
  const int32_t full_len = seq_len;  // the query's position + 1 (Grok temperature)
  if (a.kv_start && !a.kv_indices) {  // the suffix [kv_start[b], seq_len) only (shared-prefix cascade, phase 2)
    const int32_t st = min(max(a.kv_start[b], 0), seq_len);
    idx += st;
    seq_len -= st;
  }
  const int32_t splits = (a.num_kv_splits && a.max_kv_splits > 1) ? a.num_kv_splits[b] : 1;
  const bool single = (a.max_kv_splits == 1) || (a.direct_single && splits == 1);
  const int h = qb * 16 + r;
  const bool q_valid = h < a.hq;
  if (!single && a.merge_counters && seq_len == 0) {  // nobody will arrive: stage 2's zero-split result, by split 0
    if (split == 0)
      for (int i = tid; i < 16 * kMlaDv; i += 256) {
        const int q = i / kMlaDv, d = i % kMlaDv;
        if (qb * 16 + q < a.hq)
          a.o[b * a.o_stride_t + (qb * 16 + q) * a.o_stride_h + d] =
              T::from_f32(0.f * (a.v_scale / (a.sinks ? INFINITY : 0.f)));
      }
    return;
  }
  if (split >= splits) return;
  const int32_t per = ((seq_len + splits - 1) / splits + 31) / 32 * 32;  // decode_attention.py:466-472
  const int32_t lo = per * split;
  const int32_t hi = min(lo + per, seq_len);
  if (hi <= lo) {
    if (single && seq_len == 0)
      for (int i = tid; i < 16 * kMlaDv; i += 256) {
        const int q = i / kMlaDv, d = i % kMlaDv;
        if (qb * 16 + q < a.hq) a.o[b * a.o_stride_t + (qb * 16 + q) * a.o_stride_h + d] = 0;
      }
    return;
  }
  const int ntiles = (hi - lo + kMlaTile - 1) / kMlaTile;

  // ---- Q^T fragments ------------------------------------------------------------------------------
  // split-S form (default): the score block S^T[32 tokens][16 heads] is cut four ways -- wave w computes
  // token block (w & 1) over k-steps [9 (w >> 1), +9) -- and the partial sums are exchanged through LDS, so
  // a wave issues 9 MFMAs (and keeps 9 Q fragments) instead of the 36 / 18 of the redundant form.  The sum
  // is taken in one fixed order, so every wave still holds bitwise-identical scores.
  constexpr int KSW = KS / 2;  // k-steps whose Q fragments this wave keeps
  const int ks0 = KSW * (w >> 1);
  vec8 qf[KSW];
  {
    const uint16_t* qp = a.q + b * a.q_stride_t + (q_valid ? h : 0) * a.q_stride_h + 8 * g + 32 * ks0;
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
      u32x4 raw = q_valid ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[s] = __builtin_bit_cast(vec8, raw);
    }
  }
  // ---- fused RoPE (rocm_mla_decode_rope.py:119-178): q_pe = q_pe cos + rot(q_pe) sin at positions[b] --------------
  // The rope columns 512..575 are k-steps 16 / 17: fragments 7 and 8 of the waves that keep k-steps 9..17.  Lane
  // (head r, group g) holds elements c = 8 g + j of the lower half in fragment 7 and c + 32 in fragment 8: the neox
  // partner (c <-> c + 32) is the same element of the other fragment, the interleaved (GPT-J) partner the
  // neighbouring element of the same fragment -- no cross-lane traffic either way.  fp32 arithmetic, one rounding to
  // the 16-bit operand (as rx_rope_store_kv / apply_rotary_emb write the rotated q back).
  float rc[8], rs[8];   // cos / sin for this lane's eight rope elements (this lane's k rotation uses others: see below)
  const bool rope = a.rope_cache != nullptr;
  auto rope_row = [&](int idx, float& c, float& s_) {  // table entries idx (cos) and 32 + idx (sin) of request b's position
    const int64_t pos = load_idx(a.rope_pos, b, a.rope_pos64);
    if (a.rope_f32) {
      const float* t = reinterpret_cast<const float*>(a.rope_cache) + pos * a.rope_stride;
      c = t[idx];
      s_ = t[32 + idx];
    } else {
      const uint16_t* t = reinterpret_cast<const uint16_t*>(a.rope_cache) + pos * a.rope_stride;
      c = T::to_f32(t[idx]);
      s_ = T::to_f32(t[32 + idx]);
    }
  };
  if (rope && ks0 + KSW == KS) {  // waves 2, 3
    u32x4 lo4 = __builtin_bit_cast(u32x4, qf[KSW - 2]), hi4 = __builtin_bit_cast(u32x4, qf[KSW - 1]);
    float xl[8], xh[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xl[2 * j] = T::to_f32(static_cast<uint16_t>(lo4[j] & 0xffffu));
      xl[2 * j + 1] = T::to_f32(static_cast<uint16_t>(lo4[j] >> 16));
      xh[2 * j] = T::to_f32(static_cast<uint16_t>(hi4[j] & 0xffffu));
      xh[2 * j + 1] = T::to_f32(static_cast<uint16_t>(hi4[j] >> 16));
    }
    float yl[8], yh[8];
    if (a.rope_neox) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        rope_row(8 * g + j, rc[j], rs[j]);
        yl[j] = xl[j] * rc[j] - xh[j] * rs[j];
        yh[j] = xh[j] * rc[j] + xl[j] * rs[j];
      }
    } else {  // pairs (2 i, 2 i + 1), table index i = c / 2: elements c = 8 g + j (low fragment) and 32 + 8 g + j (high)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float c0, s0, c1, s1;
        rope_row(4 * g + jj, c0, s0);
        rope_row(16 + 4 * g + jj, c1, s1);
        yl[2 * jj] = xl[2 * jj] * c0 - xl[2 * jj + 1] * s0;
        yl[2 * jj + 1] = xl[2 * jj + 1] * c0 + xl[2 * jj] * s0;
        yh[2 * jj] = xh[2 * jj] * c1 - xh[2 * jj + 1] * s1;
        yh[2 * jj + 1] = xh[2 * jj + 1] * c1 + xh[2 * jj] * s1;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo4[j] = pack2<T>(yl[2 * j], yl[2 * j + 1]);
      hi4[j] = pack2<T>(yh[2 * j], yh[2 * j + 1]);
    }
    qf[KSW - 2] = __builtin_bit_cast(vec8, lo4);
    qf[KSW - 1] = __builtin_bit_cast(vec8, hi4);
  }

  // The Q fragments must have LANDED before the tile loop: hipcc's waitcnt pass merges the loop-entry
  // state (Q loads possibly pending) into the loop header and would otherwise emit vmcnt(0) in front
  // of the first MFMA of EVERY iteration, i.e. wait for the next tile's prefetch before computing.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt/expcnt untouched
  // ---- cooperative staging: thread handles 16-byte global chunks c = tid + 256 i of the tile --------
  // 16-bit rows: 72 chunks per row, 9 per thread.  fp8 rows: 36 chunks per row = 1152 per tile = 4.5 per
  // thread (the fifth pass is taken by the first 128 threads); one chunk becomes two 16-byte LDS chunks.
  constexpr int CPRG = KV8 ? kMlaChunks / 2 : kMlaChunks;                      // global chunks per row
  constexpr int NST = (kMlaTile * CPRG + 255) / 256;                           // passes: 9 or 5
  int st_row[NST], st_col[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int c = min(tid + 256 * i, kMlaTile * CPRG - 1);  // the partial last pass re-reads the last chunk
    st_row[i] = c / CPRG;
    st_col[i] = c % CPRG;
  }
  const bool last_pass_on = tid + 256 * (NST - 1) < kMlaTile * CPRG;
  // Tiles in flight per workgroup (register sets).  2 was tried for fp8 rows (half the staging
  // registers): 114.8 -> 120.9 us at the config-5 shape, i.e. the fp8 kernel is bound by the per-tile
  // compute path (S^T, softmax, staging, barrier: ~5 k cycles), not by bytes in flight.
#ifndef RX_MLA_DEPTH
#define RX_MLA_DEPTH 1
#endif
  constexpr int DEPTH = RX_MLA_DEPTH;
  KvV stg[DEPTH][NST];
  int32_t slot_n[NST];  // slots of the tile whose loads are issued next (fetched a tile early)
  auto load_slots = [&](int t) {
#pragma unroll
    for (int i = 0; i < NST; ++i)
      slot_n[i] = static_cast<int32_t>(idx[min(lo + t * kMlaTile + st_row[i], hi - 1)]);
  };
  auto issue_loads = [&](KvV (&stg)[NST]) {
#pragma unroll
    for (int i = 0; i < NST; ++i)
      stg[i] = *reinterpret_cast<const KvV*>(
          reinterpret_cast<const KvE*>(a.kv_buf) +
          mla_slot_off<LINEAR>(slot_n[i], a.page_size, a.page_stride, a.tok_stride) + (KV8 ? 16 : 8) * st_col[i]);
  };
  auto write_lds = [&](int buf, const KvV (&stg)[NST]) {
    char* kt = smem + buf * kMlaTile * kMlaLdsRow;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      if (i == NST - 1 && !last_pass_on) break;
      char* dst = kt + st_row[i] * kMlaLdsRow + st_col[i] * (KV8 ? 32 : 16);
      if constexpr (KV8) {
        *reinterpret_cast<u32x4*>(dst) = fp8x8_to_16<T>(u32x2{stg[i][0], stg[i][1]});
        *reinterpret_cast<u32x4*>(dst + 16) = fp8x8_to_16<T>(u32x2{stg[i][2], stg[i][3]});
      } else {
        *reinterpret_cast<u32x4*>(dst) = stg[i];
      }
    }
  };

  // ---- the newest token's row under fused RoPE (rocm_mla_decode_rope.py:180-199, 228-236) ------------------------
  // Its k_pe is NOT rotated yet -- in the pool (the reference's contract: the caller stored the row, gets the rotated
  // k_pe back in k_pe_out and stores again) or in k_new (this library's form: the row has not been stored at all).  The
  // workgroups of the LAST kv split hold it in their last tile: after that tile's LDS image is complete, threads
  // 0..71 rewrite the row's chunks from the source -- rope chunks rotated -- and, with k_new, q block 0 writes the
  // finished row to its pool slot (nobody reads the slot's old bytes: every workgroup that stages the row rewrites it).
  const bool patch_wg = rope && hi == seq_len && !KV8;
  const int patch_tile = patch_wg ? ntiles - 1 : -1;
  auto patch_newest_row = [&](int buf) {
    const int first = a.k_new ? 0 : 64;  // chunks rewritten: the whole row, or its 8 rope chunks
    const int c = first + tid;
    if (c < kMlaChunks) {
      const int row = (hi - 1 - lo) % kMlaTile;
      const int64_t slot = static_cast<int64_t>(idx[hi - 1]);
      uint16_t* prow = const_cast<uint16_t*>(a.kv_buf) + mla_slot_off<LINEAR>(slot, a.page_size, a.page_stride, a.tok_stride);
      const uint16_t* src = a.k_new ? a.k_new + b * a.k_new_stride : prow;
      u32x4 v = *reinterpret_cast<const u32x4*>(src + 8 * c);
      if (c >= 64) {
        const int e0 = 8 * (c - 64);  // first rope element of this chunk
        float x[8], y[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[2 * j] = T::to_f32(static_cast<uint16_t>(v[j] & 0xffffu));
          x[2 * j + 1] = T::to_f32(static_cast<uint16_t>(v[j] >> 16));
        }
        if (a.rope_neox) {
          const bool low = e0 < 32;
          const u32x4 pv = *reinterpret_cast<const u32x4*>(src + 512 + (low ? e0 + 32 : e0 - 32));
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float part = T::to_f32(static_cast<uint16_t>(j & 1 ? pv[j >> 1] >> 16 : pv[j >> 1] & 0xffffu));
            float cc, ss;
            rope_row((e0 & 31) + j, cc, ss);
            y[j] = low ? x[j] * cc - part * ss : x[j] * cc + part * ss;
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float cc, ss;
            rope_row(e0 / 2 + jj, cc, ss);
            y[2 * jj] = x[2 * jj] * cc - x[2 * jj + 1] * ss;
            y[2 * jj + 1] = x[2 * jj + 1] * cc + x[2 * jj] * ss;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = pack2<T>(y[2 * j], y[2 * j + 1]);
        if (a.kpe_out && qb == 0) *reinterpret_cast<u32x4*>(a.kpe_out + b * a.kpe_out_stride + e0) = v;
      }
      *reinterpret_cast<u32x4*>(smem + buf * kMlaTile * kMlaLdsRow + row * kMlaLdsRow + c * 16) = v;
      if (a.k_new && qb == 0) *reinterpret_cast<u32x4*>(prow + 8 * c) = v;
    }
  };

  f32x4 oacc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  float xai = 1.0f;  // Grok temperature (decode_attention.py:156-160): the query sits at seq_len - 1
  if (a.xai_len > 0 && full_len - 1 > a.xai_len)
    xai = __log2f(static_cast<float>(full_len - 1)) / __log2f(static_cast<float>(a.xai_len));
  const bool capped = a.logit_cap > 0.f;
  const float c2 = (capped ? kLog2e : a.sm_scale * kLog2e) * xai;

  // prologue: tiles 0 .. DEPTH-1 go to sets 0 .. DEPTH-1, tile 0 is written to LDS and its set refilled
  // with tile DEPTH; invariant at iteration t: set (t+1) % DEPTH holds tile t+1, slot_n the slots of
  // tile t+1+DEPTH
  load_slots(0);
#pragma unroll
  for (int dd = 0; dd < DEPTH; ++dd) {
    if (dd < ntiles) {
      issue_loads(stg[dd]);
      if (dd + 1 < ntiles) load_slots(dd + 1);
    }
  }
  write_lds(0, stg[0]);
  if (DEPTH < ntiles) {
    issue_loads(stg[0]);
    if (DEPTH + 1 < ntiles) load_slots(DEPTH + 1);
  }
  __syncthreads();
  if (patch_tile == 0) {  // (workgroup-uniform)
    patch_newest_row(0);
    __syncthreads();
  }

  const int qd = r >> 2, pp = r & 3;
  for (int t0 = 0; t0 < ntiles; t0 += DEPTH) {
#pragma unroll
   for (int u = 0; u < DEPTH; ++u) {  // unrolled so that the register set is a compile-time index
    const int t = t0 + u;
    if (t >= ntiles) break;
    const char* kt = smem + (t % NBUF) * kMlaTile * kMlaLdsRow;
    // ---- S^T = K Q^T over all 576 columns --------------------------------------------------------
    // K fragments run RX_MLA_PD reads ahead of their MFMA: a ds_read_b128 round trip is ~100+ cycles,
    // a 16x16x32 MFMA 16; hipcc left alone reads two ahead and the in-order wave then waits on LDS
    // before nearly every one of the 36 MFMAs (measured: ~10k cycles per tile, every pipe < 25 % busy).
    f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      constexpr int PD = KSW < RX_MLA_PD ? KSW : RX_MLA_PD;
      u32x4 kf[KSW];
      const char* kb0 = kt + ((w & 1) * 16 + r) * kMlaLdsRow + g * 16 + ks0 * 64;
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = *reinterpret_cast<const u32x4*>(kb0 + i * 64);
      f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KSW; ++i) {
        if (i + PD < KSW) kf[i + PD] = *reinterpret_cast<const u32x4*>(kb0 + (i + PD) * 64);
        part = T::mfma(__builtin_bit_cast(vec8, kf[i]), qf[i], part);
      }
      f32x4* xch = reinterpret_cast<f32x4*>(smem + NBUF * kMlaTile * kMlaLdsRow);
      xch[w * 64 + lane] = part;
      __syncthreads();
      sacc[0] = xch[0 * 64 + lane] + xch[2 * 64 + lane];
      sacc[1] = xch[1 * 64 + lane] + xch[3 * 64 + lane];
    }
    // ---- online softmax (identical in all four waves) -----------------------------------------------
    float sv[8];
    const int tok_base = lo + t * kMlaTile + 4 * g;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[bb][i];
    // ONE wave-uniform branch around all eight scores: written per element, hipcc emits eight branches (and the
    // tanh expansion eight times) inside the tile loop and serialises the score exchange around them
    if (capped) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sv[j] = a.logit_cap * tanhf(sv[j] * a.sm_scale / a.logit_cap);
    }
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = (tok_base + 16 * bb + i < hi) ? sv[bb * 4 + i] : -INFINITY;
    // No row maximum on the common path (round 4; rx_extend32_kernel.inc): the scores are exponentiated against the
    // STANDING running max and the lane's partial row sum is the check (every p <= it); the max tree, the cross-lane
    // quad max and the alpha exponential -- all on the tile's dependency chain between the score exchange and the PV
    // MFMAs -- run only when a lane's sum exceeds kMlaSumLimit (or is inf / NaN: the split's first tile, m = -inf).
    // Compared as bits: sums are never negative, so the unsigned order is the float order with inf and NaN on top.
    float alpha = 1.0f, psum = 0.f;
    {
      const float m_old = m_run;
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_old));
        psum += e[j];
      }
      if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, kMlaSumLimit)) != 0) {
        float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        mt = quad_row_max(mt) * c2;
        const float m_new = fmaxf(m_old, mt);
        alpha = fast_exp2(m_old - m_new);
        m_run = m_new;
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
    l_run = l_run * alpha + psum;
    u32x4 praw;
    praw[0] = pack2<T>(sv[0], sv[1]);
    praw[1] = pack2<T>(sv[2], sv[3]);
    praw[2] = pack2<T>(sv[4], sv[5]);
    praw[3] = pack2<T>(sv[6], sv[7]);
    const vec8 pf = __builtin_bit_cast(vec8, praw);
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) oacc[nb] *= alpha;
    }
    // ---- O^T[128w + ...] += V^T P^T, V = columns [0,512) of the staged rows -------------------------
    {
      const char* rp0 = kt + (4 * g + qd) * kMlaLdsRow + (128 * w) * 2 + 8 * pp;
      const char* rp1 = rp0 + 16 * kMlaLdsRow;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const u32x2 lo2 = T::ds_read_tr(rp0 + nb * 32);
        const u32x2 hi2 = T::ds_read_tr(rp1 + nb * 32);
        const vec8 av = __builtin_bit_cast(vec8, u32x4{lo2[0], lo2[1], hi2[0], hi2[1]});
        oacc[nb] = T::mfma(av, pf, oacc[nb]);
      }
    }
    // ---- stage the next tile into the other buffer -------------------------------------------------
    if constexpr (NBUF == 1) __syncthreads();  // every wave has read tile t: its buffer may be overwritten
    if (t + 1 < ntiles) {
      write_lds((t + 1) % NBUF, stg[(u + 1) % DEPTH]);
      if (t + 1 + DEPTH < ntiles) {
        issue_loads(stg[(u + 1) % DEPTH]);
        if (t + 2 + DEPTH < ntiles) load_slots(t + 2 + DEPTH);
      }
    }
    __syncthreads();
    if (t + 1 == patch_tile) {  // (workgroup-uniform, once per kernel: the image of the last tile is complete)
      patch_newest_row((t + 1) % NBUF);
      __syncthreads();
    }
   }
  }

  // ---- epilogue: every wave holds the full statistics and its own 128 output columns -------------
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (single) {
    if (!q_valid) return;
    float den = l_run;
    if (a.sinks) den += fast_exp2(a.sinks[h] * kLog2e - m_run);
    const float inv = a.v_scale / den;
    uint16_t* op = a.o + b * a.o_stride_t + h * a.o_stride_h + 128 * w + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      u32x2 pk;
      pk[0] = pack2<T>(oacc[nb][0] * inv, oacc[nb][1] * inv);
      pk[1] = pack2<T>(oacc[nb][2] * inv, oacc[nb][3] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
    }
  } else {
    if (q_valid) {
      const int64_t row = (static_cast<int64_t>(b) * a.hq + h) * a.max_kv_splits + split;
      const float inv = 1.0f / l_run;
      float* lp = a.attn_logits + row * kMlaDv + 128 * w + 4 * g;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        if (a.merge_counters) store_dev(lp + 16 * nb, oacc[nb] * inv);  // may be merged from another XCD
        else *reinterpret_cast<f32x4*>(lp + 16 * nb) = oacc[nb] * inv;
      }
      if (w == 0 && g == 0) {
        if (a.merge_counters) store_dev(a.attn_lse + row, m_run * kLn2 + __logf(l_run));
        else a.attn_lse[row] = m_run * kLn2 + __logf(l_run);
      }
    }
    if (a.merge_counters) mla_merge_if_last<T>(a, b, qb, seq_len, splits);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// fp8 latent rows, second form (round 2): the LDS image STAYS fp8 and the rows arrive by LDS-DMA.
//
// The first form above upcasts while staging (global -> registers -> 16-bit LDS image); with fp8 rows it moves half
// the bytes through the same per-tile path and spends its largest phase ("staging": wait for the rows, 36 converts and
// 9 ds_write_b128 per thread, re-issue) off the memory system: 4.0-4.4 TB/s.  Here
//   * a tile of 32 rows is 32 x 592 B in LDS (576 data + 16 pad) -- three tiles in a ring are 60 KB, two workgroups
//     per CU -- and is filled by `global_load_lds_dwordx4` (1 KiB per wave instruction, per-lane source address =
//     row gather): no staging registers, no ds_write, no convert on the way in, and TWO tiles in flight per workgroup;
//   * the upcast moves to the fragment reads and costs the same number of converts as before because the four-way
//     score split reads every K element exactly once: K fragments by ds_read_b64 (8 fp8 -> one 16-bit MFMA operand),
//     V^T fragments by ONE ds_read_b64_tr_b8 each (per 16 lanes: lane 2q + p supplies the address of row q, bytes
//     8p .. 8p+7; lane i receives byte column i of the 8 rows -- measured, tools/probe/tr8_probe.hip).  S^T row
//     rho = 4 g + i of token block bb is mapped to tile row 8 g + 4 bb + i, so that the eight k-slots of a lane
//     group's P fragment are the CONTIGUOUS rows 8 g .. 8 g + 7 the transposed read delivers;
//   * the rows' slot ids come from LDS (1024 tokens staged at a time), so the steady-state loop issues no VMEM
//     instruction but the DMA pieces and `s_waitcnt vmcnt(5)` means exactly "tile t has landed, tile t+1 may fly".
// hipcc neither counts nor orders the asm DMA: every wait on it is written out below.
constexpr int kM8Row = 592;                    // LDS row stride of the fp8 image
constexpr int kM8Cpr = kM8Row / 16;            // 37 chunks per row (36 data + 1 pad)
// Rows 16-31 of a tile image start kM8HalfShift bytes late.  A score fragment read (ds_read_b64, served per 32-lane
// half) takes rows {0-3, 8-11, 16-19, 24-27} (+4 for the odd waves) at one column: 16 rows apart is 37 bank rows apart,
// the same bank, and no row stride separates them (8 rows x an odd chunk count is 8 slots, x an even one 0) -- every
// such read took twice its cycles (SQ_LDS_BANK_CONFLICT = 0.53 of SQ_LDS_IDX_ACTIVE).  Four chunks of offset move the
// upper rows onto the eight 16-byte slots the lower ones leave free; the transposed V reads take 16 consecutive rows
// of ONE half and do not notice.
constexpr int kM8HalfShift = 64;
constexpr int kM8Pieces = 20;                  // 1-KiB DMA pieces per tile buffer: 5 per wave (18.5 carry data)
constexpr int kM8Buf = kM8Pieces * 1024;       // 20 KB
#ifndef RX_M8_RING
#define RX_M8_RING 3  // tile buffers; RING - 1 tiles are in flight or resident ahead of the one being read
#endif
constexpr int kM8Ring = RX_M8_RING;
constexpr int kM8Ahead = kM8Ring - 1;
constexpr int kM8SlotBlock = 1024;             // tokens whose slot ids are staged in LDS at a time
constexpr int kM8Lds = kM8Ring * kM8Buf + 4 * 64 * 16 + 2 * kM8SlotBlock * 4;

// one 1-KiB LDS-DMA piece: lane l's 16 bytes land at lds_dst + 16 l (recipe: cdna_hip_programming.md 5.7)
#ifndef RX_M8_NT
#define RX_M8_NT 0  // 1: non-temporal DMA loads (the rows are read once per step)
#endif
__device__ __forceinline__ void m8_dma16(const void* gsrc, uint32_t lds_dst) {
#if RX_M8_NT
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(gsrc), "s"(lds_dst) : "memory");
#else
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}

template <typename T, typename IdxT, bool LINEAR>
__global__ __launch_bounds__(256, (kM8Ring == 2 ? 3 : 2)) void decode_mla8_dma_kernel(const MlaArgs a) {
  using vec8 = typename T::vec8;
  constexpr int KS = kMlaDk / 32;       // 18 k-steps
  constexpr int NBW = kMlaDv / 16 / 4;  // 8 d-blocks of 16 per wave
  constexpr int KSW = KS / 2;           // k-steps of this wave's score part
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3 tiles][exchange][2 slot blocks]
  char* const xch_base = smem + kM8Ring * kM8Buf;
  int32_t* const slots_lds = reinterpret_cast<int32_t*>(xch_base + 4 * 64 * 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  int qb, b, split;
  {
    const int G = a.qblocks;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qb = j % G;
    const int pr = (j / G) * 8 + xcd;  // (request, split) pair, bound to one XCD (see the first form)
    if (pr >= a.bs * a.max_kv_splits) return;
    b = pr % a.bs;
    split = pr / a.bs;
  }
  int32_t seq_len;
  const IdxT* idx;
  if (a.kv_indices) {
    const int32_t beg = a.kv_indptr[b];
    seq_len = a.kv_indptr[b + 1] - beg;
    idx = reinterpret_cast<const IdxT*>(a.kv_indices) + beg;
  } else {
    const int64_t req = load_idx(a.req_pool_indices, b, a.rpi64);
    seq_len = static_cast<int32_t>(load_idx(a.seq_lens, b, a.sl64));
    idx = reinterpret_cast<const IdxT*>(a.req_to_token + req * a.req_row_stride);
  }
  const int32_t full_len = seq_len;  // the query's position + 1 (Grok temperature)
  if (a.kv_start && !a.kv_indices) {  // the suffix [kv_start[b], seq_len) only (shared-prefix cascade, phase 2)
    const int32_t st = min(max(a.kv_start[b], 0), seq_len);
    idx += st;
    seq_len -= st;
  }
  const int32_t splits = (a.num_kv_splits && a.max_kv_splits > 1) ? a.num_kv_splits[b] : 1;
  const bool single = (a.max_kv_splits == 1) || (a.direct_single && splits == 1);
  const int h = qb * 16 + r;
  const bool q_valid = h < a.hq;
  if (!single && a.merge_counters && seq_len == 0) {  // nobody will arrive: stage 2's zero-split result, by split 0
    if (split == 0)
      for (int i = tid; i < 16 * kMlaDv; i += 256) {
        const int q = i / kMlaDv, d = i % kMlaDv;
        if (qb * 16 + q < a.hq)
          a.o[b * a.o_stride_t + (qb * 16 + q) * a.o_stride_h + d] =
              T::from_f32(0.f * (a.v_scale / (a.sinks ? INFINITY : 0.f)));
      }
    return;
  }
  if (split >= splits) return;
  const int32_t per = ((seq_len + splits - 1) / splits + 31) / 32 * 32;
  const int32_t lo = per * split;
  const int32_t hi = min(lo + per, seq_len);
  if (hi <= lo) {
    if (single && seq_len == 0)
      for (int i = tid; i < 16 * kMlaDv; i += 256) {
        const int q = i / kMlaDv, d = i % kMlaDv;
        if (qb * 16 + q < a.hq) a.o[b * a.o_stride_t + (qb * 16 + q) * a.o_stride_h + d] = 0;
      }
    return;
  }
  const int ntiles = (hi - lo + kMlaTile - 1) / kMlaTile;

  // ---- Q^T fragments of this wave's k-steps
  const int ks0 = KSW * (w >> 1);
  vec8 qf[KSW];
  {
    const uint16_t* qp = a.q + b * a.q_stride_t + (q_valid ? h : 0) * a.q_stride_h + 8 * g + 32 * ks0;
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
      u32x4 raw = q_valid ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[s] = __builtin_bit_cast(vec8, raw);
    }
  }
  // Q8 (bf16 queries, round 5): QK^T on the fp8 MFMA with the latent rows AS STORED -- no upcast of the K fragments (36 of
  // the wave's 68 converts per tile).  q is the sum of TWO e4m3 terms under one power-of-two scale per (head, this wave's
  // half of the 576 columns): x = q s, hi = e4m3(x), lo = e4m3((x - hi) 16); S = (K hi + K lo / 16) / s.  A bf16 value
  // has 8 significant bits, hi keeps 4 and the remainder fits lo's 4 (tools/mla_fp8_qk_study.py: logit error 1e-7 rms,
  // worst err / bound 0.52 = the 16-bit path's; ONE term: 63).  fp16 queries would need three terms: they keep the upcast.
  // MEASURED AND NOT ENABLED (RX_MLA_Q8 = 0): parity green (60 GPU cases incl. the adversarial patterns), VALU 151 -> 119 per
  // wave and tile, MFMAs 17 -> 26 -- and the config-5 shard ran 62.4 -> 66 us.  The tile's critical path is its latency chain
  // (DMA landing -> QK^T chain -> exchange barrier -> softmax -> PV), not the wave's VALU issue: nine more dependent
  // 16x16x32 links in the first phase lengthen exactly that.  What would pay is the block-scaled 16x16x128 form (K = 128 per
  // MFMA: FEWER links, both contractions on fp8 operands) on 128-token tiles -- a different kernel, DESIGN 4.1b.
#ifndef RX_MLA_Q8
#define RX_MLA_Q8 0
#endif
  constexpr bool Q8 = RX_MLA_Q8 && std::is_same_v<T, BF16>;
  long qh[Q8 ? KSW : 1], ql[Q8 ? KSW : 1];
  float q_inv_scale = 1.0f;
  if constexpr (Q8) {
    float amax = 0.f;
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
      const u32x4 raw = __builtin_bit_cast(u32x4, qf[s]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        amax = fmaxf(amax, fabsf(__builtin_bit_cast(float, raw[j] << 16)));
        amax = fmaxf(amax, fabsf(__builtin_bit_cast(float, raw[j] & 0xffff0000u)));
      }
    }
    amax = fmaxf(amax, __shfl_xor(amax, 16));  // the head's four lane groups hold its other columns
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    // s = 2^(7 - e) for amax = m 2^e, m in [1, 2): the largest |x| lands in [128, 256) (e4m3 tops out at 448)
    const int e = static_cast<int>((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu) - 127;
    const int es = amax > 0.f ? min(max(7 - e, -100), 100) : 0;
    const float sc = __builtin_bit_cast(float, static_cast<uint32_t>(127 + es) << 23);
    q_inv_scale = __builtin_bit_cast(float, static_cast<uint32_t>(127 - es) << 23);
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
      const u32x4 raw = __builtin_bit_cast(u32x4, qf[s]);
      float x[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[2 * j] = __builtin_bit_cast(float, raw[j] << 16) * sc;
        x[2 * j + 1] = __builtin_bit_cast(float, raw[j] & 0xffff0000u) * sc;
      }
      int hw[2] = {0, 0}, lw[2] = {0, 0};
#pragma unroll
      for (int j = 0; j < 2; ++j) {  // (the word selector of the converts is an immediate: low half, then high half)
        hw[j] = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * j], x[4 * j + 1], hw[j], false);
        hw[j] = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * j + 2], x[4 * j + 3], hw[j], true);
        const f32x2 b0 = __builtin_amdgcn_cvt_pk_f32_fp8(hw[j], false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8(hw[j], true);
        lw[j] = __builtin_amdgcn_cvt_pk_fp8_f32((x[4 * j] - b0[0]) * 16.0f, (x[4 * j + 1] - b0[1]) * 16.0f, lw[j], false);
        lw[j] = __builtin_amdgcn_cvt_pk_fp8_f32((x[4 * j + 2] - b1[0]) * 16.0f, (x[4 * j + 3] - b1[1]) * 16.0f, lw[j], true);
      }
      qh[s] = static_cast<long>((static_cast<uint64_t>(static_cast<uint32_t>(hw[1])) << 32) | static_cast<uint32_t>(hw[0]));
      ql[s] = static_cast<long>((static_cast<uint64_t>(static_cast<uint32_t>(lw[1])) << 32) | static_cast<uint32_t>(lw[0]));
    }
  }
  // ---- slot ids of the first block of tokens -> LDS
  auto stage_slots = [&](int blk) {  // tokens [blk * 1024, +1024) of this split -> slot block blk & 1
    int32_t* dst = slots_lds + (blk & 1) * kM8SlotBlock;
#pragma unroll
    for (int i = 0; i < kM8SlotBlock / 256; ++i) {
      const int tok = blk * kM8SlotBlock + tid + 256 * i;
      dst[tid + 256 * i] = static_cast<int32_t>(idx[min(lo + tok, hi - 1)]);
    }
  };
  stage_slots(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and the slot ids have landed; nothing of ours is in flight yet
  __syncthreads();

  // ---- DMA of one tile: wave w issues pieces w, w + 4, ..., 5 per wave (the last ones carry padding only)
  // The five slot ids a lane needs for a tile are read from LDS ONE iteration before the tile is issued, so the issue
  // is five address computations and five DMA instructions with no LDS round trip between them.
  const uint8_t* kvb = reinterpret_cast<const uint8_t*>(a.kv_buf);
  constexpr int NP = kM8Pieces / 4;
  int prow[NP], pcol[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    // chunk position of the padded image; rows 16-31 sit kM8HalfShift bytes further (see kM8HalfShift), the four
    // positions in between re-read row 16's first chunk, the tail repeats the last chunk
    const int pos = min((w + 4 * i) * 64 + lane, kMlaTile * kM8Cpr + kM8HalfShift / 16 - 1);
    const int c = pos < 16 * kM8Cpr ? pos : max(pos - kM8HalfShift / 16, 16 * kM8Cpr);
    prow[i] = c / kM8Cpr;
    pcol[i] = 16 * min(c % kM8Cpr, kM8Cpr - 2);  // pad chunk: re-reads the row's last data chunk
  }
  int32_t pslot[NP];
  auto read_slots = [&](int t) {
    const int32_t* sl = slots_lds + ((t * kMlaTile / kM8SlotBlock) & 1) * kM8SlotBlock + (t * kMlaTile) % kM8SlotBlock;
#pragma unroll
    for (int i = 0; i < NP; ++i) pslot[i] = sl[prow[i]];
  };
  const uint8_t* psrc[NP];
  auto dma_addr = [&]() {  // pslot -> source addresses
#pragma unroll
    for (int i = 0; i < NP; ++i)
      psrc[i] = kvb + mla_slot_off<LINEAR>(static_cast<int64_t>(pslot[i]), a.page_size, a.page_stride, a.tok_stride) + pcol[i];
  };
  auto dma_pieces = [&](int t, int i0, int i1) {  // pieces [i0, i1) of tile t from psrc
    const uint32_t buf = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)) + (t % kM8Ring) * kM8Buf;
#pragma unroll
    for (int i = i0; i < i1; ++i) m8_dma16(psrc[i], __builtin_amdgcn_readfirstlane(buf + (w + 4 * i) * 1024));
  };
  auto dma_tile = [&](int t) {  // uses pslot (tile t's)
    dma_addr();
    dma_pieces(t, 0, NP);
  };
  // tiles past the end are "loaded" as well (their rows clamp to the last token): the counted waits stay uniform
#pragma unroll
  for (int tt = 0; tt < kM8Ahead; ++tt) {
    read_slots(tt);
    dma_tile(tt);
  }
  read_slots(kM8Ahead);

  f32x4 oacc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  float xai = 1.0f;
  if (a.xai_len > 0 && full_len - 1 > a.xai_len)
    xai = __log2f(static_cast<float>(full_len - 1)) / __log2f(static_cast<float>(a.xai_len));
  const bool capped = a.logit_cap > 0.f;
  const float c2 = (capped ? kLog2e : a.sm_scale * kLog2e) * xai;
  const int bb_w = w & 1;
  const int i16 = lane & 15;

  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed (ours: all but the 5 youngest pieces = tile t+1's; everybody's: the barrier)
    if constexpr (kM8Ahead == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (kM8Ahead == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (kM8Ahead == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (kM8Ahead == 4) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (kM8Ahead == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    __syncthreads();
    const char* kt = smem + (t % kM8Ring) * kM8Buf;
    // next slot block, one block ahead of the DMA that will read it (rare: every 32 tiles)
    if ((t & 31) == 0 && (t / 32 + 1) * kM8SlotBlock < hi - lo + (kM8Ahead + 2) * kMlaTile) {
      stage_slots(t / 32 + 1);  // published by the exchange barrier below; first read at iteration 32 k + 31 - kM8Ahead - 1
    }
    // tile t+2 -> the buffer tile t-1 was read from (every wave is past this iteration's barrier, i.e. done with t-1)
    dma_addr();
    dma_pieces(t + kM8Ahead, 0, 2);
    read_slots(t + kM8Ahead + 1);  // for the next iteration's issue

    // ---- partial S^T of this wave: token block bb_w, k-steps [ks0, ks0 + 9)
    f32x4 sacc[2];
    {
      const char* kb0 = kt + (8 * (r >> 2) + 4 * bb_w + (r & 3)) * kM8Row + (r >= 8 ? kM8HalfShift : 0) + 8 * g + 32 * ks0;
      constexpr int PD = 4;
      u32x2 kf[KSW];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = *reinterpret_cast<const u32x2*>(kb0 + i * 32);
      f32x4 part;
      if constexpr (Q8) {  // four independent chains (hi / lo x even / odd k-step): a dependent 16x16x32 waits out its predecessor
        f32x4 ph[2], pl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) ph[c] = pl[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
          if (i + PD < KSW) kf[i + PD] = *reinterpret_cast<const u32x2*>(kb0 + (i + PD) * 32);
          const long kraw = static_cast<long>((static_cast<uint64_t>(kf[i][1]) << 32) | kf[i][0]);
          ph[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kraw, qh[i], ph[i & 1], 0, 0, 0);
          pl[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kraw, ql[i], pl[i & 1], 0, 0, 0);
        }
        part = ((ph[0] + ph[1]) + (pl[0] + pl[1]) * 0.0625f) * q_inv_scale;
      } else {
        f32x4 part3[3];  // three independent chains: a dependent 16x16x32 waits out its predecessor
#pragma unroll
        for (int c = 0; c < 3; ++c) part3[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
          if (i + PD < KSW) kf[i + PD] = *reinterpret_cast<const u32x2*>(kb0 + (i + PD) * 32);
          part3[i % 3] = T::mfma(__builtin_bit_cast(vec8, fp8x8_to_16<T>(kf[i])), qf[i], part3[i % 3]);
        }
        part = part3[0] + part3[1] + part3[2];
      }
      f32x4* xch = reinterpret_cast<f32x4*>(xch_base);
      xch[w * 64 + lane] = part;
      __syncthreads();
      sacc[0] = xch[0 * 64 + lane] + xch[2 * 64 + lane];
      sacc[1] = xch[1 * 64 + lane] + xch[3 * 64 + lane];
    }
    dma_pieces(t + kM8Ahead, 2, 4);
    // ---- online softmax (identical in all four waves); score (bb, i) of this lane is token 8 g + 4 bb + i
    float sv[8];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[bb][i];
    if (capped) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sv[j] = a.logit_cap * tanhf(sv[j] * a.sm_scale / a.logit_cap);
    }
    if (t == ntiles - 1) {  // only the split's last tile can reach past its end
      const int tok_base = lo + t * kMlaTile + 8 * g;
#pragma unroll
      for (int j = 0; j < 8; ++j) sv[j] = (tok_base + j < hi) ? sv[j] : -INFINITY;
    }
    // (no row maximum on the common path: the sum check of the first form above)
    float alpha = 1.0f, psum = 0.f;
    {
      const float m_old = m_run;
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_old));
        psum += e[j];
      }
      if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, kMlaSumLimit)) != 0) {
        float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        mt = quad_row_max(mt) * c2;
        const float m_new = fmaxf(m_old, mt);
        alpha = fast_exp2(m_old - m_new);
        m_run = m_new;
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
    l_run = l_run * alpha + psum;
    u32x4 praw;
    praw[0] = pack2<T>(sv[0], sv[1]);
    praw[1] = pack2<T>(sv[2], sv[3]);
    praw[2] = pack2<T>(sv[4], sv[5]);
    praw[3] = pack2<T>(sv[6], sv[7]);
    const vec8 pf = __builtin_bit_cast(vec8, praw);
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) oacc[nb] *= alpha;
    }
    dma_pieces(t + kM8Ahead, 4, NP);
    // ---- O^T[128 w + ...] += V^T P^T: one transposed byte read per fragment (rows 8 g .. 8 g + 7), upcast, MFMA
    {
      const char* vp = kt + (8 * g + (i16 >> 1)) * kM8Row + (g >= 2 ? kM8HalfShift : 0) + 128 * w + 8 * (i16 & 1);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        auto lp = (__attribute__((address_space(3))) v2i_t*)(uintptr_t)(uint32_t)(uintptr_t)(vp + 16 * nb);
        const v2i_t raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32(lp);
        const vec8 av = __builtin_bit_cast(vec8, fp8x8_to_16<T>(u32x2{static_cast<uint32_t>(raw[0]), static_cast<uint32_t>(raw[1])}));
        oacc[nb] = T::mfma(av, pf, oacc[nb]);
      }
    }
  }
  // nothing of this workgroup may still be landing in LDS when the block retires
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue (as the first form)
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (single) {
    if (!q_valid) return;
    float den = l_run;
    if (a.sinks) den += fast_exp2(a.sinks[h] * kLog2e - m_run);
    const float inv = a.v_scale / den;
    uint16_t* op = a.o + b * a.o_stride_t + h * a.o_stride_h + 128 * w + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      u32x2 pk;
      pk[0] = pack2<T>(oacc[nb][0] * inv, oacc[nb][1] * inv);
      pk[1] = pack2<T>(oacc[nb][2] * inv, oacc[nb][3] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
    }
  } else {
    if (q_valid) {
      const int64_t row = (static_cast<int64_t>(b) * a.hq + h) * a.max_kv_splits + split;
      const float inv = 1.0f / l_run;
      float* lp = a.attn_logits + row * kMlaDv + 128 * w + 4 * g;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        if (a.merge_counters) store_dev(lp + 16 * nb, oacc[nb] * inv);  // may be merged from another XCD
        else *reinterpret_cast<f32x4*>(lp + 16 * nb) = oacc[nb] * inv;
      }
      if (w == 0 && g == 0) {
        if (a.merge_counters) store_dev(a.attn_lse + row, m_run * kLn2 + __logf(l_run));
        else a.attn_lse[row] = m_run * kLn2 + __logf(l_run);
      }
    }
    if (a.merge_counters) mla_merge_if_last<T>(a, b, qb, seq_len, splits);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// fp8 latent rows, third form (round 5): 64-token tiles, every wave owns 16 TOKENS of the tile.
//
// The second form above splits QK^T four ways inside a 32-token tile (2 token blocks x 2 halves of the 576 columns):
// every tile pays an exchange of partial sums through LDS behind a barrier, and all four waves then run the SAME softmax
// of the tile's 512 scores.  Measured: that kernel is bound by the tile's latency chain -- landing ->
// QK^T chain -> exchange barrier -> softmax -> PV -- not by VALU issue or HBM.  Here
//   * a tile is 64 rows x 592 B = exactly 37 one-KiB DMA pieces, two tile buffers per workgroup (two workgroups per CU);
//   * wave w computes the scores of tokens 16 w .. 16 w + 15 against ALL 576 columns (18 k-steps, no partial sums), does the
//     softmax of those 256 scores only (4 per lane instead of 8 redundant ones per 32 tokens), and publishes its P block
//     (16 tokens x 16 heads, 16-bit) in LDS; behind ONE barrier every wave reads the tile's whole P as its PV B operands
//     and accumulates its own 128 output columns.  Two barriers per 64 tokens instead of four, no score exchange;
//   * 16 CONSECUTIVE rows at one column land on 16 different 16-byte bank groups (592 B = 148 banks, 148 mod 64 = 20),
//     so neither the score fragment reads nor the transposed V reads need the second form's half shift;
//   * the standing reference max of a head must be the same in all four waves (their P blocks meet in one PV product):
//     a wave whose sum check fires raises a flag next to its P block, and behind the barrier ALL waves take the redo --
//     per-head maxima through LDS, a common new max, P recomputed and republished (three more barriers; rare).
constexpr int kT64 = 64;                                   // tokens per tile
constexpr int kT64Buf = kT64 * kM8Row;                     // 37,888 B = 37 pieces
constexpr int kT64Pieces = kT64Buf / 1024;                 // 37
constexpr int kT64PRow = 144;                              // bytes per head row of the P image (64 tokens x 2 B + 16: bank spread)
constexpr int kT64SlotBlock = 256;                         // tokens whose slot ids are staged at a time (4 tiles)
constexpr int kT64Lds = 2 * kT64Buf + 16 * kT64PRow + 64 /* flags */ + 4 * 16 * 4 /* maxima */ + 4 * 16 * 4 /* sums */ +
                        2 * kT64SlotBlock * 4;             // 80,192 B: two workgroups per CU
static_assert(kT64Buf % 1024 == 0 && 2 * kT64Lds <= 160 * 1024, "tile image / LDS budget");

template <typename T, typename IdxT, bool LINEAR>
__global__ __launch_bounds__(256, 2) void decode_mla8_t64_kernel(const MlaArgs a) {
  using vec8 = typename T::vec8;
  constexpr int KS = kMlaDk / 32;       // 18 k-steps
  constexpr int NBW = kMlaDv / 16 / 4;  // 8 d-blocks of 16 per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 tiles][P image][flags][maxima][sums][2 slot blocks]
  char* const p_img = smem + 2 * kT64Buf;
  int32_t* const flags = reinterpret_cast<int32_t*>(p_img + 16 * kT64PRow);
  float* const maxbuf = reinterpret_cast<float*>(flags + 16);
  float* const sumbuf = maxbuf + 4 * 16;
  int32_t* const slots_lds = reinterpret_cast<int32_t*>(sumbuf + 4 * 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  int qb, b, split;
  {
    const int G = a.qblocks;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qb = j % G;
    const int pr = (j / G) * 8 + xcd;  // (request, split) pair, bound to one XCD (see the first form)
    if (pr >= a.bs * a.max_kv_splits) return;
    b = pr % a.bs;
    split = pr / a.bs;
  }
  int32_t seq_len;
  const IdxT* idx;
  if (a.kv_indices) {
    const int32_t beg = a.kv_indptr[b];
    seq_len = a.kv_indptr[b + 1] - beg;
    idx = reinterpret_cast<const IdxT*>(a.kv_indices) + beg;
  } else {
    const int64_t req = load_idx(a.req_pool_indices, b, a.rpi64);
    seq_len = static_cast<int32_t>(load_idx(a.seq_lens, b, a.sl64));
    idx = reinterpret_cast<const IdxT*>(a.req_to_token + req * a.req_row_stride);
  }
  const int32_t full_len = seq_len;
  if (a.kv_start && !a.kv_indices) {
    const int32_t st = min(max(a.kv_start[b], 0), seq_len);
    idx += st;
    seq_len -= st;
  }
  const int32_t splits = (a.num_kv_splits && a.max_kv_splits > 1) ? a.num_kv_splits[b] : 1;
  const bool single = (a.max_kv_splits == 1) || (a.direct_single && splits == 1);
  const int h = qb * 16 + r;
  const bool q_valid = h < a.hq;
  if (!single && a.merge_counters && seq_len == 0) {  // nobody will arrive: stage 2's zero-split result, by split 0
    if (split == 0)
      for (int i = tid; i < 16 * kMlaDv; i += 256) {
        const int q = i / kMlaDv, d = i % kMlaDv;
        if (qb * 16 + q < a.hq)
          a.o[b * a.o_stride_t + (qb * 16 + q) * a.o_stride_h + d] = T::from_f32(0.f * (a.v_scale / (a.sinks ? INFINITY : 0.f)));
      }
    return;
  }
  if (split >= splits) return;
  const int32_t per = ((seq_len + splits - 1) / splits + 31) / 32 * 32;  // (the split boundaries of the stage-2 contract)
  const int32_t lo = per * split;
  const int32_t hi = min(lo + per, seq_len);
  if (hi <= lo) {
    if (single && seq_len == 0)
      for (int i = tid; i < 16 * kMlaDv; i += 256) {
        const int q = i / kMlaDv, d = i % kMlaDv;
        if (qb * 16 + q < a.hq) a.o[b * a.o_stride_t + (qb * 16 + q) * a.o_stride_h + d] = 0;
      }
    return;
  }
  const int ntiles = (hi - lo + kT64 - 1) / kT64;

  // ---- Q^T fragments of all 18 k-steps: lane (head r, k group g) holds q[h][32 s + 8 g .. + 8]
  vec8 qf[KS];
  {
    const uint16_t* qp = a.q + b * a.q_stride_t + (q_valid ? h : 0) * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4 raw = q_valid ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[s] = __builtin_bit_cast(vec8, raw);
    }
  }
  // slot ids of 256 tokens (4 tiles) at a time, one per thread: LOADED at the end of an iteration and written to LDS at the
  // top of the next one, behind that iteration's own vmcnt(0) -- hipcc does not count the asm DMA pieces, so a wait it places
  // for this load right after they were issued would wait for all of them
  int32_t slot_pending = 0;
  auto load_slot = [&](int blk) { slot_pending = static_cast<int32_t>(idx[min(lo + blk * kT64SlotBlock + tid, hi - 1)]); };
  auto store_slot = [&](int blk) { slots_lds[(blk & 1) * kT64SlotBlock + tid] = slot_pending; };
  load_slot(0);
  store_slot(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and the slot ids have landed
  __syncthreads();

  // ---- DMA of one tile: wave w issues pieces w, w + 4, ...: 10 / 9 / 9 / 9 of the 37
  const uint8_t* kvb = reinterpret_cast<const uint8_t*>(a.kv_buf);
  constexpr int NP = (kT64Pieces + 3) / 4;  // 10
  int prow[NP], pcol[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pos = min((w + 4 * i) * 64 + lane, kT64 * kM8Cpr - 1);  // 16-byte chunk of the image
    prow[i] = pos / kM8Cpr;
    pcol[i] = 16 * min(pos % kM8Cpr, kM8Cpr - 2);  // the pad chunk re-reads the row's last data chunk
  }
  int32_t pslot[NP];
  auto read_slots = [&](int t) {
    const int32_t* sl = slots_lds + ((t * kT64 / kT64SlotBlock) & 1) * kT64SlotBlock + (t * kT64) % kT64SlotBlock;
#pragma unroll
    for (int i = 0; i < NP; ++i) pslot[i] = sl[prow[i]];
  };
  auto dma_pieces = [&](int t, int i0, int i1) {  // pieces [i0, i1) of tile t from pslot
    const uint32_t buf = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)) + (t & 1) * kT64Buf;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      if (w + 4 * i < kT64Pieces) {  // (wave-uniform: only wave 0 has a tenth piece)
        const uint8_t* src = kvb + mla_slot_off<LINEAR>(static_cast<int64_t>(pslot[i]), a.page_size, a.page_stride, a.tok_stride) + pcol[i];
        m8_dma16(src, __builtin_amdgcn_readfirstlane(buf + (w + 4 * i) * 1024));
      }
    }
  };
  read_slots(0);
  dma_pieces(0, 0, NP);
  if (ntiles > 1) read_slots(1);

  f32x4 oacc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;   // m_run: the head's reference max (identical in all four waves); l_run: THIS wave's tokens
  float xai = 1.0f;
  if (a.xai_len > 0 && full_len - 1 > a.xai_len)
    xai = __log2f(static_cast<float>(full_len - 1)) / __log2f(static_cast<float>(a.xai_len));
  const bool capped = a.logit_cap > 0.f;
  const float c2 = (capped ? kLog2e : a.sm_scale * kLog2e) * xai;
  const int i16 = lane & 15;

  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of tile t have landed ...
    __syncthreads();                                   // ... and everybody's; everybody is done with tile t - 1 and its P image
    const char* kt = smem + (t & 1) * kT64Buf;
    // slot block t / 4 + 1 (loaded at the end of iteration t - 1): visible behind this iteration's second barrier, first
    // read at iteration t + 1 (tile t + 3)
    if ((t & 3) == 1) store_slot(t / 4 + 1);
    // tile t + 1 -> the buffer tile t - 1 was read from
    if (t + 1 < ntiles) dma_pieces(t + 1, 0, 3);

    // ---- S^T of this wave's 16 tokens: 18 k-steps, three independent chains
    f32x4 sacc;
    {
      const char* kb0 = kt + (16 * w + r) * kM8Row + 8 * g;   // A operand: row = token 16 w + r, k group g
      constexpr int PD = 4;
      u32x2 kf[KS];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = *reinterpret_cast<const u32x2*>(kb0 + i * 32);
      f32x4 part3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) part3[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        if (i + PD < KS) kf[i + PD] = *reinterpret_cast<const u32x2*>(kb0 + (i + PD) * 32);
        part3[i % 3] = T::mfma(__builtin_bit_cast(vec8, fp8x8_to_16<T>(kf[i])), qf[i], part3[i % 3]);
        if (i == 5 && t + 1 < ntiles) dma_pieces(t + 1, 3, 6);
        if (i == 11 && t + 1 < ntiles) dma_pieces(t + 1, 6, NP);
      }
      sacc = part3[0] + part3[1] + part3[2];
    }
    if (t + 2 < ntiles) read_slots(t + 2);  // for the next iteration's issue (its slot block was stored >= 1 barrier ago)
    if ((t & 3) == 0) load_slot(t / 4 + 1);
    // ---- softmax of the wave's own scores: lane (head r, group g) holds tokens 16 w + 4 g + i, i < 4
    float sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sv[i] = sacc[i];
    if (capped) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = a.logit_cap * tanhf(sv[i] * a.sm_scale / a.logit_cap);
    }
    if (t == ntiles - 1) {  // only the split's last tile can reach past its end
      const int tok_base = lo + t * kT64 + 16 * w + 4 * g;
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = (tok_base + i < hi) ? sv[i] : -INFINITY;
    }
    float e[4], psum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      e[i] = fast_exp2(__builtin_fmaf(sv[i], c2, -m_run));
      psum += e[i];
    }
    const bool mine = __builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, kMlaSumLimit)) != 0;
    char* const prow_w = p_img + r * kT64PRow + (16 * w + 4 * g) * 2;   // P image: [head][token], 16-bit
    *reinterpret_cast<u32x2*>(prow_w) = u32x2{pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3])};
    if (lane == 0) flags[w] = mine ? 1 : 0;
    __syncthreads();
    float alpha = 1.0f;
    if ((flags[0] | flags[1] | flags[2] | flags[3]) != 0) {  // (workgroup-uniform) some lane's sum ran away: the classic step, together
      float mt = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
      mt = quad_row_max(mt);                     // over the head's four lane groups: this wave's 16 tokens
      if (g == 0) maxbuf[w * 16 + r] = mt;
      __syncthreads();
      mt = fmaxf(fmaxf(maxbuf[r], maxbuf[16 + r]), fmaxf(maxbuf[32 + r], maxbuf[48 + r])) * c2;
      const float m_new = fmaxf(m_run, mt);
      alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      psum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = fast_exp2(__builtin_fmaf(sv[i], c2, -m_new));
        psum += e[i];
      }
      *reinterpret_cast<u32x2*>(prow_w) = u32x2{pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3])};
      __syncthreads();  // (the flags and maxima are rewritten only behind the next iteration's first barrier)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) oacc[nb] *= alpha;
    }
    l_run = l_run * alpha + psum;
    // ---- O^T[128 w + ...] += V^T P^T over the tile's 64 tokens: two k-steps of 32
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const vec8 pf = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(p_img + r * kT64PRow + (32 * ks + 8 * g) * 2));
      const char* vp = kt + (32 * ks + 8 * g + (i16 >> 1)) * kM8Row + 128 * w + 8 * (i16 & 1);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        auto lp = (__attribute__((address_space(3))) v2i_t*)(uintptr_t)(uint32_t)(uintptr_t)(vp + 16 * nb);
        const v2i_t raw = __builtin_amdgcn_ds_read_tr8_b64_v2i32(lp);
        const vec8 av = __builtin_bit_cast(vec8, fp8x8_to_16<T>(u32x2{static_cast<uint32_t>(raw[0]), static_cast<uint32_t>(raw[1])}));
        oacc[nb] = T::mfma(av, pf, oacc[nb]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this workgroup may still be landing in LDS when it retires

  // ---- the head's row sum: this wave's four lane groups, then the four waves
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  __syncthreads();
  if (g == 0) sumbuf[w * 16 + r] = l_run;
  __syncthreads();
  l_run = (sumbuf[r] + sumbuf[16 + r]) + (sumbuf[32 + r] + sumbuf[48 + r]);
  if (single) {
    if (!q_valid) return;
    float den = l_run;
    if (a.sinks) den += fast_exp2(a.sinks[h] * kLog2e - m_run);
    const float inv = a.v_scale / den;
    uint16_t* op = a.o + b * a.o_stride_t + h * a.o_stride_h + 128 * w + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      u32x2 pk;
      pk[0] = pack2<T>(oacc[nb][0] * inv, oacc[nb][1] * inv);
      pk[1] = pack2<T>(oacc[nb][2] * inv, oacc[nb][3] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
    }
  } else {
    if (q_valid) {
      const int64_t row = (static_cast<int64_t>(b) * a.hq + h) * a.max_kv_splits + split;
      const float inv = 1.0f / l_run;
      float* lp = a.attn_logits + row * kMlaDv + 128 * w + 4 * g;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        if (a.merge_counters) store_dev(lp + 16 * nb, oacc[nb] * inv);
        else *reinterpret_cast<f32x4*>(lp + 16 * nb) = oacc[nb] * inv;
      }
      if (w == 0 && g == 0) {
        if (a.merge_counters) store_dev(a.attn_lse + row, m_run * kLn2 + __logf(l_run));
        else a.attn_lse[row] = m_run * kLn2 + __logf(l_run);
      }
    }
    if (a.merge_counters) mla_merge_if_last<T>(a, b, qb, seq_len, splits);
  }
}

}  // namespace rx

namespace rx {
// called from rx_decode_attn (rx_decode.hip) when head_dim == 576 and v_head_dim == 512, Hkv == 1
// and V aliases K's first 512 columns
int launch_decode_mla(const rx_decode_params* p, int32_t* merge_counters, int direct_single, hipStream_t s) {
  MlaArgs a;
  a.merge_counters = merge_counters;
  a.direct_single = direct_single;
  a.q = (const uint16_t*)p->q;
  a.o = (uint16_t*)p->o;
  a.q_stride_t = p->q_stride_t;
  a.q_stride_h = p->q_stride_h;
  a.o_stride_t = p->o_stride_t;
  a.o_stride_h = p->o_stride_h;
  a.kv_buf = (const uint16_t*)p->kv.k_buf;
  a.page_size = p->kv.page_size;
  if ((a.page_size & (a.page_size - 1)) == 0) a.page_size = -(__builtin_ctz(a.page_size) + 1);
  a.page_stride = p->kv.k_page_stride;
  a.tok_stride = p->kv.k_tok_stride;
  a.kv_indptr = p->kv_indptr;
  a.kv_indices = p->kv_indices;
  a.req_to_token = p->req_to_token;
  a.kv_start = p->kv_indices ? nullptr : p->kv_start;
  a.req_row_stride = p->req_row_stride;
  a.req_pool_indices = p->req_pool_indices;
  a.rpi64 = p->req_pool_indices_is_i64;
  a.seq_lens = p->seq_lens;
  a.sl64 = p->seq_lens_is_i64;
  const int max_splits = p->max_kv_splits < 1 ? 1 : p->max_kv_splits;
  a.num_kv_splits = max_splits > 1 ? p->num_kv_splits : nullptr;
  a.max_kv_splits = max_splits;
  a.attn_logits = p->attn_logits;
  a.attn_lse = p->attn_lse;
  a.bs = p->bs;
  a.hq = p->num_q_heads;
  a.qblocks = (p->num_q_heads + 15) / 16;
  a.sm_scale = p->sm_scale * p->k_scale;
  a.v_scale = p->v_scale;
  a.logit_cap = p->logit_cap;
  a.sinks = p->sinks;
  a.xai_len = p->xai_temperature_len;
  a.rope_cache = p->rope_cos_sin;
  a.rope_f32 = p->rope_cos_sin_is_f32;
  a.rope_stride = p->rope_cos_sin_stride;
  a.rope_pos = p->rope_positions;
  a.rope_pos64 = p->rope_positions_is_i64;
  a.rope_neox = p->rope_is_neox;
  a.kpe_out = (uint16_t*)p->rope_k_pe_out;
  a.kpe_out_stride = p->rope_k_pe_out_stride;
  a.k_new = (const uint16_t*)p->k_new;
  a.k_new_stride = p->k_new_stride_t;
  if (a.rope_cache) {
    RX_REQUIRE(!p->kv.kv_fp8, "rx_decode_attn: fused RoPE reads and rewrites 16-bit latent rows (an fp8 pool: rotate with "
                              "rx_rope_store_kv first)");
    RX_REQUIRE(p->rope_dim == 64 && a.rope_pos && a.rope_stride >= 64,
               "rx_decode_attn: fused RoPE needs rotary_dim 64 (= qk_rope_head_dim), positions and a [max_pos, 64] table");
    RX_REQUIRE(p->stages != 2, "rx_decode_attn: fused RoPE belongs to a call that runs stage 1");
    RX_REQUIRE(!a.kpe_out || (((uintptr_t)a.kpe_out & 15) == 0 && a.kpe_out_stride % 8 == 0),
               "rx_decode_attn: rope_k_pe_out needs 16-byte aligned rows");
    RX_REQUIRE(!a.k_new || (((uintptr_t)a.k_new & 15) == 0 && a.k_new_stride % 8 == 0),
               "rx_decode_attn: k_new needs 16-byte aligned rows");
  } else {
    RX_REQUIRE(!a.k_new, "rx_decode_attn: k_new on the latent kernel is the fused-RoPE form (rope_cos_sin)");
  }
  const bool linear = p->kv.page_size == 1 || p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride;
  const bool idx64 = p->kv_indices != nullptr && p->kv_indices_is_i64;
  const unsigned pairs = static_cast<unsigned>(a.bs) * a.max_kv_splits;
  const unsigned grid = (pairs + 7) / 8 * 8 * a.qblocks;  // whole groups of 8 pairs (one per XCD)
  const bool kv8 = p->kv.kv_fp8 != 0;
  const bool old8 = !options().decode_mla8_dma;  // (A/B switch: the upcast-while-staging form for fp8 rows)
  const bool t64 = kv8 && !old8 && options().decode_mla8_t64;  // the 64-token-tile form (round 5)
  if (t64)
    note_dispatch("decode_mla8_t64_kernel<%s, %s, %s>", p->dtype == RX_BF16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int",
                  tbool(linear));
  else if (kv8 && !old8)
    note_dispatch("decode_mla8_dma_kernel<%s, %s, %s>", p->dtype == RX_BF16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int",
                  tbool(linear));
  else
    note_dispatch("decode_mla_kernel<%s, %s, %s, %s>", p->dtype == RX_BF16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int",
                  tbool(linear), tbool(kv8));
#define RX_MLA_L(TT, IT, LIN)                                                                          \
  do {                                                                                                 \
    if (t64) {                                                                                         \
      auto kern = decode_mla8_t64_kernel<TT, IT, LIN>;                                                 \
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),          \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kT64Lds); \
      (void)attr;                                                                                      \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kT64Lds, s, a);                                  \
    } else if (kv8 && !old8) {                                                                         \
      auto kern = decode_mla8_dma_kernel<TT, IT, LIN>;                                                 \
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),          \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, kM8Lds); \
      (void)attr;                                                                                      \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kM8Lds, s, a);                                   \
    } else if (kv8) {                                                                                  \
      hipLaunchKernelGGL((decode_mla_kernel<TT, IT, LIN, true>), dim3(grid), dim3(256), 0, s, a);      \
    } else {                                                                                           \
      hipLaunchKernelGGL((decode_mla_kernel<TT, IT, LIN, false>), dim3(grid), dim3(256), 0, s, a);     \
    }                                                                                                  \
  } while (0)
#define RX_MLA_GO(TT)                                \
  do {                                               \
    if (idx64) {                                     \
      if (linear) RX_MLA_L(TT, int64_t, true);       \
      else RX_MLA_L(TT, int64_t, false);             \
    } else {                                         \
      if (linear) RX_MLA_L(TT, int32_t, true);       \
      else RX_MLA_L(TT, int32_t, false);             \
    }                                                \
  } while (0)
  if (p->dtype == RX_BF16) RX_MLA_GO(BF16);
  else RX_MLA_GO(F16);
#undef RX_MLA_L
#undef RX_MLA_GO
  return RX_OK;
}
}  // namespace rx
