// K7 at head dims 256 / 256, 192 / 192 and 192 / 128 (Gemma-class heads and the MLA prefill shape; the reference retunes its Triton kernel for 128 < D <= 256 on gfx950,
// kernels/ops/attention/extend_attention.py:66-77) in the form rx_extend_mla.hip arrived at: the 16x16x32 kernel of
// rx_extend_nd.hip holds 16 query rows per wave at Dv = 256 (64 accumulator registers; 32 rows spill 142 registers
// there), so one K / V^T fragment read feeds ONE MFMA and the kernel is LDS-bound at 0.22 of the MFMA peak.  Here
//   * a wave holds 32 rows: 128 accumulator registers pinned in the AGPR half by inline-asm MFMAs, Q (64 registers)
//     and everything else in the 128 VGPRs, two waves per SIMD, eight waves = 256 rows per workgroup;
//   * rows are (token, q head of the kv head's group) pairs, row = token * G + g (GQA packing: short extends still
//     fill the block and a kv head's K / V tiles are staged once for its whole group);
//   * 64-token K and V tiles come by LDS-DMA (no staging registers) into two stages of padded images (two pad
//     chunks per row: conflict-free ds_read_b128 / ds_read_b64_tr_b16, see YGeom); waves 0-3 issue their pieces at the top
//     of an iteration, their SIMD partners 4-7 behind their first QK^T, so one computes while the other sits in the
//     memory queue; slot ids of 256 tokens at a time come into LDS by DMA as well;
//   * thresholded running max (2^8 slack, exact algebra): the 128-register rescale runs on the first tile and almost
//     never again; the mask is one compare + select per score on every tile.
// Causal / non-causal, skip_prefix / skip_extend, LSE, k / v scales, logit cap, sliding window (tiles wholly below a
// workgroup's window are never loaded), 16-bit pools.  Sinks, masks and short rows stay with rx_extend_nd.hip / the
// generic kernel.
#include <type_traits>

#include "rx_common.h"

namespace rx {

#ifndef RX_D256_PD
#define RX_D256_PD 4    // K fragments read ahead of their MFMA
#endif
#ifndef RX_D256_NPRE
#define RX_D256_NPRE 4  // V^T fragment pairs read ahead
#endif
constexpr int kYTT = 64;
constexpr int kYSlotBlock = 256;
constexpr int kYRows = 256;                  // query rows per workgroup
constexpr float kYSlack = 8.0f;
constexpr float kYSumLimit = 4096.0f;  // a lane's partial row sum above this sends the wave to the max-based step

// (Dk, Dv): 256 / 256, and the MLA prefill shape 192 / 128 (qk_nope 128 + rope 64 against v 128) and 192 / 192
template <int DK, int DV>
struct YGeom {
  static constexpr int KCPR = DK * 2 / 16, VCPR = DV * 2 / 16;  // data chunks per row
  // Image rows carry TWO pad chunks (32 B).  ds_read_b128 is served in four 16-lane groups that are NOT contiguous
  // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), so a group mixes rows 0-3 / 12-15 of k-group g with rows
  // 4-11 of g + 1: with one pad chunk (the round-2 layout: 33 / 36 chunks) two of them share a bank -- every K read
  // and every transposed V read took twice its cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.47 measured,
  // 0.50 by the bank rule); with two, rows step 8 banks and both read patterns are conflict-free.
  static constexpr int KC = KCPR + 2;
  static constexpr int VC = VCPR + 2;
  static constexpr int KROW = KC * 16, VROW = VC * 16;
  static constexpr int KPIECES = (kYTT * KC + 63) / 64, VPIECES = (kYTT * VC + 63) / 64;
  static constexpr int KIMG = KPIECES * 1024, VIMG = VPIECES * 1024;
  static constexpr int STAGE = KIMG + VIMG;
  static constexpr int SLOTS_AT = 2 * STAGE;
  static constexpr int BOUNCE_AT = SLOTS_AT + 2 * kYSlotBlock * 4;
  static constexpr int TBL_AT = BOUNCE_AT + 8 * 1024;  // row-pointer tables: 2 tiles x 64 rows x {K ptr, V ptr}
  static constexpr int GEO_AT = TBL_AT + 2 * kYTT * 16;  // per-thread piece geometry (loop invariant), one dword per piece
  static constexpr int GEO_N = (KPIECES + 7) / 8 + ((KC == VC && KCPR == VCPR) ? 0 : (VPIECES + 7) / 8);
  static constexpr int LDS = GEO_AT + GEO_N * 512 * 4;  // 256 / 256: 161792 B
  static_assert(LDS <= 160 * 1024, "image geometry");
};

struct ExtD256Args {
  const uint16_t* q;
  const uint16_t* k_ext;
  const uint16_t* v_ext;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t, o_stride_h;
  const uint16_t* k_buf;
  const uint16_t* v_buf;
  int32_t page_shift;  // log2(page_size), or -1 for a pool that is linear in the slot
  int64_t k_page_stride, k_tok_stride, k_head_stride, v_page_stride, v_tok_stride, v_head_stride;
  const void* qo_indptr;
  int32_t qo64;
  const int32_t* kv_indptr;
  const void* kv_indices;
  int32_t idx64;
  float* lse;
  int64_t lse_stride_t, lse_stride_h;
  int32_t bs, hkv, group, mblocks;
  float sm_scale, k_scale, v_scale, logit_cap;
  int32_t causal, skip_prefix, skip_extend, window;  // window <= 0: off
  const float* sinks;  // fp32 [Hq] or null
};

typedef __attribute__((address_space(3))) const u32x4* y_lds_u32x4;
typedef __attribute__((address_space(3))) const int32_t* y_lds_i32;
__device__ __forceinline__ u32x4 y_lds_read16(uint32_t addr) { return *reinterpret_cast<y_lds_u32x4>(addr); }
__device__ __forceinline__ int32_t y_lds_read4(uint32_t addr) { return *reinterpret_cast<y_lds_i32>(addr); }
// M0 is set and not restored (see rx_extend_mla.hip: hipcc uses M0 for nothing else in this kernel)
__device__ __forceinline__ void y_dma4(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void y_dma16(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <typename T>
__device__ __forceinline__ void y_pv_mfma(u32x4 a, u32x4 b, f32x4& c) {
  if constexpr (std::is_same_v<T, BF16>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// S^T accumulators pinned in VGPRs: left to itself hipcc puts the builtin's result in the AGPR half -- which
// the 128 O^T accumulators fill -- and moves 16 of THEM out and back around every half tile's scores (104 v_accvgpr_read +
// 72 v_accvgpr_write per tile and wave in the loop's ISA).  hipcc pads nothing around asm MFMAs: the four chains
// (2 token blocks x 2 row blocks) are issued round robin, so a dependent MFMA follows its predecessor by four 8-pass
// instructions, and the scores are read by VALU only behind y_scores_ready().
template <typename T>
__device__ __forceinline__ void y_qk_mfma0(u32x4 a, u32x4 b, f32x4& c) {
  if constexpr (std::is_same_v<T, BF16>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}
template <typename T>
__device__ __forceinline__ void y_qk_mfma(u32x4 a, u32x4 b, f32x4& c) {
  if constexpr (std::is_same_v<T, BF16>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void y_scores_ready(f32x4 (&sacc)[2][2]) {
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // XDL write -> VALU read of an 8-pass result: 11 wait states
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int b = 0; b < 2; ++b) asm volatile("" : "+v"(sacc[c][b]));
}
template <int N>
__device__ __forceinline__ void y_settle(f32x4 (&o)[N]) {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+a"(o[i]));
}

// EXTRAS: logit cap / sliding window compiled in (the plain instance loses 8-9 % to their scalars and branches even
// when both are off: 961 vs 881 TFLOP/s at 256 / 256)
template <typename T, int DK, int DV, bool EXTRAS>
__global__ __launch_bounds__(512, 1) void extend_d256_kernel(const ExtD256Args a) {
  using vec8 = typename T::vec8;
  using Y = YGeom<DK, DV>;
  constexpr int KS = DK / 32, NB = DV / 16;  // k-steps, d-blocks
  constexpr int kYKrow = Y::KROW, kYVrow = Y::VROW, kYKimg = Y::KIMG, kYStage = Y::STAGE, kYSlotsAt = Y::SLOTS_AT,
                kYBounceAt = Y::BOUNCE_AT, kYKc = Y::KC, kYVc = Y::VC, kYKpieces = Y::KPIECES, kYVpieces = Y::VPIECES;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages: K image | V image][2 slot blocks]
  const uint32_t smem_u = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // kv head = block mod Hkv: a kv head's prefix rows stay in one XCD's L2 (blocks go to the XCDs round robin)
  int bid = blockIdx.x;
  const int kvh = bid % a.hkv;
  bid /= a.hkv;
  const int mb = a.mblocks - 1 - bid % a.mblocks;  // heaviest query blocks first under the causal mask
  const int req = bid / a.mblocks;

  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t G = a.group;
  const int32_t R = E * G;  // query rows of (request, kv head): row = token * G + g
  const int32_t row0 = mb * kYRows;
  if (row0 >= R) return;    // workgroup-uniform
  const int32_t rbase = row0 + 32 * w;
  const bool active = rbase < R;

  // row -> token by multiply-high.  G = 1 (MHA: Gemma-7B-class heads) has no 32-bit magic -- 2^32 / 1 wraps to 0 and every
  // row would map to token 0 (found in round 3 with the D = 64 instance; the round-2 kernel had this bug for MHA models
  // at 256 / 192 with extends past 128 tokens) -- so it is the identity there.
  const uint32_t g_magic = G == 1 ? 0u : static_cast<uint32_t>(0x100000000ull / static_cast<uint32_t>(G)) + 1u;
  auto row_tok = [&](int m) { return G == 1 ? m : static_cast<int32_t>(__umulhi(static_cast<uint32_t>(m), g_magic)); };

  // ---- Q^T fragments: block c, lane (r, g) holds Q[row rbase + 16 c + r][32 s + 8 g .. +8]
  vec8 qf[2][KS];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int m = rbase + 16 * c + r;
    const bool ok = m < R;
    const int32_t tk = row_tok(ok ? m : 0);
    const int32_t hd = kvh * G + ((ok ? m : 0) - tk * G);
    const uint16_t* qp = a.q + (qo0 + tk) * a.q_stride_t + hd * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[c][s] = __builtin_bit_cast(vec8, raw);
    }
  }

  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t tok_hi_wg = (min(R, row0 + kYRows) - 1) / G + 1;
  const int32_t tok_hi_w = active ? (min(R, rbase + 32) - 1) / G + 1 : 0;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? tok_hi_wg : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? tok_hi_w : E);
  const int nt1 = (p_len + kYTT - 1) / kYTT;
  const int nt2 = (n_end_wg + kYTT - 1) / kYTT;
  const int nt = nt1 + nt2;
  // sliding window (extend_attention.py:385-390, 556-561): query token m sees cached token n iff P + m <= n + W and
  // new token n iff m <= n + W.  Tiles wholly below the workgroup's first row's bound are never loaded.
  const bool windowed = EXTRAS && a.window > 0;
  const int32_t tok_lo_wg = row0 / G, tok_lo_w = rbase / G;
  int t_begin = 0;
  if (windowed) {
    t_begin = min(nt1, max(0, P + tok_lo_wg - a.window) / kYTT);
    if (t_begin == nt1) t_begin += min(nt2, max(0, tok_lo_wg - a.window) / kYTT);
  }
  const bool capped = EXTRAS && a.logit_cap > 0.f;

  const char* const idx_b = reinterpret_cast<const char*>(a.kv_indices);
  const int idx_sh = a.idx64 ? 3 : 2;
  auto stage_slots = [&](int blk) {  // waves 0-3: 256 slot ids by DMA (no compiler-visible VMEM in the loop)
    if (w < 4) {
      const int v = blk * kYSlotBlock + tid;
      const int64_t e = kv0 + max(min(v, p_len - 1), 0);
      y_dma4(idx_b + (e << idx_sh), __builtin_amdgcn_readfirstlane(smem_u + kYSlotsAt + ((blk & 1) * kYSlotBlock + 64 * w) * 4));
    }
  };
  constexpr int TPB = kYSlotBlock / kYTT;
  if (t_begin < nt1) {  // the slot block of the first tile (and the next one: the loop stages block b + 1 at tile b * TPB)
    stage_slots(t_begin / TPB);
    if (t_begin % TPB != 0 && (t_begin / TPB + 1) * kYSlotBlock < nt1 * kYTT) stage_slots(t_begin / TPB + 1);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and the slot ids have landed
  __syncthreads();

  const char* const kbuf_b = reinterpret_cast<const char*>(a.k_buf + kvh * a.k_head_stride);
  const char* const vbuf_b = reinterpret_cast<const char*>(a.v_buf + kvh * a.v_head_stride);
  const char* const kext_b = reinterpret_cast<const char*>(a.k_ext + qo0 * a.k_stride_t + kvh * a.k_stride_h);
  const char* const vext_b = reinterpret_cast<const char*>(a.v_ext + qo0 * a.v_stride_t + kvh * a.v_stride_h);
  const int32_t sh_p = a.page_shift < 0 ? 31 : a.page_shift;
  // ---- table-driven issue.  The ISA of the per-piece form above was a third of the loop's instruction stream (per
  // piece: slot read + wait, page / token split, two 64-bit multiplies, ~12 VALU + 10 SALU + two branches), and this
  // kernel pays ~5 cycles for EVERY instruction a wave issues (rx_extend_d256.hip header, round 3).  One wave per tile
  // writes the 64 rows' K and V pointers to LDS two tiles ahead (lane = row); a piece is then: its (row, column) by
  // three adds, one ds_read_b64, one 64-bit add, the DMA.
  constexpr int kYTblAt = Y::TBL_AT;
  auto build_table = [&](int u) {  // wave u & 7, lane = row of tile u
    if (w != (u & 7)) return;
    const uint32_t dst = smem_u + kYTblAt + ((u & 1) * kYTT + lane) * 16;
    const char* kp;
    const char* vp;
    if (u < nt1) {
      const uint32_t sl = smem_u + kYSlotsAt + 4 * (((u * kYTT / kYSlotBlock) & 1) * kYSlotBlock + (u * kYTT) % kYSlotBlock + lane);
      const uint32_t slot = static_cast<uint32_t>(y_lds_read4(sl));
      const uint32_t lo = sh_p == 31 ? slot : (slot & ((1u << sh_p) - 1u));
      uint64_t ko = static_cast<uint64_t>(lo) * (2u * static_cast<uint32_t>(a.k_tok_stride));
      uint64_t vo = static_cast<uint64_t>(lo) * (2u * static_cast<uint32_t>(a.v_tok_stride));
      if (sh_p != 31) {
        ko += static_cast<uint64_t>(slot >> sh_p) * (2u * static_cast<uint32_t>(a.k_page_stride));
        vo += static_cast<uint64_t>(slot >> sh_p) * (2u * static_cast<uint32_t>(a.v_page_stride));
      }
      kp = kbuf_b + ko;
      vp = vbuf_b + vo;
    } else {
      const uint32_t n = static_cast<uint32_t>(max(min((u - nt1) * kYTT + lane, n_end_wg - 1), 0));
      kp = kext_b + static_cast<uint64_t>(n) * (2u * static_cast<uint32_t>(a.k_stride_t));
      vp = vext_b + static_cast<uint64_t>(n) * (2u * static_cast<uint32_t>(a.v_stride_t));
    }
    const uint64_t k64 = reinterpret_cast<uint64_t>(kp), v64 = reinterpret_cast<uint64_t>(vp);
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(dst) =
        u32x4{static_cast<uint32_t>(k64), static_cast<uint32_t>(k64 >> 32), static_cast<uint32_t>(v64), static_cast<uint32_t>(v64 >> 32)};
  };
  // a lane's chunk positions are loop invariant: piece w + 8 i holds chunk 64 w + lane + 512 i of the padded image; its
  // (row * 16 | column * 16 << 16) is computed once per image geometry and kept in registers (RX_D256_GEO)
  auto piece_geo = [&](auto cpr_c, auto np_c, auto data_c, uint32_t (&geo)[(decltype(np_c)::value + 7) / 8]) {
    constexpr int CPR = decltype(cpr_c)::value, NP = (decltype(np_c)::value + 7) / 8, DATA = decltype(data_c)::value;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int c0 = 64 * w + ln;
    int row = c0 / CPR, col = c0 - row * CPR;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      // (the last piece's tail repeats row 63; pad chunks re-read the row's last data chunk)
      geo[i] = 16u * static_cast<uint32_t>(min(row, kYTT - 1)) | (16u * static_cast<uint32_t>(min(col, DATA - 1))) << 16;
      row += 512 / CPR;
      col += 512 % CPR;
      if (col >= CPR) {
        col -= CPR;
        row += 1;
      }
    }
  };
  // ... and kept in LDS (kNPK [+ kNPV] dwords per thread): in registers they cost the 256 / 256 instance four Q fragments
  // (32 spilled dwords, eight scratch reloads per tile)
  constexpr int kNPK = (kYKpieces + 7) / 8, kNPV = (kYVpieces + 7) / 8;
  constexpr bool kSameGeo = (kYKc == kYVc) && (Y::KCPR == Y::VCPR);
  constexpr int kYGeoAt = Y::GEO_AT;
  {
    uint32_t geo_k[kNPK];
    piece_geo(std::integral_constant<int, kYKc>{}, std::integral_constant<int, kYKpieces>{},
              std::integral_constant<int, Y::KCPR>{}, geo_k);
#pragma unroll
    for (int i = 0; i < kNPK; ++i)
      *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(smem_u + kYGeoAt + (i * 512 + tid) * 4) = geo_k[i];
    if constexpr (!kSameGeo) {
      uint32_t geo_v[kNPV];
      piece_geo(std::integral_constant<int, kYVc>{}, std::integral_constant<int, kYVpieces>{},
                std::integral_constant<int, Y::VCPR>{}, geo_v);
#pragma unroll
      for (int i = 0; i < kNPV; ++i)
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(smem_u + kYGeoAt + ((kNPK + i) * 512 + tid) * 4) = geo_v[i];
    }
  }  // (read back by the same thread only: no barrier needed, the LDS queue is in order per wave)
  typedef __attribute__((address_space(3))) const u32x2* lds_u32x2;
  typedef __attribute__((address_space(3))) const uint32_t* lds_u32;
  auto issue_pieces = [&](int t, auto np_c, const uint32_t (&geo)[(decltype(np_c)::value + 7) / 8], bool vside) {
    constexpr int NPIECES = decltype(np_c)::value;
    constexpr int NP = (NPIECES + 7) / 8;
    const uint32_t tb = smem_u + kYTblAt + (t & 1) * (kYTT * 16) + (vside ? 8 : 0);  // 1 KiB aligned (+8): OR-able
    const uint32_t img = smem_u + (t & 1) * kYStage + (vside ? kYKimg : 0);
    u32x2 ptr[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) ptr[i] = *reinterpret_cast<lds_u32x2>(tb | (geo[i] & 0xffffu));
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (w + 8 * i < NPIECES) {  // wave-uniform
        const uint64_t src = (static_cast<uint64_t>(ptr[i][1]) << 32 | ptr[i][0]) + (geo[i] >> 16);
        y_dma16(reinterpret_cast<const void*>(src), __builtin_amdgcn_readfirstlane(img + (w + 8 * i) * 1024));
      }
    }
  };
  auto dma_tile_tbl = [&](int t) {
    uint32_t gk[kNPK];
#pragma unroll
    for (int i = 0; i < kNPK; ++i) gk[i] = *reinterpret_cast<lds_u32>(smem_u + kYGeoAt + (i * 512 + tid) * 4);
    issue_pieces(t, std::integral_constant<int, kYKpieces>{}, gk, false);
    if constexpr (kSameGeo) {
      issue_pieces(t, std::integral_constant<int, kYVpieces>{}, gk, true);
    } else {
      uint32_t gv[kNPV];
#pragma unroll
      for (int i = 0; i < kNPV; ++i) gv[i] = *reinterpret_cast<lds_u32>(smem_u + kYGeoAt + ((kNPK + i) * 512 + tid) * 4);
      issue_pieces(t, std::integral_constant<int, kYVpieces>{}, gv, true);
    }
  };
  auto dma_tile = [&](int t) {
    dma_tile_tbl(t);
  };
  if (t_begin < nt) build_table(t_begin);
  if (t_begin + 1 < nt) build_table(t_begin + 1);
  __syncthreads();  // the first two tables are readable
  if (t_begin < nt) dma_tile(t_begin);

  f32x4 oacc[2][NB];
  float m_run[2], l_run[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    m_run[c] = -INFINITY;
    l_run[c] = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) oacc[c][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int qd = r >> 2, pp = r & 3;
  const uint32_t k_lane = r * kYKrow + g * 16;
  const uint32_t v_lane = (4 * g + qd) * kYVrow + 8 * (pp & 1) + (pp >> 1) * 16;
  const uint32_t bounce = smem_u + kYBounceAt + (w * 64 + lane) * 16;  // this lane's 16 bytes of the rescale bounce
  const bool late = w >= 4;  // the SIMD partner of an early wave: issues its pieces behind its first QK^T

  for (int t = t_begin; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t have landed
    __syncthreads();                                   // everybody's have; everybody is done with tile t - 1
    if (t % TPB == 0 && (t / TPB + 1) * kYSlotBlock < nt1 * kYTT) stage_slots(t / TPB + 1);
    if (t + 2 < nt) build_table(t + 2);  // read by tile t + 2's issue, which runs behind the next barrier
    const bool more = t + 1 < nt;
    if (more && !late) dma_tile(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    const bool prefix = t < nt1;
    const int tile_n0 = (prefix ? t : t - nt1) * kYTT;
    const int32_t lim = prefix ? p_len : n_end_w;
    bool late_pending = more && late;
    // (a window: the tile may lie wholly below this wave's first row's bound)
    const int32_t win_lo_w = windowed ? (prefix ? P : 0) + tok_lo_w - a.window : INT32_MIN;
    if (!active || tile_n0 >= lim || tile_n0 + kYTT <= win_lo_w) {
      if (late_pending) dma_tile(t + 1);
      continue;
    }
    const uint32_t kt = smem_u + (t & 1) * kYStage;
    const uint32_t vt = kt + kYKimg;
    const float cs = prefix ? a.sm_scale * a.k_scale : a.sm_scale;
    const float c2 = capped ? kLog2e : cs * kLog2e;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int n0 = tile_n0 + 32 * hh;
      if (n0 >= lim || n0 + 32 <= win_lo_w) continue;  // nothing visible to this wave in this half (wave-uniform)
      const bool half_full = !windowed && n0 + 32 <= (prefix ? p_len : min(n_end_w, a.causal ? tok_lo_w + 1 : E));
      // ---- S^T = K Q^T: tokens 16 bb + 4 g + i of the half on the lane, query row r of block c
      f32x4 sacc[2][2];
      {
        // i = 2 s + blk: token blocks alternate, so the four accumulator chains are issued round robin
        constexpr int PD = RX_D256_PD;
        const uint32_t krow = kt + k_lane + 32 * hh * kYKrow;
        auto kfrag = [&](int i) { return y_lds_read16(krow + (i & 1) * 16 * kYKrow + (i >> 1) * 64); };
        u32x4 kf[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) kf[i] = kfrag(i);
#pragma unroll
        for (int i2 = 0; i2 < 2 * KS; i2 += 2) {
          // fragments in pairs, the LATER one first: LDS returns in order, so its s_waitcnt covers both -- one wait
          // instruction per two fragments (every instruction is ~5 cycles here; the two token blocks are different
          // accumulators, so the sums are unchanged)
#pragma unroll
          for (int j = 1; j >= 0; --j) {
            const int i = i2 + j;
            const u32x4 ka = kf[i % PD];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (i < 2) y_qk_mfma0<T>(ka, __builtin_bit_cast(u32x4, qf[c][0]), sacc[c][i & 1]);
              else y_qk_mfma<T>(ka, __builtin_bit_cast(u32x4, qf[c][i >> 1]), sacc[c][i & 1]);
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (i2 + j + PD < 2 * KS) kf[(i2 + j) % PD] = kfrag(i2 + j + PD);
          __builtin_amdgcn_sched_barrier(0);  // source order is the pipeline
        }
        y_scores_ready(sacc);
      }
      if (hh == 0 && late_pending) {
        dma_tile(t + 1);
        late_pending = false;
      }
      __builtin_amdgcn_sched_barrier(0);
      u32x4 pf[2];
      float alpha_c[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float sv[8];
        // the block's 8 scores per lane as the softmax takes them: capped, masked
        auto load_sv = [&]() {
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[c][bb][i];
          if (capped) {  // logit cap (wave-uniform branch): cap * tanh(s * scale / cap)
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = a.logit_cap * tanhf(sv[j] * cs / a.logit_cap);
          }
          // (a half that every row of the wave sees in full -- the whole prefix but its ragged end, the new tokens below
          // the wave's first row -- takes no mask: 2 of its ~10 VALU per score; wave-uniform branch)
          if (!half_full) {
            int lnm = lane;
            asm volatile("" : "+v"(lnm));
            const int32_t tk1 = row_tok(rbase + 16 * c + (lnm & 15)) + 1;
            const int32_t vis = (prefix ? p_len : min(n_end_w, a.causal ? tk1 : E)) - n0 - 4 * (lnm >> 4);  // visible: index < vis
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = (16 * bb + i < vis) ? sv[bb * 4 + i] : -INFINITY;
            if (windowed) {  // ... and index >= the row's window bound (wave-uniform branch)
              const int32_t wlo = (prefix ? P : 0) + tk1 - 1 - a.window - n0 - 4 * (lnm >> 4);
#pragma unroll
              for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = (16 * bb + i >= wlo) ? sv[bb * 4 + i] : -INFINITY;
            }
          }
        };
        load_sv();
        // Softmax WITHOUT a row maximum on the common path (round 4, as rx_extend32_kernel.inc): the scores are
        // exponentiated against the STANDING running max, and the lane's partial row sum is the check -- every p is <= it,
        // so while it stays <= kYSumLimit nothing overflowed or lost precision against the running scale.  Only when a
        // lane's sum runs away (or is NaN: the first tile's m = -inf) the wave takes the max-based step and redoes the
        // block.  Saves the 7-max chain, the cross-lane quad max, the select and the alpha exponential per 8 scores.
        float alpha = 1.0f, psum = 0.f;
        {
          const float m_old = m_run[c];
          float e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            e[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_old));
            psum += e[j];
          }
          // (compared as BITS: the sum is never negative, so the unsigned order is the float order with +inf and every
          // NaN on top -- the file is built with -fno-honor-nans, under which !(x <= limit) need not catch a NaN, and a
          // masked score against m = -inf is exactly that)
          if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, kYSumLimit)) != 0) {
            load_sv();
            float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
            mt = quad_row_max(mt) * c2;
            const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;  // extend_attention.py:474-475
            const float m_new = (mt_fixed > m_old + kYSlack) ? mt_fixed : m_old;
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
        if (EXTRAS && prefix && a.v_scale != 1.0f) {  // (a scaled V pool takes the EXTRAS instance)
#pragma unroll
          for (int j = 0; j < 8; ++j) sv[j] *= a.v_scale;
        }
        pf[c][0] = pack2<T>(sv[0], sv[1]);
        pf[c][1] = pack2<T>(sv[2], sv[3]);
        pf[c][2] = pack2<T>(sv[4], sv[5]);
        pf[c][3] = pack2<T>(sv[6], sv[7]);
        alpha_c[c] = alpha;
        __builtin_amdgcn_sched_barrier(0);
      }
      // ONE rescale branch behind both blocks' softmax (first tile; then only on a 2^8 jump): inside the block loop
      // the branch is the kernel's register peak (53 spilled registers)
      if (__builtin_amdgcn_ballot_w64(alpha_c[0] != 1.0f || alpha_c[1] != 1.0f) != 0) {
        // The accumulators never pass through compiler-visible code: any C++ access makes hipcc route their live ranges
        // through the VGPR half (53 spilled registers, Q fragments reloaded every half tile).  They bounce through one
        // KiB of LDS instead -- ds_write from / ds_read into the AGPRs in asm, the multiply on a VGPR copy in between.
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // XDL write -> LDS read of the accumulators
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            asm volatile("ds_write_b128 %0, %1" : : "v"(bounce), "a"(oacc[c][nb]) : "memory");
            f32x4 tv = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>(bounce);
            tv *= alpha_c[c];
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(bounce) = tv;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=a"(oacc[c][nb]) : "v"(bounce) : "memory");
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- O^T += V^T P^T
      const uint32_t rp0 = vt + v_lane + 32 * hh * kYVrow;
      const uint32_t rp1 = rp0 + 16 * kYVrow;
      constexpr int NPRE = RX_D256_NPRE;
      u32x2 vlo[NPRE], vhi[NPRE];
#pragma unroll
      for (int nb = 0; nb < NPRE; ++nb) {
        vlo[nb] = T::ds_read_tr((const void*)(uintptr_t)(rp0 + nb * 32));
        vhi[nb] = T::ds_read_tr((const void*)(uintptr_t)(rp1 + nb * 32));
      }
#pragma unroll
      for (int nb2 = 0; nb2 < NB; nb2 += 2) {  // (d blocks in pairs, the later pair of reads first: see the K fragments)
#pragma unroll
        for (int j = 1; j >= 0; --j) {
          const int nb = nb2 + j;
          const u32x2 lo = vlo[nb % NPRE], hi = vhi[nb % NPRE];
          const u32x4 av = u32x4{lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
          for (int c = 0; c < 2; ++c) y_pv_mfma<T>(av, pf[c], oacc[c][nb]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nb = nb2 + j;
          if (nb + NPRE < NB) {
            vlo[nb % NPRE] = T::ds_read_tr((const void*)(uintptr_t)(rp0 + (nb + NPRE) * 32));
            vhi[nb % NPRE] = T::ds_read_tr((const void*)(uintptr_t)(rp1 + (nb + NPRE) * 32));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (late_pending) dma_tile(t + 1);  // (the first half had nothing visible)
  }

  if (!active) return;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    y_settle(oacc[c]);
    float l = l_run[c];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int m = rbase + 16 * c + r;
    if (m >= R) continue;
    const int32_t tk = m / G, hd = kvh * G + (m - tk * G);
    float den = l;
    if (a.sinks) den += fast_exp2(a.sinks[hd] * kLog2e - m_run[c]);  // attention sinks join the softmax sum (extend_attention.py:633-635)
    const float inv = 1.0f / den;
    uint16_t* op = a.o + (qo0 + tk) * a.o_stride_t + hd * a.o_stride_h + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x2 pk;
      pk[0] = pack2<T>(oacc[c][nb][0] * inv, oacc[c][nb][1] * inv);
      pk[1] = pack2<T>(oacc[c][nb][2] * inv, oacc[c][nb][3] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.lse && g == 0) a.lse[(qo0 + tk) * a.lse_stride_t + hd * a.lse_stride_h] = m_run[c] * kLn2 + __logf(l);
  }
}

// what the kernel serves: head dims 256 / 256 (192, 96, 64) on a 16-bit pool, no tree mask / unified form / Grok temperature, aligned
// tensors, and extends long enough to fill 256-row workgroups (short ones: rx_extend_nd.hip's smaller blocks)
bool extend_d256_supports(const rx_extend_params* p) {
  const int dk = p->head_dim, dv = p->v_head_dim;
  // (128 / 128 as well, on request -- option extend_d256_at128: this kernel form against the 32x32x16 kernel of
  // rx_extend32.hip on the headline shape.)  D = 64 takes this form by default since round 3 (588-609 TFLOP/s at the
  // config-3 chunk against 533 for the 16x16 kernel of rx_extend.hip with 64 queries per wave) and so does D = 96;
  // options extend_d256_at64 / extend_d256_at96 = 0 turn that off.
  const Options& opt = options();
  const bool at128 = opt.extend_d256_at128 != 0, at64 = opt.extend_d256_at64 != 0, at96 = opt.extend_d256_at96 != 0;
  if (!((dk == 256 && dv == 256) || (dk == 192 && (dv == 128 || dv == 192)) || (at128 && dk == 128 && dv == 128) ||
        (at64 && dk == 64 && dv == 64) || (at96 && dk == 96 && dv == 96)) || p->kv.kv_fp8)
    return false;
  if (p->custom_mask || p->xai_temperature_len > 0 || p->unified_prefix_lens || p->q_pack > 1 || p->window_kv_offsets)
    return false;
  const int64_t all = p->q_stride_t | p->q_stride_h | p->k_stride_t | p->k_stride_h | p->v_stride_t | p->v_stride_h |
                      p->kv.k_page_stride | p->kv.k_tok_stride | p->kv.k_head_stride | p->kv.v_page_stride |
                      p->kv.v_tok_stride | p->kv.v_head_stride;
  if (all % 8 != 0 || (p->o_stride_t | p->o_stride_h) % 4 != 0) return false;
  if ((((uintptr_t)p->q | (uintptr_t)p->k_extend | (uintptr_t)p->v_extend | (uintptr_t)p->kv.k_buf | (uintptr_t)p->kv.v_buf) & 15) != 0 ||
      ((uintptr_t)p->o & 7) != 0)
    return false;
  const bool linear = p->kv.page_size == 1 || (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                                               p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  if (!linear && (p->kv.page_size & (p->kv.page_size - 1)) != 0) return false;
  const int64_t group = p->num_q_heads / p->num_kv_heads;
  if (static_cast<int64_t>(p->max_extend_len + 1) * group * group >= (1ll << 31)) return false;  // row -> token by multiply-high
  // Rows of the longest request.  Until round 4 only calls with more than half a workgroup's rows (128) came here and
  // short extends ran one workgroup per (request, q head) on rx_extend_nd.hip / rx_extend.hip; with the rows of a kv
  // head's group packed into one block the template wins from a few rows up wherever there is a prefix to walk
  // (tools/probe/short_ext.py, 64 requests, TFLOP/s before / after: D 256, 1 k + 32 tokens 227 / 652; 2 k + 16: 126 / 410;
  // 4 k + 8: 64 / 230; D 64, 2 k + 32: 214 / 541; 192 / 128, 2 k + 16: 86 / 317; D 96, 2 k + 8: 36 / 220) and loses only
  // where a workgroup walks next to nothing (no prefix + 32 tokens: 42 / 35) -- hence the tile estimate below.
  const int64_t rows = static_cast<int64_t>(p->max_extend_len) * group;
  if (rows > 128) return true;
  return rows >= opt.extend_d256_min_rows && (p->avg_kv_len_hint + p->max_extend_len / 2) / 64 >= 4;
}

int launch_extend_d256(const rx_extend_params* p, hipStream_t s) {
  ExtD256Args a;
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
  const bool linear = p->kv.page_size == 1 || (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                                               p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  a.page_shift = linear ? -1 : __builtin_ctz(p->kv.page_size);
  a.k_page_stride = p->kv.k_page_stride; a.k_tok_stride = p->kv.k_tok_stride; a.k_head_stride = p->kv.k_head_stride;
  a.v_page_stride = p->kv.v_page_stride; a.v_tok_stride = p->kv.v_tok_stride; a.v_head_stride = p->kv.v_head_stride;
  a.qo_indptr = p->qo_indptr; a.qo64 = p->qo_indptr_is_i64;
  a.kv_indptr = p->kv_indptr; a.kv_indices = p->kv_indices; a.idx64 = p->kv_indices_is_i64;
  a.lse = p->lse; a.lse_stride_t = p->lse_stride_t; a.lse_stride_h = p->lse_stride_h;
  a.bs = p->bs; a.hkv = p->num_kv_heads;
  a.group = p->num_q_heads / p->num_kv_heads;
  a.mblocks = static_cast<int32_t>((static_cast<int64_t>(p->max_extend_len) * a.group + kYRows - 1) / kYRows);
  a.sm_scale = p->sm_scale; a.k_scale = p->k_scale; a.v_scale = p->v_scale; a.logit_cap = p->logit_cap;
  a.causal = p->is_causal; a.skip_prefix = p->skip_prefix; a.skip_extend = p->skip_extend;
  a.window = p->sliding_window_size;
  a.sinks = p->sinks;
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hkv * a.mblocks;
#define RX_D256(TT, DK_, DV_, EX_)                                                                                     \
  do {                                                                                                               \
    constexpr int lds_ = YGeom<DK_, DV_>::LDS;                                                                       \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(extend_d256_kernel<TT, DK_, DV_, EX_>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_);            \
    (void)attr;                                                                                                      \
    hipLaunchKernelGGL((extend_d256_kernel<TT, DK_, DV_, EX_>), dim3(grid), dim3(512), lds_, s, a);                   \
  } while (0)
#define RX_D256_DIMS(TT, EX_)                                 \
  do {                                                        \
    if (dk == 64) RX_D256(TT, 64, 64, EX_);                   \
    else if (dk == 96) RX_D256(TT, 96, 96, EX_);              \
    else if (dk == 128) RX_D256(TT, 128, 128, EX_);           \
    else if (dk == 256) RX_D256(TT, 256, 256, EX_);           \
    else if (dv == 128) RX_D256(TT, 192, 128, EX_);           \
    else RX_D256(TT, 192, 192, EX_);                          \
  } while (0)
  const bool bf = p->dtype == RX_BF16;
  const int dk = p->head_dim, dv = p->v_head_dim;
  const bool extras = a.window > 0 || a.logit_cap > 0.f || a.v_scale != 1.0f;
  note_dispatch("extend_d256_kernel<%s, %d, %d, %s>|%s,g%d", bf ? "rx::BF16" : "rx::F16", dk, dv, tbool(extras),
                linear ? "linear" : "paged", a.group);
  if (bf) {
    if (extras) RX_D256_DIMS(BF16, true);
    else RX_D256_DIMS(BF16, false);
  } else {
    if (extras) RX_D256_DIMS(F16, true);
    else RX_D256_DIMS(F16, false);
  }
#undef RX_D256_DIMS
#undef RX_D256
  return RX_OK;
}

}  // namespace rx
