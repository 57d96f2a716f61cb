// K7 for the latent (absorbed) MLA shape: q [T, Hq, 576] against ONE kv head whose rows are the 576-wide latent
// (kv_lora_rank 512 + rope 64) and whose values are the first 512 columns of the same row.  This is what an extend
// over a cached prefix runs at for DeepSeek-class models on the reference's backend (forward_absorb_core ->
// attn_mqa -> TritonAttnBackend.forward_extend, triton_backend.py:1290-1437; kernel extend_attention.py:241-661
// with Lq = 576, Lv = 512): every radix-cache hit of an MLA model takes it.  Until round 2 it ran the scalar
// generic kernel here (6 TFLOP/s).
//
// Shape of the problem on gfx950: 2176 FLOP per (query row, token) and only 8 softmax values per lane and 16 x 32
// block, so VALU does not matter; what does is registers, LDS bandwidth and what a tile's loads cost the wave that
// issues them.  With 16x16x32 MFMAs an O^T block of 16 rows x 512 columns is 128 accumulator registers:
//   * a wave holds 16 query rows: 128 accumulators pinned in the AGPR half by inline-asm MFMAs (hipcc left alone moves
//     them through v_accvgpr copies), the 18 Q fragments (72 registers) and everything VALU touches in the 128 VGPRs;
//     TWO waves per SIMD, eight waves = 128 rows per workgroup;
//   * rows are (token, q head) pairs, row = token * Hq + head -- all heads share the one kv head, so packing costs
//     nothing and a request with 8 new tokens still fills a block; a request's workgroups are bound to one XCD's L2;
//   * tiles of 32 tokens come global -> LDS by `global_load_lds_dwordx4` (lane l's 16 bytes land at base + 16 l, the
//     per-lane source address does the row gather) into rows padded to 73 chunks: K fragments by ds_read_b128, V^T
//     fragments by ds_read_b64_tr_b16 FROM THE SAME IMAGE when v aliases k[..., :512] (the pool always does:
//     get_value_buffer is a view of the latent buffer), from a second image of the stage for tiles whose v is a tensor
//     of its own (the new tokens' k is a fresh concat in the reference's model code);
//   * a 1-KiB DMA piece costs the ISSUING wave ~300 cycles (tools/probe/dma_issue.hip: a wave's vector-memory path
//     moves ~3.6 B per cycle whatever the instruction), so waves 0-3 issue their 5 pieces at the top of an iteration
//     and their SIMD partners 4-7 behind their QK^T: one computes while the other sits in the memory queue.  (First
//     form of this kernel: 32 rows per wave, one wave per SIMD, half the LDS reads per FLOP -- and 2.7 k of a tile's
//     7.4 k cycles in DMA issue with nothing on the SIMD to hide them: 650 TFLOP/s against 750 here.  DESIGN 4.2b.)
//   * two stages, `s_waitcnt vmcnt(0)` + one barrier per tile; slot ids of 256 tokens at a time come into LDS by DMA
//     as well, so the loop has no compiler-visible VMEM;
//   * the running max moves only when a tile's max exceeds it by 2^8 (exact algebra, rx_extend32.hip), and the rescale
//     then goes through LDS: ds_write from / ds_read into the AGPRs in asm, the multiply on a VGPR copy in between.
//     Any C++ access to the accumulators makes hipcc route their live ranges through the VGPR half -- with
//     `oacc[nb] *= alpha` in this rarely taken branch only 8 of the 18 Q fragments fit in registers;
//   * the mask is one compare + select per score on every tile (8 per lane against 68 MFMAs): no second body.
// Causal / non-causal, skip_prefix / skip_extend, LSE, k / v scales.  Windows, caps, sinks, masks: generic kernel.
#include <type_traits>

#include "rx_common.h"

namespace rx {

constexpr int kXDk = 576, kXDv = 512, kXTT = 32;
// 74 chunks per LDS row (72 data + 2 pad): with ONE pad chunk (rounds 1-2) the non-contiguous 16-lane groups of
// ds_read_b128 (MI355X_MICROARCH.md, LDS) and the 32-lane halves of the transposed V reads both put two rows on one bank
constexpr int kXCpr = kXDk * 2 / 16 + 2;
constexpr int kXRow = kXCpr * 16;               // 1184 B
constexpr int kXPieces = (kXTT * kXCpr + 63) / 64;  // 37 1-KiB DMA pieces per image
constexpr int kXImg = kXPieces * 1024;
constexpr int kXRows = 128;                     // query rows per workgroup
constexpr float kXSlack = 8.0f;                 // log2 units a row's max may run ahead of its reference
constexpr float kXSumLimit = 4096.0f;           // a lane's partial row sum above this sends the wave to the max-based step

struct ExtMlaArgs {
  const uint16_t* q;
  const uint16_t* k_ext;
  const uint16_t* v_ext;
  uint16_t* o;
  int64_t q_stride_t, q_stride_h, k_stride_t, v_stride_t, o_stride_t, o_stride_h;
  const uint16_t* k_buf;
  const uint16_t* v_buf;
  int32_t page_shift;  // log2(page_size), or -1 for a pool that is linear in the slot
  int64_t k_page_stride, k_tok_stride, v_page_stride, v_tok_stride;
  const void* qo_indptr;
  int32_t qo64;
  const int32_t* kv_indptr;
  const void* kv_indices;
  int32_t idx64;
  float* lse;
  int64_t lse_stride_t, lse_stride_h;
  int32_t bs, hq, mblocks, xcd_bind;
  float sm_scale, k_scale, v_scale;
  int32_t causal, skip_prefix, skip_extend;
  int32_t share_p, share_e;  // v aliases k[..., :512] in the pool / in the new tokens
};

// 64 slot ids (4 B per lane) global -> LDS
// M0 is set and NOT restored: a restore right behind the DMA waits until the DMA has consumed M0 and keeps costing the
// wave 60-100 cycles per piece however far apart the pieces are (tools/probe/dma_issue.hip: forms 0 / 1).  hipcc uses
// M0 for nothing else in this kernel (checked in the ISA: no m0 operand outside these statements; a kernel with
// dynamically indexed register arrays or ds_gws would).
__device__ __forceinline__ void x_dma4(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void x_dma16(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}

// LDS by 32-bit address: the loop keeps no 64-bit generic pointers alive
typedef __attribute__((address_space(3))) const u32x4* x_lds_u32x4;
typedef __attribute__((address_space(3))) const int32_t* x_lds_i32;
__device__ __forceinline__ u32x4 x_lds_read16(uint32_t addr) { return *reinterpret_cast<x_lds_u32x4>(addr); }
__device__ __forceinline__ int32_t x_lds_read4(uint32_t addr) { return *reinterpret_cast<x_lds_i32>(addr); }

template <typename T>
__device__ __forceinline__ void x_pv_mfma(u32x4 a, u32x4 b, f32x4& c) {
  if constexpr (std::is_same_v<T, BF16>) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// hipcc's hazard recogniser does not see the asm MFMAs: the wait states between an XDL write of the accumulators
// and a VALU read of them (and back) are supplied here
// (asm volatile statements keep their order: the accumulators are re-defined AFTER the nops, so no read of one can
// be scheduled in front of them)
template <int N>
__device__ __forceinline__ void x_settle(f32x4 (&o)[N]) {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+a"(o[i]));
}

// LDS: the stages | two blocks of slot ids | 1 KiB per wave for the rescale's way through LDS
struct XGeom8 {  // (the eight-wave form's geometry; the four-wave first form is in the history and DESIGN 4.2b)
  static constexpr int NW = 8;
  static constexpr int STAGE = 2 * kXImg;  // K image | V image (the second one only for tiles whose v is a tensor of its own)
#ifndef RX_XMLA_NSTAGE8
#define RX_XMLA_NSTAGE8 2  // (four one-image stages, three tiles in flight: no faster -- the landing wait is not what costs)
#endif
  static constexpr int NSTAGE = RX_XMLA_NSTAGE8;  // a power of two; NSTAGE - 1 tiles in flight
  static constexpr int AHEAD = NSTAGE - 1;
  static constexpr int SLOTS_AT = NSTAGE * STAGE;
  static constexpr int SLOTBLK = 256;
  static constexpr int BOUNCE_AT = SLOTS_AT + 2 * SLOTBLK * 4;
  static constexpr int TBL_AT = BOUNCE_AT + NW * 1024;  // row-pointer tables: 2 tiles x 32 rows x {K ptr, V ptr}
  static constexpr int LDS = TBL_AT + 2 * kXTT * 16;    // 162816 B
};
static_assert(XGeom8::LDS <= 160 * 1024, "LDS budget");

// OWNV: the code for tiles whose v is a tensor of its own compiled in (instance picked by the launcher)
template <typename T, bool OWNV>
__global__ __launch_bounds__(512, 1) void extend_mla_kernel(const ExtMlaArgs a) {
  using vec8 = typename T::vec8;
  using G = XGeom8;
  constexpr int KS = kXDk / 32, NB = kXDv / 16;
  constexpr int kXSlotBlock = G::SLOTBLK;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [stages: K image | V image][2 slot blocks][rescale bounce of the 8 waves]
  const uint32_t smem_u = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  int req, mb;
  if (a.xcd_bind) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    req = (j / a.mblocks) * 8 + xcd;
    mb = a.mblocks - 1 - j % a.mblocks;
    if (req >= a.bs) return;
  } else {
    req = blockIdx.x / a.mblocks;
    mb = a.mblocks - 1 - blockIdx.x % a.mblocks;
  }
  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t R = E * a.hq;
  const int32_t row0 = mb * kXRows;
  if (row0 >= R) return;
  const int32_t rbase = row0 + 16 * w;
  const int32_t tok_lo_w = rbase / a.hq;  // the wave's first query token (rows are (token, head) pairs)
  const bool active = rbase < R;

  vec8 qf[KS];
  // (one q head: 2^32 / 1 has no 32-bit magic -- the identity; see rx_extend_d256.hip)
  const uint32_t hq_magic = a.hq == 1 ? 0u : static_cast<uint32_t>(0x100000000ull / static_cast<uint32_t>(a.hq)) + 1u;
  auto row_tok = [&](int m) { return a.hq == 1 ? m : static_cast<int32_t>(__umulhi(static_cast<uint32_t>(m), hq_magic)); };
  {
    const int m = rbase + r;
    const bool ok = m < R;
    const int32_t tk = row_tok(ok ? m : 0);
    const int32_t hd = (ok ? m : 0) - tk * a.hq;
    const uint16_t* qp = a.q + (qo0 + tk) * a.q_stride_t + hd * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      qf[s] = __builtin_bit_cast(vec8, raw);
    }
  }

  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t tok_hi_wg = (min(R, row0 + kXRows) - 1) / a.hq + 1;
  const int32_t tok_hi_w = active ? (min(R, rbase + 16) - 1) / a.hq + 1 : 0;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? tok_hi_wg : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? tok_hi_w : E);
  const int nt1 = (p_len + kXTT - 1) / kXTT;
  const int nt2 = (n_end_wg + kXTT - 1) / kXTT;
  const int nt = nt1 + nt2;

  const char* const idx_b = reinterpret_cast<const char*>(a.kv_indices);
  const int idx_sh = a.idx64 ? 3 : 2;
  auto stage_slots = [&](int blk) {  // waves 0-3: 256 slot ids
    if (w < 4) {
      const int v = blk * kXSlotBlock + tid;
      const int64_t e = kv0 + max(min(v, p_len - 1), 0);
      x_dma4(idx_b + (e << idx_sh), __builtin_amdgcn_readfirstlane(smem_u + G::SLOTS_AT + ((blk & 1) * kXSlotBlock + 64 * w) * 4));
    }
  };
  if (nt1 > 0) stage_slots(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  const char* const kbuf_b = reinterpret_cast<const char*>(a.k_buf);
  const char* const kext_b = reinterpret_cast<const char*>(a.k_ext + qo0 * a.k_stride_t);
  const char* const vbuf_b = reinterpret_cast<const char*>(a.v_buf);
  const char* const vext_b = reinterpret_cast<const char*>(a.v_ext + qo0 * a.v_stride_t);
  auto own_v = [&](int t) { return OWNV && ((t < nt1) ? !a.share_p : !a.share_e); };  // tile t's v rows are a tensor of their own
  const int32_t sh_p = a.page_shift < 0 ? 31 : a.page_shift;
  constexpr int kPieces = 37;
  constexpr int NP = (kPieces + G::NW - 1) / G::NW;  // 5
  static_assert(G::AHEAD == 1, "the table is built two tiles ahead of its use: one tile in flight");
  // One wave per tile writes the 32 rows' K and V pointers two tiles ahead (lane = row: slot lookup, page / token split
  // and the 64-bit multiplies once per ROW); a piece is then its row and column, one ds_read_b64, one 64-bit add, the DMA.
  // The per-piece form above was ~30 instructions x 5 pieces per tile and wave next to 68 MFMAs, in a kernel that pays
  // ~5 cycles for every instruction it issues (rx_extend_d256.hip).
  auto build_table = [&](int u) {
    if (w != (u & 7) || u >= nt) return;
    const int rw = lane & 31;
    const char* kp;
    const char* vp;
    if (u < nt1) {
      const uint32_t sl = smem_u + G::SLOTS_AT + 4 * (((u * kXTT / kXSlotBlock) & 1) * kXSlotBlock + (u * kXTT) % kXSlotBlock + rw);
      const uint32_t slot = static_cast<uint32_t>(x_lds_read4(sl));
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
      const uint32_t n = static_cast<uint32_t>(max(min((u - nt1) * kXTT + rw, n_end_wg - 1), 0));
      kp = kext_b + static_cast<uint64_t>(n) * (2u * static_cast<uint32_t>(a.k_stride_t));
      vp = vext_b + static_cast<uint64_t>(n) * (2u * static_cast<uint32_t>(a.v_stride_t));
    }
    const uint64_t k64 = reinterpret_cast<uint64_t>(kp), v64 = reinterpret_cast<uint64_t>(vp);
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(smem_u + G::TBL_AT + ((u & 1) * kXTT + rw) * 16) =
        u32x4{static_cast<uint32_t>(k64), static_cast<uint32_t>(k64 >> 32), static_cast<uint32_t>(v64), static_cast<uint32_t>(v64 >> 32)};
  };
  auto dma_tile_tbl = [&](int t, int ring) {
    const uint32_t tb = smem_u + G::TBL_AT + (t & 1) * (kXTT * 16);
    const uint32_t img = smem_u + (ring & (G::NSTAGE - 1)) * G::STAGE;
    const bool ownv = own_v(t);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int c0 = 64 * w + ln;
    int row = c0 / kXCpr, col = c0 - row * kXCpr;
    typedef __attribute__((address_space(3))) const u32x2* lds_u32x2;
    u32x2 kp[NP], vp[NP];
    uint32_t c16[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool past = row >= kXTT;
      const uint32_t ra = tb + 16u * static_cast<uint32_t>(past ? kXTT - 1 : row);
      kp[i] = *reinterpret_cast<lds_u32x2>(ra);
      if (ownv) vp[i] = *reinterpret_cast<lds_u32x2>(ra + 8);
      c16[i] = 16u * static_cast<uint32_t>((past || col >= kXCpr - 2) ? kXCpr - 3 : col);  // pad chunks re-read the row's last data chunk
      row += 512 / kXCpr;
      col += 512 % kXCpr;
      if (col >= kXCpr) {
        col -= kXCpr;
        row += 1;
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (w + G::NW * i < kPieces) {
        const uint64_t ks = (static_cast<uint64_t>(kp[i][1]) << 32 | kp[i][0]) + c16[i];
        x_dma16(reinterpret_cast<const void*>(ks), __builtin_amdgcn_readfirstlane(img + (w + G::NW * i) * 1024));
        if (ownv) {  // the same piece of the V image (columns past the 64 v chunks re-read the last one)
          const uint64_t vs = (static_cast<uint64_t>(vp[i][1]) << 32 | vp[i][0]) + min(c16[i], static_cast<uint32_t>(kXDv * 2 - 16));
          x_dma16(reinterpret_cast<const void*>(vs), __builtin_amdgcn_readfirstlane(img + kXImg + (w + G::NW * i) * 1024));
        }
      }
    }
  };
  build_table(0);
  build_table(1);
  __syncthreads();  // the first two tables are readable
  auto dma_tile = [&](int t, int ring) { dma_tile_tbl(t, ring); };
  if (nt > 0) {  // tiles past the end are "loaded" as well (the last one again): the counted waits stay uniform
#pragma unroll
    for (int i = 0; i < G::AHEAD; ++i) dma_tile(min(i, nt - 1), i);
  }

  f32x4 oacc[NB];
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int qd = r >> 2, pp = r & 3;
  const int vrow0 = 4 * g + qd;
  const uint32_t k_lane = r * kXRow + g * 16;
  const uint32_t v_lane = vrow0 * kXRow + 8 * (pp & 1) + (pp >> 1) * 16;
  const uint32_t bounce = smem_u + G::BOUNCE_AT + (w * 64 + lane) * 16;
  const bool late = w >= 4;  // the SIMD partner of an early wave: issues its pieces behind its QK^T

  for (int t = 0; t < nt; ++t) {
    // this wave's pieces of tile t have landed: everything but the AHEAD - 1 youngest tiles' 4 (waves 0-4: 5, so they
    // wait for a piece or two more than they must; a slot-block DMA in between only makes the wait stricter)
    if constexpr (G::AHEAD == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (G::AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else static_assert(G::AHEAD == 1 || G::AHEAD == 3, "two or four stages");
    __syncthreads();
    constexpr int TPB = kXSlotBlock / kXTT;
    if (t % TPB == 0 && (t / TPB + 1) * kXSlotBlock < nt1 * kXTT) stage_slots(t / TPB + 1);
    build_table(t + 2);  // read by tile t + 2's issue, behind the next barrier
    const bool more = true;
    const int t_next = min(t + G::AHEAD, nt - 1), r_next = t + G::AHEAD;
    if (!late) dma_tile(t_next, r_next);
    __builtin_amdgcn_sched_barrier(0);
    const bool prefix = t < nt1;
    const int n0 = (prefix ? t : t - nt1) * kXTT;
    const int32_t lim = prefix ? p_len : n_end_w;
    if (!active || n0 >= lim) {
      if (late) dma_tile(t_next, r_next);
      continue;
    }
    const uint32_t kt = smem_u + (t & (G::NSTAGE - 1)) * G::STAGE;

    // ---- S^T = K Q^T: tokens 16 bb + 4 g + i of the tile on the lane, query row r
    f32x4 sacc[2];
    sacc[0] = sacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int PD = 4;
      const uint32_t krow = kt + k_lane;
      auto kfrag = [&](int j) { return x_lds_read16(krow + (j & 1) * 16 * kXRow + (j >> 1) * 64); };  // j = 2 s + bb
      u32x4 kf[PD];
#pragma unroll
      for (int j = 0; j < PD; ++j) kf[j] = kfrag(j);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const vec8 qb = qf[s];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int j = 2 * s + bb;
          const vec8 ka = __builtin_bit_cast(vec8, kf[j % PD]);
          if (j + PD < 2 * KS) kf[j % PD] = kfrag(j + PD);
          sacc[bb] = T::mfma(ka, qb, sacc[bb]);
        }
        __builtin_amdgcn_sched_barrier(0);  // source order IS the pipeline (hipcc left alone hoists the reads and serialises them)
      }
    }
    if (more && late) dma_tile(t_next, r_next);  // waves 4-7 issue their pieces behind QK^T
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t rp0 = kt + (own_v(t) ? kXImg : 0) + v_lane;
    const uint32_t rp1 = rp0 + 16 * kXRow;
    const float cs = prefix ? a.sm_scale * a.k_scale : a.sm_scale;
    const float c2 = cs * kLog2e;
    u32x4 pf;
    {
      float sv[8];
      auto load_sv = [&]() {  // the tile's 8 scores per lane as the softmax takes them (masked)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[bb][i];
        // (a tile every row of the wave sees in full takes no mask -- wave-uniform branch; rx_extend_d256.hip)
        if (n0 + kXTT > (prefix ? p_len : min(n_end_w, a.causal ? tok_lo_w + 1 : E))) {
          int lnm = lane;
          asm volatile("" : "+v"(lnm));
          const int32_t tk1 = row_tok(rbase + (lnm & 15)) + 1;
          const int32_t vis = (prefix ? p_len : min(n_end_w, a.causal ? tk1 : E)) - n0 - 4 * (lnm >> 4);
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = (16 * bb + i < vis) ? sv[bb * 4 + i] : -INFINITY;
        }
      };
      load_sv();
      // no row maximum on the common path (round 4, rx_extend_d256.hip / rx_extend32_kernel.inc): exponentials against
      // the standing running max, the lane's partial row sum as the check (compared as bits: NaN-proof under
      // -fno-honor-nans); the max-based step only when a lane's sum runs away
      float alpha = 1.0f, psum = 0.f;
      {
        const float m_old = m_run;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          e[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_old));
          psum += e[j];
        }
        if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, psum) > __builtin_bit_cast(uint32_t, kXSumLimit)) != 0) {
          load_sv();
          float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
          mt = quad_row_max(mt) * c2;
          const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;
          const float m_new = (mt_fixed > m_old + kXSlack) ? mt_fixed : m_old;
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
      if (prefix && a.v_scale != 1.0f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] *= a.v_scale;
      }
      pf[0] = pack2<T>(sv[0], sv[1]);
      pf[1] = pack2<T>(sv[2], sv[3]);
      pf[2] = pack2<T>(sv[4], sv[5]);
      pf[3] = pack2<T>(sv[6], sv[7]);
      __builtin_amdgcn_sched_barrier(0);
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
        // the accumulators never pass through compiler-visible code (any C++ access makes hipcc route their live
        // ranges through the VGPR half): they bounce through LDS -- ds_write from / ds_read into the AGPRs in asm,
        // the multiply on a VGPR copy in between (rx_extend_d256.hip)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          asm volatile("ds_write_b128 %0, %1" : : "v"(bounce), "a"(oacc[nb]) : "memory");
          f32x4 tv = *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>(bounce);
          tv *= alpha;
          *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(bounce) = tv;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=a"(oacc[nb]) : "v"(bounce) : "memory");
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NPRE = 4;
    u32x2 vlo[NPRE], vhi[NPRE];
#pragma unroll
    for (int nb = 0; nb < NPRE; ++nb) {
      vlo[nb] = T::ds_read_tr((const void*)(uintptr_t)(rp0 + nb * 32));
      vhi[nb] = T::ds_read_tr((const void*)(uintptr_t)(rp1 + nb * 32));
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const u32x2 lo = vlo[nb % NPRE], hi = vhi[nb % NPRE];
      if (nb + NPRE < NB) {
        vlo[nb % NPRE] = T::ds_read_tr((const void*)(uintptr_t)(rp0 + (nb + NPRE) * 32));
        vhi[nb % NPRE] = T::ds_read_tr((const void*)(uintptr_t)(rp1 + (nb + NPRE) * 32));
      }
      const u32x4 av = u32x4{lo[0], lo[1], hi[0], hi[1]};
      x_pv_mfma<T>(av, pf, oacc[nb]);
    }
  }

  if (!active) return;
  x_settle(oacc);
  float l = l_run;
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const int m = rbase + r;
  if (m >= R) return;
  const float inv = 1.0f / l;
  const int32_t tk = m / a.hq, hd = m - tk * a.hq;
  uint16_t* op = a.o + (qo0 + tk) * a.o_stride_t + hd * a.o_stride_h + 4 * g;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    u32x2 pk;
    pk[0] = pack2<T>(oacc[nb][0] * inv, oacc[nb][1] * inv);
    pk[1] = pack2<T>(oacc[nb][2] * inv, oacc[nb][3] * inv);
    *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
    __builtin_amdgcn_sched_barrier(0);
  }
  if (a.lse && g == 0) a.lse[(qo0 + tk) * a.lse_stride_t + hd * a.lse_stride_h] = m_run * kLn2 + __logf(l);
}

bool extend_mla_supports(const rx_extend_params* p) {
  if (p->head_dim != kXDk || p->v_head_dim != kXDv || p->num_kv_heads != 1 || p->kv.kv_fp8) return false;
  if (p->sliding_window_size > 0 || p->logit_cap > 0.f || p->sinks || p->custom_mask || p->xai_temperature_len > 0 ||
      p->unified_prefix_lens || p->q_pack > 1)
    return false;
  const int64_t all = p->q_stride_t | p->q_stride_h | p->k_stride_t | p->v_stride_t | p->kv.k_page_stride |
                      p->kv.k_tok_stride | p->kv.v_page_stride | p->kv.v_tok_stride;
  if (all % 8 != 0 || (p->o_stride_t | p->o_stride_h) % 4 != 0) return false;
  if ((((uintptr_t)p->q | (uintptr_t)p->k_extend | (uintptr_t)p->v_extend | (uintptr_t)p->kv.k_buf | (uintptr_t)p->kv.v_buf) & 15) != 0 ||
      ((uintptr_t)p->o & 7) != 0)
    return false;
  const bool linear = p->kv.page_size == 1 || (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                                               p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  if (!linear && (p->kv.page_size & (p->kv.page_size - 1)) != 0) return false;
  return static_cast<int64_t>(p->max_extend_len + 1) * p->num_q_heads * p->num_q_heads < (1ll << 31);  // row -> token by multiply-high
}

int launch_extend_mla(const rx_extend_params* p, hipStream_t s) {
  ExtMlaArgs a;
  a.q = (const uint16_t*)p->q;
  a.k_ext = (const uint16_t*)p->k_extend;
  a.v_ext = (const uint16_t*)p->v_extend;
  a.o = (uint16_t*)p->o;
  a.q_stride_t = p->q_stride_t; a.q_stride_h = p->q_stride_h;
  a.k_stride_t = p->k_stride_t; a.v_stride_t = p->v_stride_t;
  a.o_stride_t = p->o_stride_t; a.o_stride_h = p->o_stride_h;
  a.k_buf = (const uint16_t*)p->kv.k_buf;
  a.v_buf = (const uint16_t*)p->kv.v_buf;
  const bool linear = p->kv.page_size == 1 || (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                                               p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  a.page_shift = linear ? -1 : __builtin_ctz(p->kv.page_size);
  a.k_page_stride = p->kv.k_page_stride; a.k_tok_stride = p->kv.k_tok_stride;
  a.v_page_stride = p->kv.v_page_stride; a.v_tok_stride = p->kv.v_tok_stride;
  a.qo_indptr = p->qo_indptr; a.qo64 = p->qo_indptr_is_i64;
  a.kv_indptr = p->kv_indptr; a.kv_indices = p->kv_indices; a.idx64 = p->kv_indices_is_i64;
  a.lse = p->lse; a.lse_stride_t = p->lse_stride_t; a.lse_stride_h = p->lse_stride_h;
  a.bs = p->bs; a.hq = p->num_q_heads;
  a.mblocks = static_cast<int32_t>((static_cast<int64_t>(p->max_extend_len) * p->num_q_heads + kXRows - 1) / kXRows);
  a.xcd_bind = p->bs >= 8;
  a.sm_scale = p->sm_scale; a.k_scale = p->k_scale; a.v_scale = p->v_scale;
  a.causal = p->is_causal; a.skip_prefix = p->skip_prefix; a.skip_extend = p->skip_extend;
  a.share_p = p->kv.v_buf == p->kv.k_buf && p->kv.v_page_stride == p->kv.k_page_stride && p->kv.v_tok_stride == p->kv.k_tok_stride;
  a.share_e = p->v_extend == p->k_extend && p->v_stride_t == p->k_stride_t;
  const unsigned groups = a.xcd_bind ? static_cast<unsigned>((a.bs + 7) / 8) * 8 : static_cast<unsigned>(a.bs);
  const unsigned grid = groups * a.mblocks;
  const bool bf = p->dtype == RX_BF16;
  if (!options().extend_mla_shared_v) a.share_p = a.share_e = 0;  // (A/B switch: the own-v-image path for aliased tensors too)
#define RX_XMLA(TT, OV)                                                                                       \
  do {                                                                                                        \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(extend_mla_kernel<TT, OV>), \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, XGeom8::LDS); \
    (void)attr;                                                                                               \
    hipLaunchKernelGGL((extend_mla_kernel<TT, OV>), dim3(grid), dim3(512), XGeom8::LDS, s, a);                \
  } while (0)
  const bool ownv = !(a.share_p && a.share_e);
  note_dispatch("extend_mla_kernel<%s, %s>|%s", bf ? "rx::BF16" : "rx::F16", tbool(ownv), linear ? "linear" : "paged");
  if (bf) {
    if (ownv) RX_XMLA(BF16, true);
    else RX_XMLA(BF16, false);
  } else {
    if (ownv) RX_XMLA(F16, true);
    else RX_XMLA(F16, false);
  }
#undef RX_XMLA
  return RX_OK;
}

}  // namespace rx
