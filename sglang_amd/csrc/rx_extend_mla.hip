// K7 for the latent (absorbed) MLA shape: q [T, Hq, 576] against ONE kv head whose rows are the 576-wide latent
// (kv_lora_rank 512 + rope 64) and whose values are the first 512 columns of the same row.  This is what an extend
// over a cached prefix runs at for DeepSeek-class models on the reference's backend (forward_absorb_core ->
// attn_mqa -> TritonAttnBackend.forward_extend, triton_backend.py:1290-1437; kernel extend_attention.py:241-661
// with Lq = 576, Lv = 512): every radix-cache hit of an MLA model takes it.  Until round 2 it ran the scalar
// generic kernel here.
//
// Shape of the problem on gfx950: 2176 FLOP per (query row, token) and only 8 + 8 softmax values per lane per
// 32 x 32 block, so VALU does not matter; what does is registers and LDS bandwidth.  With 16x16x32 MFMAs an O^T
// block of 16 rows x 512 columns is 128 accumulator registers, and every K / V^T fragment read from LDS (1 KiB
// per wave instruction) feeds as many MFMAs as the wave has 16-row blocks.  Two blocks per wave (32 rows) make
// LDS time equal to matrix time; that needs 256 accumulator registers, i.e. ONE wave per SIMD with the
// accumulators in the AGPR half (pinned by inline-asm MFMAs: hipcc left alone moves them through v_accvgpr
// copies), the Q fragments (144 registers) in the VGPR half and NO staging registers:
//   * a workgroup is 4 waves x 32 query rows; rows are (token, q head) pairs, row = token * Hq + head -- all
//     heads share the one kv head, so packing costs nothing and a request with 8 new tokens still fills a block;
//   * tiles of 32 tokens go global -> LDS by `global_load_lds_dwordx4` (lane l's 16 bytes land at base + 16 l, the
//     per-lane source address does the row gather), into rows padded to 73 chunks: K fragments by ds_read_b128,
//     V^T fragments by ds_read_b64_tr_b16 FROM THE SAME IMAGE when v aliases k[..., :512] (the pool always does:
//     get_value_buffer is a view of the latent buffer, memory_pool.py MLATokenToKVPool), from a second image
//     otherwise (the new tokens' k is a fresh concat in the reference's model code);
//   * two stages: tile t + 1 flies while tile t is computed; `s_waitcnt vmcnt(0)` + one barrier per tile;
//   * slot ids of 1024 tokens at a time sit in LDS, so the loop has no compiler-visible VMEM;
//   * the running max moves only when a tile's max exceeds it by 2^8 (exact algebra, see rx_extend32.hip): the
//     256-register rescale runs on the first tile and almost never again.
// Causal / non-causal, skip_prefix / skip_extend, LSE, k / v scales.  Windows, caps, sinks, masks: generic kernel.
//
// Two forms.  The one described above (extend_mla_kernel, four waves) is bound by what its DMA costs the issuing wave
// (~270 cycles per 1-KiB piece, 10 per tile, nothing on the SIMD to hide them: DESIGN 4.2b); it serves tensors whose v
// is a tensor of its own.  Aliased tensors -- the common case -- take extend_mla8w_kernel further down: 16 rows per
// wave, TWO waves per SIMD, so that a SIMD partner computes while a wave issues its 5 pieces; it pays with twice the
// LDS reads per FLOP and is LDS-bound instead (650 -> 720-730 TFLOP/s at the bench shape, 712 -> 865 at larger ones).
#include <type_traits>

#include "rx_common.h"

namespace rx {

#ifndef RX_XMLA_DBG
#define RX_XMLA_DBG 0  // dev: 1 = DMA and waits only, 2 = compute only (no DMA in the loop); results are wrong
#endif
#ifndef RX_XMLA_SPREAD
#define RX_XMLA_SPREAD 1  // the DMA pieces of a tile are issued between the MFMAs of the whole iteration (0: all at its top; 2: inside the softmax, the phase without LDS reads)
#endif
#ifndef RX_XMLA_NSTAGE
#define RX_XMLA_NSTAGE 4
#endif

constexpr int kXDk = 576, kXDv = 512, kXTT = 32;
constexpr int kXCpr = kXDk * 2 / 16 + 1;        // 73 chunks per LDS row (72 data + 1 pad)
constexpr int kXRow = kXCpr * 16;               // 1168 B: 9 (odd) chunks past a multiple of 256
constexpr int kXPieces = 37;                    // 1-KiB DMA pieces per image (36.5 carry rows)
constexpr int kXImg = kXPieces * 1024;
// SHARED (v aliases k[..., :512] for the pool AND the new tokens): a ring of four one-image stages, three tiles in
// flight.  Otherwise two stages of K image | V image, one tile in flight.  Both: the Q fragments of the last k-step
// parked in LDS (8 VGPRs the wave does not have: with them in registers hipcc spills a Q fragment and reloads it every
// tile, and a scratch reload waits for every DMA issued before it), slot ids of 256 tokens at a time.
template <bool SHARED>
struct XGeom {
  static constexpr int NSTAGE = SHARED ? RX_XMLA_NSTAGE : 2;
  static constexpr int AHEAD = NSTAGE - 1;
  static constexpr int STAGE = SHARED ? kXImg : 2 * kXImg;
  static constexpr int SLOTBLK = 256;                   // tokens whose slot ids are staged in LDS at a time
  static constexpr int NQL = 1;                         // k-steps of Q kept in LDS
  static constexpr int SLOTS_AT = NSTAGE * STAGE;
  static constexpr int QTAIL_AT = SLOTS_AT + 2 * SLOTBLK * 4;
  static constexpr int LDS = QTAIL_AT + 4 * 2 * NQL * 1024;  // 161792 B of 163840
};
constexpr int kXRows = 128;                     // query rows per workgroup
constexpr float kXSlack = 8.0f;                 // log2 units a row's max may run ahead of its reference

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
// element offset of a KV slot; page_shift < 0: the pool is linear in the slot (page_stride == page_size * tok_stride)
__device__ __forceinline__ int64_t x_slot_off(int32_t slot, int32_t shift, int64_t page_stride, int64_t tok_stride) {
  if (shift < 0) return mul_u32(slot, tok_stride);
  return mul_u32(slot >> shift, page_stride) + mul_u32(slot & ((1 << shift) - 1), tok_stride);
}

#ifndef RX_XMLA_STAMP
#define RX_XMLA_STAMP 0  // 1: diagnostic build, s_memtime phase stamps of wave 0 go to lse[8 * block ...] (tools/mla_extend_bench.py STAMPS=1)
#endif

template <typename T, bool SHARED>
__global__ __launch_bounds__(256, 1) void extend_mla_kernel(const ExtMlaArgs a) {
  using vec8 = typename T::vec8;
  using G = XGeom<SHARED>;
  constexpr int KS = kXDk / 32, NB = kXDv / 16;
  constexpr int KSR = KS - G::NQL;  // k-steps of Q held in registers
  constexpr int kXSlotBlock = G::SLOTBLK;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [stages][2 slot blocks][Q tails of the 4 waves]
  const uint32_t smem_u = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // all workgroups of a request read the same rows: with 8 or more requests a request is bound to one XCD's L2
  int req, mb;
  if (a.xcd_bind) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    req = (j / a.mblocks) * 8 + xcd;
    mb = a.mblocks - 1 - j % a.mblocks;  // heaviest query blocks first under the causal mask
    if (req >= a.bs) return;
  } else {
    req = blockIdx.x / a.mblocks;
    mb = a.mblocks - 1 - blockIdx.x % a.mblocks;
  }
  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t R = E * a.hq;            // query rows of the request: row = token * Hq + head
  const int32_t row0 = mb * kXRows;
  if (row0 >= R) return;                 // workgroup-uniform
  const int32_t rbase = row0 + 32 * w;
  const bool active = rbase < R;         // inactive waves still issue their DMA pieces and hit the barriers

  // ---- Q^T fragments: block c, lane (r, g) holds Q[row rbase + 16 c + r][32 s + 8 g .. +8]
  vec8 qf[2][KSR];
  const uint32_t qtail = smem_u + G::QTAIL_AT + (w * 2 * G::NQL * 64 + lane) * 16;  // this wave's; fragment j at + j KiB
  // row -> token by a multiply-high (exact for rows < 2^32 / Hq, checked by the launcher): the causal mask then
  // needs no per-lane token registers
  const uint32_t hq_magic = static_cast<uint32_t>(0x100000000ull / static_cast<uint32_t>(a.hq)) + 1u;
  auto row_tok = [&](int m) { return static_cast<int32_t>(__umulhi(static_cast<uint32_t>(m), hq_magic)); };
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int m = rbase + 16 * c + r;
    const bool ok = m < R;
    const int32_t tk = row_tok(ok ? m : 0);
    const int32_t hd = (ok ? m : 0) - tk * a.hq;
    const uint16_t* qp = a.q + (qo0 + tk) * a.q_stride_t + hd * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      if (s < KSR) qf[c][s] = __builtin_bit_cast(vec8, raw);
      else *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(qtail + (c * G::NQL + s - KSR) * 1024) = raw;
    }
  }

  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t tok_hi_wg = (min(R, row0 + kXRows) - 1) / a.hq + 1;  // one past the last token of the workgroup's rows
  const int32_t tok_hi_w = active ? (min(R, rbase + 32) - 1) / a.hq + 1 : 0;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? tok_hi_wg : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? tok_hi_w : E);
  const int nt1 = (p_len + kXTT - 1) / kXTT;
  const int nt2 = (n_end_wg + kXTT - 1) / kXTT;
  const int nt = nt1 + nt2;

  // ---- slot ids of 256 prefix tokens at a time -> LDS, by DMA as well: the loop has no compiler-visible VMEM, so no
  // wait of hipcc's drains the ring (the new tokens' rows are consecutive: no ids needed)
  const char* const idx_b = reinterpret_cast<const char*>(a.kv_indices);
  const int idx_sh = a.idx64 ? 3 : 2;  // int64 ids: the low dword
  auto stage_slots = [&](int blk) {
    const int v = blk * kXSlotBlock + tid;
    const int64_t e = kv0 + max(min(v, p_len - 1), 0);
    x_dma4(idx_b + (e << idx_sh), __builtin_amdgcn_readfirstlane(smem_u + G::SLOTS_AT + ((blk & 1) * kXSlotBlock + 64 * w) * 4));
  };
  if (nt1 > 0) stage_slots(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q and the slot ids have landed; nothing of ours is in flight
  __syncthreads();

  // ---- DMA of tile t into stage t & 1: wave w issues pieces w, w + 4, ... of the K image (and of the V image when
  // the tile's v rows are their own tensor).  Piece p, lane l = chunk 64 p + l of the padded image.
  const char* const kbuf_b = reinterpret_cast<const char*>(a.k_buf);
  const char* const vbuf_b = reinterpret_cast<const char*>(a.v_buf);
  const char* const kext_b = reinterpret_cast<const char*>(a.k_ext + qo0 * a.k_stride_t);
  const char* const vext_b = reinterpret_cast<const char*>(a.v_ext + qo0 * a.v_stride_t);
  constexpr int NP = (kXPieces + 3) / 4;  // 10
  // per-kind constants of the address arithmetic, branch-free: offset = (slot >> shift) * pstride + (slot & mask) * tstride
  // (a pool that is linear in the slot and the new tokens' tensors: shift 31, i.e. page 0 and the whole slot as "in-page")
  const int32_t sh_p = a.page_shift < 0 ? 31 : a.page_shift;
  auto dma_tile = [&](int t, int ring) {  // tile t -> stage ring % NSTAGE
    const bool pre = t < nt1;
    const uint32_t sl = smem_u + G::SLOTS_AT + 4 * (((t * kXTT / kXSlotBlock) & 1) * kXSlotBlock + (t * kXTT) % kXSlotBlock);
    const bool own_v = !SHARED && (pre ? !a.share_p : !a.share_e);
    const uint32_t kimg = smem_u + (ring % G::NSTAGE) * G::STAGE;
    const char* const kb = pre ? kbuf_b : kext_b;
    const char* const vb = pre ? vbuf_b : vext_b;
    const int32_t sh = pre ? sh_p : 31;
    const uint32_t mask = sh == 31 ? 0x7fffffffu : (1u << sh) - 1u;
    const uint32_t k_ps = pre ? static_cast<uint32_t>(a.k_page_stride) : 0u, k_ts = static_cast<uint32_t>(pre ? a.k_tok_stride : a.k_stride_t);
    const uint32_t v_ps = pre ? static_cast<uint32_t>(a.v_page_stride) : 0u, v_ts = static_cast<uint32_t>(pre ? a.v_tok_stride : a.v_stride_t);
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque: keeps the per-piece chunk arithmetic inside the loop (hoisted, it costs 30 registers)
    // piece w + 4 i, lane l = chunk 64 w + l + 256 i of the padded image: (row, col) step by (3, 37) mod 73 chunks per
    // row -- three VALU per piece instead of a division
    const int c0 = 64 * w + ln;  // < 256
    int row = (c0 >= kXCpr) + (c0 >= 2 * kXCpr) + (c0 >= 3 * kXCpr);
    int col = c0 - kXCpr * row;
    int32_t slot[NP], col16[NP];  // all LDS reads first: one round trip, not ten
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool past = row >= kXTT;                                    // the last piece's tail: repeats the last chunk
      const int rw = past ? kXTT - 1 : row;
      col16[i] = 16 * ((past || col == kXCpr - 1) ? kXCpr - 2 : col);  // the pad chunk re-reads the row's last data chunk
      slot[i] = pre ? x_lds_read4(sl + 4 * rw) : max(min((t - nt1) * kXTT + rw, n_end_wg - 1), 0);
      col += 256 % kXCpr;
      row += 256 / kXCpr;
      if (col >= kXCpr) {
        col -= kXCpr;
        row += 1;
      }
    }
    const bool paged = pre && sh != 31;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (w + 4 * i < kXPieces) {  // wave-uniform
        const uint32_t in = static_cast<uint32_t>(slot[i]) & mask;
        uint64_t ko = static_cast<uint64_t>(in) * (2 * k_ts) + static_cast<uint32_t>(col16[i]);
        if (paged) ko += static_cast<uint64_t>(static_cast<uint32_t>(slot[i]) >> sh) * (2 * k_ps);
        x_dma16(kb + ko, __builtin_amdgcn_readfirstlane(kimg + (w + 4 * i) * 1024));
        if (own_v) {
          uint64_t vo = static_cast<uint64_t>(in) * (2 * v_ts) + static_cast<uint32_t>(min(col16[i], kXDv * 2 - 16));
          if (paged) vo += static_cast<uint64_t>(static_cast<uint32_t>(slot[i]) >> sh) * (2 * v_ps);
          x_dma16(vb + vo, __builtin_amdgcn_readfirstlane(kimg + kXImg + (w + 4 * i) * 1024));
        }
      }
    }
  };
  if (nt > 0) {
#pragma unroll
    for (int i = 0; i < G::AHEAD; ++i) dma_tile(min(i, nt - 1), i);
  }
  // The LDS-DMA path takes in ~12 B per cycle and CU (a tile's 37 KB: ~3.2 k cycles -- MORE than its 2.2 k cycles of
  // MFMA), and a wave that issues into the full queue stalls: ten pieces at the top of an iteration cost 2.9 k cycles
  // in which the SIMD does nothing else.  The SHARED form therefore issues ONE piece every ~14 MFMAs: state carried
  // between the issue points = the lane's (row, col) in the padded image and the NEXT piece's slot id (read from LDS
  // one issue point ahead), the tile's constants are scalars.
  struct {
    uint32_t sl, img, mask, ts2, ps2;
    int32_t sh, ext0;
    const char* base;
    bool pre, paged;
  } pc;
  int p_rc = 0;  // row * 128 + col
  int32_t p_slot = 0;
  auto piece_slot = [&]() {
    const int rw = min(p_rc >> 7, kXTT - 1);
    p_slot = pc.pre ? x_lds_read4(pc.sl + 4 * rw) : max(min(pc.ext0 + rw, n_end_wg - 1), 0);
  };
  auto piece_begin = [&](int t, int ring) {
    pc.pre = t < nt1;
    pc.sl = smem_u + G::SLOTS_AT + 4 * (((t * kXTT / kXSlotBlock) & 1) * kXSlotBlock + (t * kXTT) % kXSlotBlock);
    pc.img = smem_u + (ring % G::NSTAGE) * G::STAGE;
    pc.base = pc.pre ? kbuf_b : kext_b;
    pc.sh = pc.pre ? sh_p : 31;
    pc.mask = pc.sh == 31 ? 0x7fffffffu : (1u << pc.sh) - 1u;
    pc.ts2 = 2u * static_cast<uint32_t>(pc.pre ? a.k_tok_stride : a.k_stride_t);
    pc.ps2 = pc.pre ? 2u * static_cast<uint32_t>(a.k_page_stride) : 0u;
    pc.paged = pc.pre && pc.sh != 31;
    pc.ext0 = (t - nt1) * kXTT;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int c0 = 64 * w + ln;  // < 256
    const int row0 = (c0 >= kXCpr) + (c0 >= 2 * kXCpr) + (c0 >= 3 * kXCpr);
    p_rc = c0 + (128 - kXCpr) * row0;
    piece_slot();
  };
  auto piece_issue = [&](int k) {  // piece w + 4 k of the tile piece_begin() named; then the next piece's (row, col, slot)
    if (w + 4 * k < kXPieces) {   // wave-uniform
      const int p_col = p_rc & 127;
      const uint32_t col16 = 16u * ((p_rc >= kXTT * 128 || p_col == kXCpr - 1) ? kXCpr - 2 : p_col);
      uint64_t ko = static_cast<uint64_t>(static_cast<uint32_t>(p_slot) & pc.mask) * pc.ts2 + col16;
      if (pc.paged) ko += static_cast<uint64_t>(static_cast<uint32_t>(p_slot) >> pc.sh) * pc.ps2;
      x_dma16(pc.base + ko, __builtin_amdgcn_readfirstlane(pc.img + (w + 4 * k) * 1024));
    }
    p_rc += 128 * (256 / kXCpr) + 256 % kXCpr;
    if ((p_rc & 127) >= kXCpr) p_rc += 128 - kXCpr;
    if (k + 1 < NP) piece_slot();
  };
  constexpr bool SPREAD = SHARED && RX_XMLA_SPREAD && RX_XMLA_DBG == 0;

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
  const int vrow0 = 4 * g + qd;  // V^T read: row inside a 16-token block
  const uint32_t k_lane = r * kXRow + g * 16;
  const uint32_t v_lane = vrow0 * kXRow + 8 * (pp & 1) + (pp >> 1) * 16;

#if RX_XMLA_STAMP
  uint32_t st_acc[6] = {0, 0, 0, 0, 0, 0};
  uint32_t st_prev = (uint32_t)__builtin_amdgcn_s_memtime();
#define X_STAMP(i)                                                 \
  do {                                                             \
    const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();  \
    st_acc[i] += now_ - st_prev;                                   \
    st_prev = now_;                                                \
  } while (0)
#else
#define X_STAMP(i)
#endif
  for (int t = 0; t < nt; ++t) {
    // this wave's pieces of tile t have landed: everything but the two youngest tiles' 2 x 9 (wave 0: 2 x 10, so it
    // waits for two pieces more than it must; a slot-block DMA in between only makes the wait stricter)
    if constexpr (!SHARED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (G::AHEAD == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (G::AHEAD == 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // everybody's have; everybody is done with tile t - 1
    X_STAMP(0);  // landing wait + barrier
    constexpr int TPB = kXSlotBlock / kXTT;  // tiles per slot block
    if (t % TPB == 0 && (t / TPB + 1) * kXSlotBlock < nt1 * kXTT) stage_slots(t / TPB + 1);
#if RX_XMLA_DBG != 2
    if constexpr (SPREAD) piece_begin(min(t + G::AHEAD, nt - 1), t + G::AHEAD);
    else if constexpr (SHARED) dma_tile(min(t + G::AHEAD, nt - 1), t + G::AHEAD);  // past the end: the last tile again (uniform counts)
    else if (t + 1 < nt) dma_tile(t + 1, t + 1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(1);  // DMA issue
#if RX_XMLA_DBG == 1
    continue;
#endif
    const bool prefix = t < nt1;
    const int n0 = (prefix ? t : t - nt1) * kXTT;
    const int32_t lim = prefix ? p_len : n_end_w;
    if (!active || n0 >= lim) {  // nothing visible to this wave (wave-uniform)
      if constexpr (SPREAD) {
#pragma unroll
        for (int k = 0; k < NP; ++k) piece_issue(k);
      }
      continue;
    }
    const uint32_t kt = smem_u + (t % G::NSTAGE) * G::STAGE;
    const uint32_t vt = kt + ((SHARED || (prefix ? a.share_p : a.share_e)) ? 0 : kXImg);

    // ---- S^T = K Q^T: tokens 16 bb + 4 g + i of the tile on the lane, query row r of block c
    f32x4 sacc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) sacc[c][0] = sacc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // a single wave per SIMD: the K fragments are read PD steps ahead of the MFMAs that use them
      constexpr int PD = 4;
      const uint32_t krow = kt + k_lane;
      auto kfrag = [&](int i) { return x_lds_read16(krow + (i / KS) * 16 * kXRow + (i % KS) * 64); };
      u32x4 qt[2][G::NQL];  // parked Q fragments: temporaries of this phase only
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < G::NQL; ++j) qt[c][j] = x_lds_read16(qtail + (c * G::NQL + j) * 1024);
      u32x4 kf[PD];
#pragma unroll
      for (int i = 0; i < PD; ++i) kf[i] = kfrag(i);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        const vec8 ka = __builtin_bit_cast(vec8, kf[i % PD]);
        if (i + PD < 2 * KS) kf[i % PD] = kfrag(i + PD);
        if constexpr (SPREAD) {
          if (RX_XMLA_SPREAD == 1 && i % 9 == 3) piece_issue(i / 9);  // pieces 0 .. 3
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const vec8 qb = (i % KS < KSR) ? qf[c][i % KS < KSR ? i % KS : 0] : __builtin_bit_cast(vec8, qt[c][i % KS < KSR ? 0 : i % KS - KSR]);
          sacc[c][i / KS] = T::mfma(ka, qb, sacc[c][i / KS]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(2);  // QK^T
    const uint32_t rp0 = vt + v_lane;
    const uint32_t rp1 = rp0 + 16 * kXRow;
    const float cs = prefix ? a.sm_scale * a.k_scale : a.sm_scale;
    const float c2 = cs * kLog2e;
    u32x4 pf[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float sv[8];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[c][bb][i];
      {
        // the mask is one compare + select per score on every tile (16 per lane against 136 MFMAs): no second body
        int lnm = lane;
        asm volatile("" : "+v"(lnm));  // (opaque: the row arithmetic stays inside the loop)
        const int32_t tk1 = row_tok(rbase + 16 * c + (lnm & 15)) + 1;  // rows past the request's end: masked by n_end_w only, never stored
        const int32_t vis = (prefix ? p_len : min(n_end_w, a.causal ? tk1 : E)) - n0 - 4 * (lnm >> 4);  // visible: index < vis
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = (16 * bb + i < vis) ? sv[bb * 4 + i] : -INFINITY;
      }
#define XP(k_)                                    \
  do {                                            \
    if constexpr (SPREAD && RX_XMLA_SPREAD == 2) { \
      __builtin_amdgcn_sched_barrier(0);          \
      piece_issue(5 * c + (k_));                  \
      __builtin_amdgcn_sched_barrier(0);          \
    }                                             \
  } while (0)
      XP(0);
      float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
      mt = quad_row_max(mt) * c2;  // c2 > 0: max commutes with the scale
      XP(1);
      const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;  // extend_attention.py:474-475
      const float m_new = (mt_fixed > m_run[c] + kXSlack) ? mt_fixed : m_run[c];
      const float alpha = fast_exp2(m_run[c] - m_new);
      m_run[c] = m_new;
      float psum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sv[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_new));
        psum += sv[j];
        if (j == 2) XP(2);
        if (j == 5) XP(3);
      }
      l_run[c] = l_run[c] * alpha + psum;
      if (prefix && a.v_scale != 1.0f) {  // per-tensor V scale of the cached part (wave-uniform branch)
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] *= a.v_scale;
      }
      pf[c][0] = pack2<T>(sv[0], sv[1]);
      pf[c][1] = pack2<T>(sv[2], sv[3]);
      pf[c][2] = pack2<T>(sv[4], sv[5]);
      pf[c][3] = pack2<T>(sv[6], sv[7]);
      if constexpr (SPREAD && RX_XMLA_SPREAD == 1) piece_issue(4 + c);  // pieces 4, 5
      if constexpr (SPREAD && RX_XMLA_SPREAD == 2) XP(4);
      __builtin_amdgcn_sched_barrier(0);
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // first tile; then only on a 2^8 jump
        x_settle(oacc[c]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          oacc[c][nb] *= alpha;
          asm volatile("" : "+a"(oacc[c][nb]));  // back in its AGPRs before the next one is read
        }
        x_settle(oacc[c]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(3);  // softmax
    // ---- O^T += V^T P^T (the first V^T reads are issued here, not above the softmax: 16 registers the wave does not have)
    constexpr int NPRE = 4;  // V^T fragments read ahead of their MFMA
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
      if constexpr (SPREAD) {
        if (RX_XMLA_SPREAD == 1 && nb % 8 == 3) piece_issue(6 + nb / 8);  // pieces 6 .. 9
      }
      const u32x4 av = u32x4{lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
      for (int c = 0; c < 2; ++c) x_pv_mfma<T>(av, pf[c], oacc[c][nb]);
    }
    X_STAMP(4);  // PV
  }
#if RX_XMLA_STAMP
  if (w == 0 && lane == 0 && a.lse) {
    uint32_t* dbg = reinterpret_cast<uint32_t*>(a.lse) + 8 * blockIdx.x;
    for (int i = 0; i < 5; ++i) dbg[i] = st_acc[i];
    dbg[5] = nt;
  }
#endif

  if (!active) return;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    x_settle(oacc[c]);
    float l = l_run[c];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int m = rbase + 16 * c + r;
    if (m >= R) continue;
    const float inv = 1.0f / l;
    const int32_t tk = m / a.hq, hd = m - tk * a.hq;
    uint16_t* op = a.o + (qo0 + tk) * a.o_stride_t + hd * a.o_stride_h + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      u32x2 pk;
      pk[0] = pack2<T>(oacc[c][nb][0] * inv, oacc[c][nb][1] * inv);
      pk[1] = pack2<T>(oacc[c][nb][2] * inv, oacc[c][nb][3] * inv);
      *reinterpret_cast<u32x2*>(op + 16 * nb) = pk;
      __builtin_amdgcn_sched_barrier(0);  // one accumulator at a time through the VGPR half
    }
    if (a.lse && g == 0 && !RX_XMLA_STAMP) a.lse[(qo0 + tk) * a.lse_stride_t + hd * a.lse_stride_h] = m_run[c] * kLn2 + __logf(l);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The eight-wave form (SHARED tensors only): 16 rows per wave, two waves per SIMD.  A wave's vector-memory path moves
// ~3.6 B per cycle (tools/probe/dma_issue.hip), so with four waves per CU every tile's DMA costs each wave 2.7 k
// cycles that nothing hides; here a wave issues 5 pieces and its SIMD partner computes meanwhile (waves 0-3 issue at
// the top of the iteration, waves 4-7 behind their QK^T).  Price: a K / V^T fragment feeds ONE MFMA, i.e. twice the
// LDS reads per FLOP, and 256 registers per wave: 128 AGPR accumulators, and of the 18 Q fragments only 18 - NQL8
// stay in VGPRs -- the others are parked in LDS and read once per tile (the QK^T loop runs k-step-outer for that).
#ifndef RX_XMLA_NQL8
#define RX_XMLA_NQL8 10
#endif
struct XGeom8 {
  static constexpr int NW = 8;
  static constexpr int NQL = RX_XMLA_NQL8;
  static constexpr int STAGE = kXImg;
  static constexpr int SLOTS_AT = 2 * STAGE;
  static constexpr int SLOTBLK = 256;
  static constexpr int QTAIL_AT = SLOTS_AT + 2 * SLOTBLK * 4;
  static constexpr int LDS = QTAIL_AT + NW * NQL * 1024;
};
static_assert(XGeom8::LDS <= 160 * 1024, "LDS budget of the eight-wave form");

template <typename T>
__global__ __launch_bounds__(512, 1) void extend_mla8w_kernel(const ExtMlaArgs a) {
  using vec8 = typename T::vec8;
  using G = XGeom8;
  constexpr int KS = kXDk / 32, NB = kXDv / 16;
  constexpr int KSR = KS - G::NQL;
  constexpr int kXSlotBlock = G::SLOTBLK;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][2 slot blocks][parked Q of the 8 waves]
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
  const bool active = rbase < R;

  vec8 qf[KSR];
  const uint32_t qtail = smem_u + G::QTAIL_AT + (w * G::NQL * 64 + lane) * 16;
  const uint32_t hq_magic = static_cast<uint32_t>(0x100000000ull / static_cast<uint32_t>(a.hq)) + 1u;
  auto row_tok = [&](int m) { return static_cast<int32_t>(__umulhi(static_cast<uint32_t>(m), hq_magic)); };
  {
    const int m = rbase + r;
    const bool ok = m < R;
    const int32_t tk = row_tok(ok ? m : 0);
    const int32_t hd = (ok ? m : 0) - tk * a.hq;
    const uint16_t* qp = a.q + (qo0 + tk) * a.q_stride_t + hd * a.q_stride_h + 8 * g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 32 * s) : u32x4{0, 0, 0, 0};
      if (s < KSR) qf[s] = __builtin_bit_cast(vec8, raw);
      else *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(qtail + (s - KSR) * 1024) = raw;
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
  const int32_t sh_p = a.page_shift < 0 ? 31 : a.page_shift;
  constexpr int kPieces = 37;
  constexpr int NP = (kPieces + G::NW - 1) / G::NW;  // 5
  // all pieces of tile t (this wave's: w, w + 8, ...) -> stage t & 1
  auto dma_tile = [&](int t) {
    const bool pre = t < nt1;
    const uint32_t sl = smem_u + G::SLOTS_AT + 4 * (((t * kXTT / kXSlotBlock) & 1) * kXSlotBlock + (t * kXTT) % kXSlotBlock);
    const uint32_t img = smem_u + (t & 1) * G::STAGE;
    const char* const base = pre ? kbuf_b : kext_b;
    const int32_t sh = pre ? sh_p : 31;
    const uint32_t mask = sh == 31 ? 0x7fffffffu : (1u << sh) - 1u;
    const uint32_t ts2 = 2u * static_cast<uint32_t>(pre ? a.k_tok_stride : a.k_stride_t);
    const uint32_t ps2 = pre ? 2u * static_cast<uint32_t>(a.k_page_stride) : 0u;
    const bool paged = pre && sh != 31;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int c0 = 64 * w + ln;  // < 512
    int row = c0 / (kXCpr), col = c0 - row * kXCpr;
    int32_t slot[NP], col16[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool past = row >= kXTT;
      const int rw = past ? kXTT - 1 : row;
      col16[i] = 16 * ((past || col == kXCpr - 1) ? kXCpr - 2 : col);
      slot[i] = pre ? x_lds_read4(sl + 4 * rw) : max(min((t - nt1) * kXTT + rw, n_end_wg - 1), 0);
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
        uint64_t ko = static_cast<uint64_t>(static_cast<uint32_t>(slot[i]) & mask) * ts2 + static_cast<uint32_t>(col16[i]);
        if (paged) ko += static_cast<uint64_t>(static_cast<uint32_t>(slot[i]) >> sh) * ps2;
        x_dma16(base + ko, __builtin_amdgcn_readfirstlane(img + (w + G::NW * i) * 1024));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (nt > 0) dma_tile(0);

  f32x4 oacc[NB];
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) oacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int qd = r >> 2, pp = r & 3;
  const int vrow0 = 4 * g + qd;
  const uint32_t k_lane = r * kXRow + g * 16;
  const uint32_t v_lane = vrow0 * kXRow + 8 * (pp & 1) + (pp >> 1) * 16;
  const bool late = w >= 4;  // the SIMD partner of an early wave: issues its pieces behind its QK^T
#undef X_STAMP
#if RX_XMLA_STAMP
  uint32_t st_acc[6] = {0, 0, 0, 0, 0, 0};
  uint32_t st_prev = (uint32_t)__builtin_amdgcn_s_memtime();
#define X_STAMP(i)                                                 \
  do {                                                             \
    const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();  \
    st_acc[i] += now_ - st_prev;                                   \
    st_prev = now_;                                                \
  } while (0)
#else
#define X_STAMP(i)
#endif

  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    X_STAMP(0);
    constexpr int TPB = kXSlotBlock / kXTT;
    if (t % TPB == 0 && (t / TPB + 1) * kXSlotBlock < nt1 * kXTT) stage_slots(t / TPB + 1);
    const bool more = t + 1 < nt;
    if (more && !late) dma_tile(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(1);
    const bool prefix = t < nt1;
    const int n0 = (prefix ? t : t - nt1) * kXTT;
    const int32_t lim = prefix ? p_len : n_end_w;
    if (!active || n0 >= lim) {
      if (more && late) dma_tile(t + 1);
      continue;
    }
    const uint32_t kt = smem_u + (t & 1) * G::STAGE;

    // ---- S^T = K Q^T, k-step-outer: a parked Q fragment is read once per tile
    f32x4 sacc[2];
    sacc[0] = sacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int PD = 4;
      const uint32_t krow = kt + k_lane;
      auto kfrag = [&](int j) { return x_lds_read16(krow + (j & 1) * 16 * kXRow + (j >> 1) * 64); };  // j = 2 s + bb
      u32x4 kf[PD];
#pragma unroll
      for (int j = 0; j < PD; ++j) kf[j] = kfrag(j);
      u32x4 qq[2] = {};
      if (KSR < 2) {
#pragma unroll
        for (int s = KSR; s < 2; ++s) qq[s & 1] = x_lds_read16(qtail + (s - KSR) * 1024);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const vec8 qb = (s < KSR) ? qf[s < KSR ? s : 0] : __builtin_bit_cast(vec8, qq[s & 1]);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int j = 2 * s + bb;
          const vec8 ka = __builtin_bit_cast(vec8, kf[j % PD]);
          if (j + PD < 2 * KS) kf[j % PD] = kfrag(j + PD);
          sacc[bb] = T::mfma(ka, qb, sacc[bb]);
        }
        if (s + 2 < KS && s + 2 >= KSR) qq[s & 1] = x_lds_read16(qtail + (s + 2 - KSR) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    X_STAMP(2);
#ifndef RX_XMLA_LATE_AT
#define RX_XMLA_LATE_AT 1  // where waves 4-7 issue their pieces: 1 behind QK^T, 2 behind the softmax
#endif
    if (RX_XMLA_LATE_AT == 1 && more && late) dma_tile(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(1);
    const uint32_t rp0 = kt + v_lane;
    const uint32_t rp1 = rp0 + 16 * kXRow;
    const float cs = prefix ? a.sm_scale * a.k_scale : a.sm_scale;
    const float c2 = cs * kLog2e;
    u32x4 pf;
    {
      float sv[8];
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = sacc[bb][i];
      {
        int lnm = lane;
        asm volatile("" : "+v"(lnm));
        const int32_t tk1 = row_tok(rbase + (lnm & 15)) + 1;
        const int32_t vis = (prefix ? p_len : min(n_end_w, a.causal ? tk1 : E)) - n0 - 4 * (lnm >> 4);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int i = 0; i < 4; ++i) sv[bb * 4 + i] = (16 * bb + i < vis) ? sv[bb * 4 + i] : -INFINITY;
      }
      float mt = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
      mt = quad_row_max(mt) * c2;
      const float mt_fixed = (mt == -INFINITY) ? -1e20f : mt;
      const float m_new = (mt_fixed > m_run + kXSlack) ? mt_fixed : m_run;
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sv[j] = fast_exp2(__builtin_fmaf(sv[j], c2, -m_new));
        psum += sv[j];
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
        x_settle(oacc);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          oacc[nb] *= alpha;
          asm volatile("" : "+a"(oacc[nb]));
        }
        x_settle(oacc);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(3);
    if (RX_XMLA_LATE_AT == 2 && more && late) dma_tile(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    X_STAMP(1);
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
    X_STAMP(4);
  }
#if RX_XMLA_STAMP
  if ((w == 0 || w == 4) && lane == 0 && a.lse) {
    uint32_t* dbg = reinterpret_cast<uint32_t*>(a.lse) + 16 * blockIdx.x + 2 * w;  // wave 0 at +0, wave 4 at +8
    for (int i = 0; i < 5; ++i) dbg[i] = st_acc[i];
    dbg[5] = nt;
  }
#endif

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
  if (a.lse && g == 0 && !RX_XMLA_STAMP) a.lse[(qo0 + tk) * a.lse_stride_t + hd * a.lse_stride_h] = m_run * kLn2 + __logf(l);
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
  static const bool no_shared = getenv("RX_XMLA_NO_SHARED") != nullptr;  // dev: the two-image form for aliased tensors too
  const bool shared = a.share_p && a.share_e && !no_shared;
  if (no_shared) a.share_p = a.share_e = 0;
#define RX_XMLA(TT, SH)                                                                                              \
  do {                                                                                                               \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(extend_mla_kernel<TT, SH>),     \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, XGeom<SH>::LDS);  \
    (void)attr;                                                                                                      \
    hipLaunchKernelGGL((extend_mla_kernel<TT, SH>), dim3(grid), dim3(256), XGeom<SH>::LDS, s, a);                    \
  } while (0)
  static const bool eight = getenv("RX_XMLA_4W") == nullptr;  // (dev: RX_XMLA_4W keeps the four-wave form for aliased tensors too)
  if (eight && shared) {
    static const hipError_t a8b = hipFuncSetAttribute(reinterpret_cast<const void*>(extend_mla8w_kernel<BF16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, XGeom8::LDS);
    static const hipError_t a8h = hipFuncSetAttribute(reinterpret_cast<const void*>(extend_mla8w_kernel<F16>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, XGeom8::LDS);
    (void)a8b;
    (void)a8h;
    if (bf) hipLaunchKernelGGL(extend_mla8w_kernel<BF16>, dim3(grid), dim3(512), XGeom8::LDS, s, a);
    else hipLaunchKernelGGL(extend_mla8w_kernel<F16>, dim3(grid), dim3(512), XGeom8::LDS, s, a);
    return RX_OK;
  }
  if (bf) {
    if (shared) RX_XMLA(BF16, true);
    else RX_XMLA(BF16, false);
  } else {
    if (shared) RX_XMLA(F16, true);
    else RX_XMLA(F16, false);
  }
#undef RX_XMLA
  return RX_OK;
}

}  // namespace rx
