// K7, head_dim 128, the plain call (causal / non-causal extend over a cached prefix, LSE, sinks; no mask, window,
// cap, packing, fp8 pool): extend attention as FOUR waves of 64 query rows, one wave per SIMD.
//
// Same contract as rx_extend32.hip (extend_attention_fwd, kernels/ops/attention/extend_attention.py:664-812; _fwd_kernel
// :241-661).  Why a third D = 128 kernel: the eight-wave form of rx_extend32.hip is VALU-issue bound (two waves of a
// SIMD share its issue port: 2.26 k issue cycles against 2.05 k of matrix time per tile) and re-reads every K / V^T
// fragment once per 32 query rows.  Here a wave owns its SIMD and the whole 512-register file:
//   * 64 query rows per wave (two 32-row blocks): every K / V^T fragment read from LDS feeds TWO MFMAs;
//   * O^T (128 accumulator registers) and Q (64) live in the AGPR half and are touched by inline-asm MFMAs only; the
//     score tiles, P and everything VALU works on stay in the 256 arch VGPRs;
//   * K / V tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no compiler-visible
//     VMEM in the loop) into an XOR-swizzled image -- chunk ^= ((row & 3) << 2) | ((row >> 2) & 3) inside each 256-byte
//     row, applied on the SOURCE side of the DMA and on the fragment reads (cdna_hip_programming.md T10, image (b)):
//     ds_read_b128 K fragments and ds_read_b64_tr_b16 V^T fragments are both conflict-free; slot ids come by DMA too;
//   * a single in-order wave overlaps VALU with the matrix pipe only inside the shadow of the MFMA in front of it, so
//     the steady-state iteration is GENERATED (tools/gen_extend_pw.py -> rx_extend_pw_body.inc): 64 gaps per tile,
//     each with one MFMA and its share of softmax micro-ops, the stages of one score in different gaps.
// Tiles that need a mask (causal diagonal, ragged ends) take a plain one-query-block-at-a-time body.
#include "rx_common.h"

#ifndef RX_PW_BODY_INC
#define RX_PW_BODY_INC "rx_extend_pw_body.inc"  // dev: alternative schedules from tools/gen_extend_pw.py (PW_GEN_TAG)
#endif
#ifndef RX_PW_RK
#define RX_PW_RK 4  // fragment rings of the generated body (PW_GEN_RK / PW_GEN_RV)
#endif
#ifndef RX_PW_RV
#define RX_PW_RV 4
#endif
#ifndef RX_PW_DMA32
// dev experiment: 1 = the tile DMA with a scalar base and 32-bit lane offsets (global_load_lds ... saddr form); valid only
// while every row offset of the K / V tensors fits 32 bits
#define RX_PW_DMA32 0
#endif
#ifndef RX_PW_LSUM_MFMA
// 1: the softmax denominators by 8 extra MFMAs per tile against a ones fragment (pw_lsum) instead of 64 VALU adds; the
// generated body must match (PW_GEN_LSUM_VALU=0).  Measured 738 vs 778-797 TFLOP/s at config 3: an extra MFMA is a whole
// gap of the in-order wave (~45 cycles), a row-sum add 4 -- kept for the record, off.
#define RX_PW_LSUM_MFMA 0
#endif
#ifndef RX_PW_STAMP
#define RX_PW_STAMP 0  // 1: diagnostic build with s_memtime stamps per group of the generated body (tools/pw_stamps.py); outputs are clobbered
#endif
#ifndef RX_PW_ABL
#define RX_PW_ABL 0  // dev ablations (results are garbage): 2 no LDS fragment reads, 4 no DMA, 8 no per-tile barrier
#endif

namespace rx {

struct ExtPwArgs {
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
  int32_t bs, hq, hkv, group, mblocks;
  float sm_scale, k_scale;
  int32_t causal, skip_prefix, skip_extend;
  const float* sinks;
};

typedef __attribute__((ext_vector_type(16))) float pw_f32x16;

constexpr int kPwTok = 64;                 // tokens per tile
constexpr int kPwImg = kPwTok * 256;       // one image (K or V): 64 rows of 256 B
constexpr int kPwTile = 2 * kPwImg;        // ring slot: K image | V image  (32 KiB)
constexpr int kPwRing = 4;                 // tile t - 1 is still read (its V rows 32..63) while tile t + 2 lands.  Tile t sits in slot
                                           // gray(t & 3) = 0, 1, 3, 2: consecutive tiles differ in ONE address bit (32 KiB / 64 KiB
                                           // alternately), so a fragment address moves to the next tile by one XOR
constexpr int kPwOffAt = kPwRing * kPwTile;  // row-offset table: 2 blocks of 256 byte offsets (64-bit) into the K / V tensors
constexpr int kPwOffBlock = 256;           // rows per block = 4 tiles
constexpr int kPwIdsAt = kPwOffAt + 2 * kPwOffBlock * 8;  // staging of one block's slot ids (LDS-DMA target)
constexpr int kPwLds = kPwIdsAt + kPwOffBlock * 4;
constexpr int kPwPieces = 8;               // LDS-DMA pieces a wave issues per tile (the counted vmcnt waits rely on it)
constexpr float kPwSlack = 8.0f;           // thresholded running max (rx_extend32.hip: kMaxSlack)
constexpr int kPwRows = 256;               // query rows per workgroup

__host__ __device__ constexpr int pw_gray(int i) { return i ^ (i >> 1); }
typedef __attribute__((address_space(3))) const u32x4* pw_lds_u32x4;
typedef __attribute__((address_space(3))) const int32_t* pw_lds_i32;
__device__ __forceinline__ u32x4 pw_lds16(uint32_t addr) {
  if constexpr ((RX_PW_ABL & 2) != 0) return u32x4{addr, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  return *reinterpret_cast<pw_lds_u32x4>(addr);
}
__device__ __forceinline__ int32_t pw_lds4(uint32_t addr) { return *reinterpret_cast<pw_lds_i32>(addr); }
// LDS-DMA: M0 = wave-uniform LDS destination, lane i lands at M0 + 16 i (4 i for the dword form); M0 is not
// restored (hipcc uses it for nothing else in this kernel; rx_extend_mla.hip)
__device__ __forceinline__ void pw_dma4(const void* gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void pw_dma16(const void* gsrc, uint32_t lds_dst) {
  if constexpr ((RX_PW_ABL & 4) != 0) return;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}

// one tile piece: M0 = dst_base + imm (the M0 write's wait state is filled by the address add), address = base + row offset
template <int IMM>
__device__ __forceinline__ void pw_dma16_at(const char* base, uint64_t off, uint32_t dst_base) {
  if constexpr ((RX_PW_ABL & 4) != 0) return;
  uint64_t addr;
  asm volatile("s_add_u32 m0, %2, %3\n\tv_lshl_add_u64 %0, %1, 0, %4\n\tglobal_load_lds_dwordx4 %0, off"
               : "=&v"(addr) : "v"(off), "s"(dst_base), "n"(IMM), "v"(base) : "memory", "scc");
}

template <int IMM>
__device__ __forceinline__ void pw_dma16_s32(const char* sbase, uint32_t voff, uint32_t dst_base) {
  if constexpr ((RX_PW_ABL & 4) != 0) return;
  asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3" : : "v"(voff), "s"(dst_base), "n"(IMM), "s"(sbase) : "memory", "scc");
}

// PV step with the accumulator pinned to the AGPR half; QK^T step with Q read from AGPRs (rx_extend32.hip: pv_mfma / qk_mfma)
template <typename T>
__device__ __forceinline__ void pw_pv(u32x4 a, u32x4 b, pw_f32x16& c) {
  if constexpr (std::is_same_v<T, BF16>) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <typename T, bool FIRST>
__device__ __forceinline__ void pw_qk(u32x4 k, const typename T::vec8& q, pw_f32x16& sc) {
  const u32x4 qr = __builtin_bit_cast(u32x4, q);
  if constexpr (std::is_same_v<T, BF16>) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(sc) : "v"(k), "a"(qr));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sc) : "v"(k), "a"(qr));
  } else {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(sc) : "v"(k), "a"(qr));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(sc) : "v"(k), "a"(qr));
  }
}
// (s_nop 1 in front of every row-sum MFMA: hipcc rematerialises the constant `ones` operand by a v_mov right in front of
// the statement, and it pads no VALU-write -> MFMA-read hazard for inline asm.)
// The softmax denominators: l[q] = sum_k P~[k][q] = (ones x P^T)[any row][q], summed by the matrix pipe over the ROUNDED
// probabilities the PV product multiplies -- 8 MFMAs per tile instead of 64 VALU adds in an issue-bound loop.  The two
// accumulators (query block 0 / 1) are a[224:239] / a[240:255], named literally: every row of the 32 x 32 result is the
// same sum, only element 0 is ever read or rescaled, and the compiler must not see them (it moved a compiler-visible
// accumulator pair through 64 v_accvgpr copies per tile).  Every statement that writes them lists them as clobbers,
// which keeps hipcc's own AGPR values (O^T, Q) out of them and makes the kernel descriptor allocate them.
#define PW_LACC_CLOBBER_0 "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239"
#define PW_LACC_CLOBBER_1 "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <typename T, int QB, bool FIRST>
__device__ __forceinline__ void pw_lsum(u32x4 ones, u32x4 p) {
  if constexpr (QB == 0) {
    if constexpr (std::is_same_v<T, BF16>) {
      if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, 0" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_0);
      else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_0);
    } else {
      if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[224:239], %0, %1, 0" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_0);
      else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[224:239], %0, %1, a[224:239]" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_0);
    }
  } else {
    if constexpr (std::is_same_v<T, BF16>) {
      if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, 0" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_1);
      else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_1);
    } else {
      if constexpr (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[240:255], %0, %1, 0" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_1);
      else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[240:255], %0, %1, a[240:255]" : : "v"(ones), "v"(p) : PW_LACC_CLOBBER_1);
    }
  }
}
// element 0 of the accumulator (the sum so far) times alpha, in place; s_nop: XDL write -> accvgpr read, and the
// accvgpr write -> XDL read is far (the next row-sum MFMA is at least a group away)
template <int QB>
__device__ __forceinline__ void pw_lscale(float alpha) {
  float tmp;
  if constexpr (QB == 0)
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a224\n\ts_nop 1\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a224, %0\n\ts_nop 3"
                 : "=&v"(tmp) : "v"(alpha) : PW_LACC_CLOBBER_0);
  else
    asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a240\n\ts_nop 1\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a240, %0\n\ts_nop 3"
                 : "=&v"(tmp) : "v"(alpha) : PW_LACC_CLOBBER_1);
}
template <int QB>
__device__ __forceinline__ float pw_lread() {
  float v;
  if constexpr (QB == 0) asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a224\n\ts_nop 1" : "=v"(v));
  else asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a240\n\ts_nop 1" : "=v"(v));
  return v;
}
__device__ __forceinline__ void pw_settle(pw_f32x16 (&o)[4]) {  // XDL write -> VALU read (and back): by hand around asm
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(o[0]), "+a"(o[1]), "+a"(o[2]), "+a"(o[3]));
}

// this file is built with -fno-honor-nans (sglang_amd/build.py): fmaxf on an MFMA result then needs no canonicalising v_max
__device__ __forceinline__ float pw_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float pw_swap_max(float x) {  // max over lanes l and l ^ 32
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}

template <typename T>
__global__ __launch_bounds__(256, 1) void extend_pw_kernel(const ExtPwArgs a) {
  using vec8 = typename T::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t smem_u = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, h = lane >> 5;

  // block index -> (kv head fastest: a kv head's prefix rows stay in one XCD's L2; heaviest query blocks first)
  int bid = blockIdx.x;
  const int kvh = bid % a.hkv;
  bid /= a.hkv;
  const int mb = a.mblocks - 1 - bid % a.mblocks;
  bid /= a.mblocks;
  const int head = kvh * a.group + bid % a.group;
  const int req = bid / a.group;

  const int64_t qo0 = load_idx(a.qo_indptr, req, a.qo64);
  const int32_t E = static_cast<int32_t>(load_idx(a.qo_indptr, req + 1, a.qo64) - qo0);
  const int32_t kv0 = a.kv_indptr[req];
  const int32_t P = a.kv_indptr[req + 1] - kv0;
  const int32_t qb0 = mb * kPwRows;
  if (qb0 >= E) return;  // workgroup-uniform
  const int32_t qbase = qb0 + 64 * w;
  const bool active = qbase < E;

  // ---- Q^T fragments: block qb, lane (ql, h) holds Q[qbase + 32 qb + ql][16 ks + 8 h .. +8]; kept as AGPR-class values
  vec8 qf[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int m = qbase + 32 * qb + ql;
    const bool ok = m < E;
    const uint16_t* qp = a.q + (qo0 + (ok ? m : 0)) * a.q_stride_t + head * a.q_stride_h + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4 raw = ok ? *reinterpret_cast<const u32x4*>(qp + 16 * ks) : u32x4{0, 0, 0, 0};
      qf[qb][ks] = __builtin_bit_cast(vec8, raw);
    }
  }

  const int32_t p_len = a.skip_prefix ? 0 : P;
  const int32_t n_end_wg = a.skip_extend ? 0 : (a.causal ? min(E, qb0 + kPwRows) : E);
  const int32_t n_end_w = a.skip_extend ? 0 : (a.causal ? min(E, qbase + 64) : E);
  const int nt1 = (p_len + kPwTok - 1) / kPwTok;
  const int nt2 = (n_end_wg + kPwTok - 1) / kPwTok;
  const int nt = nt1 + nt2;

  // ---- the row-offset table.  The tile DMA needs, per image row, the byte offset of that token's K / V row in its
  // tensor: for a cached token slot -> (slot >> sh) * page_stride + (slot & mask) * tok_stride, for a new token
  // index * stride -- 64-bit multiplies that would cost the loop ~100 VALU per tile if every lane did them for its
  // rows.  They are done ONCE per row instead: the rows of the concatenated tile list (prefix tiles, then new-token
  // tiles) are cut into blocks of 256; thread tid computes row blk * 256 + tid and stores the offset in LDS (two blocks
  // resident).  K and V tensors have equal strides (extend_pw_supports), so one offset serves both.  No compiler-visible
  // VMEM in the loop (hipcc would wait vmcnt(0) for it and drain the tile DMA in flight): the slot ids of block B come
  // into an LDS staging row by DMA, issued right after barrier 4B - 4 (before that iteration's tile pieces, so the
  // counted wait at the top of the next tile covers it), are turned into offsets after barrier 4B - 3, and the table
  // block is complete at barrier 4B - 2, where the DMA of tile 4B (the block's first) is issued.
  const char* const idx_b = reinterpret_cast<const char*>(a.kv_indices);
  const int idx_sh = a.idx64 ? 3 : 2;
  const int32_t sh_p = a.page_shift < 0 ? 31 : a.page_shift;
  const uint32_t mask_p = sh_p == 31 ? 0x7fffffffu : (1u << sh_p) - 1u;
  const uint64_t ts_p = 2ull * static_cast<uint64_t>(a.k_tok_stride), ps_p = sh_p == 31 ? 0ull : 2ull * static_cast<uint64_t>(a.k_page_stride);
  const uint64_t ts_e = 2ull * static_cast<uint64_t>(a.k_stride_t);
  const uint32_t lds_s = __builtin_amdgcn_readfirstlane(smem_u);  // provably wave-uniform: M0 values stay scalar
  auto off_row_is_prefix = [&](int row) { return row < nt1 * kPwTok; };
  auto off_dma_ids = [&](int blk) {              // phase A: this wave's 64 slot ids of block blk -> LDS staging
    if (nt1 * kPwTok <= blk * kPwOffBlock) return;   // a block of new tokens only (wave-uniform)
    const int row = blk * kPwOffBlock + tid;
    const int64_t e = kv0 + max(min(row, p_len - 1), 0);
    pw_dma4(idx_b + (e << idx_sh), lds_s + kPwIdsAt + 64 * w * 4);
  };
  auto off_store = [&](int blk) {                // phase B: offset of row blk * 256 + tid -> table
    const int row = blk * kPwOffBlock + tid;
    uint64_t off;
    if (off_row_is_prefix(row)) {
      const uint32_t sl = static_cast<uint32_t>(pw_lds4(smem_u + kPwIdsAt + 4 * tid));
      off = static_cast<uint64_t>(sl & mask_p) * ts_p + static_cast<uint64_t>(sl >> sh_p) * ps_p;
    } else {
      const int32_t tok = max(min(row - nt1 * kPwTok, n_end_wg - 1), 0);
      off = static_cast<uint64_t>(static_cast<uint32_t>(tok)) * ts_e;
    }
    *reinterpret_cast<uint64_t*>(smem + kPwOffAt + ((blk & 1) * kPwOffBlock + tid) * 8) = off;
  };
  const int n_off_blocks = (nt * kPwTok + kPwOffBlock - 1) / kPwOffBlock;
  for (int blk = 0; blk < min(n_off_blocks, 2); ++blk) {  // the first two blocks, before anything else is in flight
    off_dma_ids(blk);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    off_store(blk);
    __syncthreads();   // the staging row is free again
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): Q has landed
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {  // re-define as AGPR-class values: no copy in front of every asm use
      u32x4 tq = __builtin_bit_cast(u32x4, qf[qb][ks]);
      asm volatile("" : "+a"(tq));
      qf[qb][ks] = __builtin_bit_cast(vec8, tq);
    }
  __syncthreads();

  // ---- the tile DMA: 16 K + 16 V pieces of 1 KiB (4 rows) per tile; wave w takes pieces w, w + 4, w + 8, w + 12 of
  // both.  Lane: image row 4 w + 16 j + (lane >> 4) of piece j, image chunk lane & 15 = source chunk ^ swizzle(row).
  const char* const kbuf_b = reinterpret_cast<const char*>(a.k_buf + kvh * a.k_head_stride);
  const char* const vbuf_b = reinterpret_cast<const char*>(a.v_buf + kvh * a.v_head_stride);
  const char* const kext_b = reinterpret_cast<const char*>(a.k_ext + qo0 * a.k_stride_t + kvh * a.k_stride_h);
  const char* const vext_b = reinterpret_cast<const char*>(a.v_ext + qo0 * a.v_stride_t + kvh * a.v_stride_h);
  const uint32_t cx = 16u * static_cast<uint32_t>((lane & 15) ^ ((((lane >> 4) & 3) << 2) | (w & 3)));
  const uint32_t r0_8 = smem_u + kPwOffAt + 8u * static_cast<uint32_t>(4 * w + (lane >> 4));  // table address of row j = 0
  // per-type lane bases (prefix / new tokens), switched when the DMA crosses from prefix tiles to new-token tiles
  const char* d_kx = nullptr;
  const char* d_vx = nullptr;
  const char* d_ks = nullptr;  // (RX_PW_DMA32: the same bases without the lane part, wave-uniform)
  const char* d_vs = nullptr;
  uint32_t d_dst = 0, d_tab = 0;   // scalars: ring slot of the tile being fetched, its first row's table offset
  auto dma_tile_begin = [&](int tn_raw) {
    const int tn = min(tn_raw, nt - 1);  // past the end: re-fetch the last tile into the free slot (never read)
    const bool pre = tn < nt1;
    d_kx = (pre ? kbuf_b : kext_b) + cx;
    d_vx = (pre ? vbuf_b : vext_b) + cx;
    if constexpr (RX_PW_DMA32 != 0) {
      auto uni = [](const char* p) {
        const uint64_t v = reinterpret_cast<uint64_t>(p);
        const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
        const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
        return reinterpret_cast<const char*>((static_cast<uint64_t>(hi) << 32) | lo);
      };
      d_ks = uni(pre ? kbuf_b : kext_b);
      d_vs = uni(pre ? vbuf_b : vext_b);
    }
    d_dst = lds_s + pw_gray(tn_raw & 3) * kPwTile + 1024 * w;
    d_tab = (static_cast<uint32_t>(tn) * kPwTok * 8u) & (2 * kPwOffBlock * 8 - 1);
  };
  uint64_t d_off[4];
  auto dma_row = [&](int j) {  // one ds_read_b64: rows of piece j are 16 j further down the table
    d_off[j] = *reinterpret_cast<__attribute__((address_space(3))) const uint64_t*>(r0_8 + d_tab + 128 * j);
  };
  auto dma_piece = [&](auto jc, auto vc) {  // compile-time piece: its LDS offset is an immediate of the M0 add
    constexpr int j = decltype(jc)::value, vside = decltype(vc)::value;
    if constexpr (RX_PW_DMA32 != 0) {
      const char* sb = vside ? d_vs : d_ks;
      pw_dma16_s32<vside * kPwImg + 4096 * j>(sb, static_cast<uint32_t>(d_off[j]) + cx, d_dst);
    } else {
      pw_dma16_at<vside * kPwImg + 4096 * j>(vside ? d_vx : d_kx, d_off[j], d_dst);
    }
  };
  auto dma_tile_all = [&](int tn) {
    dma_tile_begin(tn);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_row(j);
    dma_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    dma_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    dma_piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    dma_piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    dma_piece(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    dma_piece(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
    dma_piece(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
    dma_piece(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
  };
  // the offset-table pipeline, called right after barrier t and BEFORE that iteration's tile pieces (see above)
  auto off_table_step = [&](int t) {
    if ((t + 3) % 4 == 0) {
      const int blk = (t + 3) / 4;
      if (blk >= 2 && blk < n_off_blocks) off_store(blk);
    }
    if ((t + 4) % 4 == 0) {
      const int blk = (t + 4) / 4;
      if (blk >= 2 && blk < n_off_blocks) off_dma_ids(blk);
    }
  };
  if (nt > 0) {  // two tiles ahead from the start
    dma_tile_all(0);
    dma_tile_all(1);
  }

  // O^T (compiler-allocated AGPRs) and the denominators (pw_lsum: literal AGPRs)
  pw_f32x16 oacc[2][4];
  const uint32_t one2 = std::is_same_v<T, BF16> ? 0x3F803F80u : 0x3C003C00u;
  const u32x4 ones_frag = {one2, one2, one2, one2};
  float m_run[2], l_run[2] = {0.f, 0.f};
  if constexpr (RX_PW_LSUM_MFMA != 0) {  // the denominators start at zero: ones x 0
    const u32x4 z = {0, 0, 0, 0};
    pw_lsum<T, 0, true>(ones_frag, z);
    pw_lsum<T, 1, true>(ones_frag, z);
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = -INFINITY;

#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[qb][db][i] = 0.f;
  }

  // ---- fragment addresses in the swizzled image (ring slot 0 first; PW_TOGGLE moves them to the next slot at the end of a tile).  K fragment (b, ks):
  // row 32 b + ql, chunk (2 ks + h) ^ sw(ql).  V^T fragment (step, db), read sec = 0 / 1: row 16 step + 4 h + qd
  // + 8 sec, chunk (4 db + 2 dg + (pp >> 1)) ^ sw(row), byte 8 (pp & 1) -- sw(row) = (qd << 2) | ((h + 2 sec) & 3).
  uint32_t ka[8], va[8], vp[8];  // K / V^T fragment addresses of the current tile, V^T of the previous one
  {
    const int swq = ((ql & 3) << 2) | ((ql >> 2) & 3);
    const int tq = lane & 15, qd = tq >> 2, pp = tq & 3, dg = (lane >> 4) & 1;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ka[ks] = smem_u + ql * 256 + (((2 * ks + h) ^ swq) << 4);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int sec = 0; sec < 2; ++sec) {
        const int swr = (qd << 2) | ((h + 2 * sec) & 3);
        va[2 * db + sec] = smem_u + kPwImg + (4 * h + qd + 8 * sec) * 256 + (((4 * db + 2 * dg + (pp >> 1)) ^ swr) << 4) + 8 * (pp & 1);
      }
  }
  auto ld_v = [&](int prev, int step, int db) {  // prev: 1 = from the previous tile's slot
    if constexpr ((RX_PW_ABL & 2) != 0) return u32x4{(uint32_t)step, (uint32_t)db, 0x3c003c00u, 0x3c003c00u};
    const u32x2 lo2 = T::ds_read_tr((const void*)(uintptr_t)((prev ? vp[2 * db] : va[2 * db]) + step * 4096));
    const u32x2 hi2 = T::ds_read_tr((const void*)(uintptr_t)((prev ? vp[2 * db + 1] : va[2 * db + 1]) + step * 4096));
    return u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
  };

  // one barrier per tile: tile t has landed (every wave waited for its own pieces), everybody is done with tile t - 1
#if RX_PW_STAMP
  uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t st_prev = (uint32_t)st_t0;
  uint32_t st_tiles = 0;
#define PW_STAMP(i)                                                  \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();    \
    st_acc[i] += now_ - st_prev;                                     \
    st_prev = now_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#else
#define PW_STAMP(i)
#endif
  // vmcnt(8): this wave's pieces of tile t have landed -- they are older than the 8 pieces of tile t + 1, the only
  // vector-memory operations that may still be in flight (the id DMA of off_table_step is issued before its iteration's pieces)
  auto tile_top = [&](int t) {
    static_assert(kPwPieces == 8, "the literal in the wait below");
    PW_STAMP(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    PW_STAMP(6);
    if constexpr ((RX_PW_ABL & 8) == 0) __syncthreads();
    PW_STAMP(7);
    off_table_step(t);
  };
  auto is_fast = [&](int t) {  // both 32-key blocks fully visible to all 64 rows of this wave
    if (t < nt1) return (t + 1) * kPwTok <= p_len;
    const int n_hi = (t - nt1 + 1) * kPwTok;
    return n_hi <= E && n_hi <= n_end_w && (!a.causal || n_hi - 1 <= qbase);
  };

#define PW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PW_LDK(b, ks) pw_lds16(ka[ks] + (b) * 8192)
#define PW_LDV(step, db) ld_v(0, step, db)
#define PW_LDVP(step, db) ld_v(1, step, db)
#define PW_TOGGLE(x)                  \
  do {                                \
    (x) ^= d_adv;                     \
    asm volatile("" : "+v"(x));       \
  } while (0)
#define PW_TOGGLE_V(i)                \
  do {                                \
    vp[i] = va[i];                    \
    va[i] ^= d_adv;                   \
    asm volatile("" : "+v"(va[i]), "+v"(vp[i])); \
  } while (0)
// jump test of one stream: does any row of this wave need a new reference maximum?  (raw-score threshold thr = (m + slack) / c2)
#define PW_JUMP(blk, qb)                                                                      \
  do {                                                                                        \
    if (__builtin_amdgcn_ballot_w64(ma##blk[qb] > thr[qb]) != 0) {                            \
      const float mt_ = fmaxf(pw_swap_max(ma##blk[qb]) * c2r, -1e20f);                        \
      const float mn_ = (mt_ > mref[qb] + kPwSlack) ? mt_ : mref[qb];                         \
      alpha##blk[qb] = fast_exp2(mref[qb] - mn_);                                             \
      mref[qb] = mn_;                                                                         \
      thr[qb] = (mn_ + kPwSlack) * c2inv;                                                     \
      jump##blk = true;                                                                       \
    }                                                                                         \
  } while (0)
#define PW_QK(first, sdst, kfr, qb, ks) pw_qk<T, first>(kfr, qf[qb][ks], sdst)
#define PW_PV(vfr, pfr, qb, db) pw_pv<T>(vfr, pfr, oacc[qb][db])
#define PW_LSUM(pfr, qb) pw_lsum<T, qb, false>(ones_frag, pfr)
#define PW_ROW(j) dma_row(j)
#define PW_DMA(j, vs) dma_piece(std::integral_constant<int, j>{}, std::integral_constant<int, vs>{})
#define PW_RESCALE(al, flag)                                            \
  do {                                                                  \
    if (flag) {                                                         \
      _Pragma("unroll") for (int qb_ = 0; qb_ < 2; ++qb_) {             \
        pw_settle(oacc[qb_]);                                           \
        _Pragma("unroll") for (int db_ = 0; db_ < 4; ++db_) oacc[qb_][db_] *= al[qb_]; \
        pw_settle(oacc[qb_]);                                           \
        if constexpr (RX_PW_LSUM_MFMA != 0) {                           \
          if (qb_ == 0) pw_lscale<0>(al[0]);                            \
          else pw_lscale<1>(al[1]);                                     \
        }                                                               \
        al[qb_] = 1.0f;                                                 \
      }                                                                 \
      flag = false;                                                     \
    }                                                                   \
  } while (0)
#define max3f pw_max3
#define max2f fmaxf
#define half_swap_max pw_swap_max

  int t = 0;
  uint32_t d_adv = kPwTile;  // the address bit that differs between the slots of tiles t and t + 1: 32 KiB for even t, 64 KiB for odd t
  auto ring_step = [&]() {   // after ++t
    d_adv = (t & 1) ? 2 * kPwTile : kPwTile;
  };
  while (t < nt) {
    if (active && is_fast(t)) {
      // ===== a run of fully visible tiles [t, fe): the generated pipeline
      int fe;
      if (t < nt1) fe = min(nt1, p_len / kPwTok);
      else fe = min(nt, nt1 + min(min(E, n_end_w), a.causal ? qbase + 1 : E) / kPwTok);
      fe = max(fe, t + 1);
      const float c2r = (t < nt1 ? a.sm_scale * a.k_scale : a.sm_scale) * kLog2e;
      const float c2inv = 1.0f / c2r;
      pw_f32x16 s0[2], s1[2];
      u32x4 pk0[2][2], pk1[2][2];
      u32x4 kf[RX_PW_RK], vfa[RX_PW_RV];
      float ma0[2], mb0[2], ma1[2], mb1[2], mref[2], thr[2], alpha0[2], alpha1[2], psa0[2], psb0[2], psa1[2], psb1[2];
      bool jump0 = false, jump1 = false;   // wave-uniform: a stream of block 0 / 1 moved a reference maximum
#define PW_TV(b, q) float tv##b##_##q##_0, tv##b##_##q##_1, tv##b##_##q##_2, tv##b##_##q##_3, tv##b##_##q##_4, tv##b##_##q##_5, \
    tv##b##_##q##_6, tv##b##_##q##_7, tv##b##_##q##_8, tv##b##_##q##_9, tv##b##_##q##_10, tv##b##_##q##_11, tv##b##_##q##_12,   \
    tv##b##_##q##_13, tv##b##_##q##_14, tv##b##_##q##_15
      PW_TV(0, 0);
      PW_TV(0, 1);
      PW_TV(1, 0);
      PW_TV(1, 1);
#undef PW_TV
      tile_top(t);
      // dummy predecessor: block 1 of "tile t - 1" with every score -inf (P = 0, no jump, sums unchanged), its stream
      // run up to slot 7; the PV of the dummy multiplies rows of THIS tile (finite values) by zero: vp = va
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s1[qb][i] = -INFINITY;
        mref[qb] = m_run[qb];
        thr[qb] = (m_run[qb] + kPwSlack) * c2inv;   // -inf for a row that has seen nothing: its first block jumps
        alpha0[qb] = alpha1[qb] = 1.0f;
        pk1[qb][0] = pk1[qb][1] = u32x4{0, 0, 0, 0};
        ma1[qb] = mb1[qb] = -INFINITY;
        psa0[qb] = psb0[qb] = psa1[qb] = psb1[qb] = 0.f;
        // the generated G1 starts at slot 8: it adds elements that slot 7 exponentiated (0) and exponentiates the
        // elements whose F ran in slot 7 (-inf); see X_SLOTS / F_SLOTS of the generator
#pragma unroll
        for (int i = 0; i < 6; ++i) s1[qb][i] = 0.f;
      }
      tv1_0_6 = tv1_1_6 = -INFINITY;
      if (!(mref[0] > -INFINITY)) mref[0] = -1e20f;   // F of the dummy computes fma(-inf, c2, -m): keep m finite
      if (!(mref[1] > -INFINITY)) mref[1] = -1e20f;
#pragma unroll
      for (int i = 0; i < 8; ++i) vp[i] = va[i];
      dma_tile_begin(t + 2);
      for (;;) {
#include RX_PW_BODY_INC
#if RX_PW_STAMP
        ++st_tiles;
#endif
        ++t;
        ring_step();
        if (t >= fe) break;
        tile_top(t);
        // the tile fetched next: everything but its ring slot and table row carries over, except where the prefix ends
        if (t + 2 == nt1 || t + 2 >= nt) dma_tile_begin(t + 2);
        else {
          d_dst ^= ((t + 1) & 1) ? 2 * kPwTile : kPwTile;   // slot of tile t + 2 from the slot of tile t + 1 (Gray order)
          d_tab = (d_tab + kPwTok * 8) & (2 * kPwOffBlock * 8 - 1);
        }
      }
#include "rx_extend_pw_drain.inc"
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) m_run[qb] = mref[qb];
      continue;
    }
    // ===== boundary tiles (causal diagonal, ragged ends) and tiles this wave does not see: one barrier, the next
    // tile's DMA at once, one query block at a time
    tile_top(t);
    dma_tile_all(t + 2);
    const bool prefix = t < nt1;
    const int tile_n0 = (prefix ? t : t - nt1) * kPwTok;
    const int32_t lim = prefix ? p_len : n_end_w;
    if (active && tile_n0 < lim) {
      const int nblk = (tile_n0 + 32 < lim) ? 2 : 1;
      const float c2 = (prefix ? a.sm_scale * a.k_scale : a.sm_scale) * kLog2e;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        pw_f32x16 sacc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b < nblk) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
              const u32x4 kfr = pw_lds16(ka[ks] + b * 8192);
              if (ks == 0) pw_qk<T, true>(kfr, qf[qb][ks], sacc[b]);
              else pw_qk<T, false>(kfr, qf[qb][ks], sacc[b]);
            }
          }
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sacc[0]), "+v"(sacc[1]));  // XDL write -> VALU read
        const int m_row = qbase + 32 * qb + ql;
        // visible keys of this lane's row: list index n < vis (prefix: all valid ones; new tokens: up to the row itself)
        const int32_t vis = (prefix ? p_len : min(n_end_w, a.causal ? m_row + 1 : E)) - tile_n0 - 4 * h;
        float mt = -INFINITY;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b < nblk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int c = 32 * b + (i & 3) + 8 * (i >> 2);
              sacc[b][i] = (c < vis) ? sacc[b][i] : -INFINITY;
              mt = fmaxf(mt, sacc[b][i]);
            }
          }
        }
        mt = fmaxf(pw_swap_max(mt) * c2, -1e20f);  // extend_attention.py:474-475
        const float m_new = (mt > m_run[qb] + kPwSlack) ? mt : m_run[qb];  // thresholded running max (see the generator)
        const float alpha = fast_exp2(m_run[qb] - m_new);
        m_run[qb] = m_new;
        u32x4 pk[2][2];
        float psum = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (b < nblk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float v0 = fast_exp2(__builtin_fmaf(sacc[b][2 * i], c2, -m_new));
              const float v1 = fast_exp2(__builtin_fmaf(sacc[b][2 * i + 1], c2, -m_new));
              if constexpr (RX_PW_LSUM_MFMA == 0) psum += v0 + v1;
              pk[b][i >> 2][i & 3] = pack2<T>(v0, v1);
            }
          }
        }
        l_run[qb] = l_run[qb] * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
          pw_settle(oacc[qb]);
#pragma unroll
          for (int db = 0; db < 4; ++db) oacc[qb][db] *= alpha;
          pw_settle(oacc[qb]);
          if constexpr (RX_PW_LSUM_MFMA != 0) {
            if (qb == 0) pw_lscale<0>(alpha);
            else pw_lscale<1>(alpha);
          }
        }
#pragma unroll
        for (int step = 0; step < 4; ++step) {
          if (step < 2 * nblk) {
#pragma unroll
            for (int db = 0; db < 4; ++db) pw_pv<T>(ld_v(0, step, db), pk[step >> 1][step & 1], oacc[qb][db]);
            if constexpr (RX_PW_LSUM_MFMA != 0) {
              if (qb == 0) pw_lsum<T, 0, false>(ones_frag, pk[step >> 1][step & 1]);
              else pw_lsum<T, 1, false>(ones_frag, pk[step >> 1][step & 1]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // the fragment addresses follow the ring
      PW_TOGGLE(ka[i]);
      PW_TOGGLE_V(i);
    }
    ++t;
    ring_step();
  }
#undef max3f
#undef max2f
#undef half_swap_max

  // ---- epilogue: each wave transposes its 64 x 128 block through LDS (the tiles are dead after one more barrier) and
  // writes whole 256-byte rows (rx_extend32.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) tile's DMA must not land on the transposed rows
  __syncthreads();
  if (!active) return;
  constexpr int kORow = 272;
  char* obuf = smem + w * (32 * kORow);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l;
    if constexpr (RX_PW_LSUM_MFMA != 0) {
      l = qb == 0 ? pw_lread<0>() : pw_lread<1>();  // every row of the ones product holds the row sum of this lane's query
    } else {
      float a2 = l_run[qb], b2 = l_run[qb];  // the lane's 16 keys per block + the other half's
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(a2), "+v"(b2));
      l = a2;
    }
    const int m_row = qbase + 32 * qb + ql;
    float den = l;
    if (a.sinks) den += fast_exp2(a.sinks[head] * kLog2e - m_run[qb]);
    const float inv = 1.0f / den;
    pw_settle(oacc[qb]);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {  // registers 4 gq .. 4 gq + 3 = d 32 db + 8 gq + 4 h + 0..3
        u32x2 pk2;
        pk2[0] = pack2<T>(oacc[qb][db][4 * gq] * inv, oacc[qb][db][4 * gq + 1] * inv);
        pk2[1] = pack2<T>(oacc[qb][db][4 * gq + 2] * inv, oacc[qb][db][4 * gq + 3] * inv);
        *reinterpret_cast<u32x2*>(obuf + ql * kORow + (32 * db + 8 * gq + 4 * h) * 2) = pk2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int32_t row0 = qbase + 32 * qb;
#pragma unroll
    for (int pss = 0; pss < 8; ++pss) {
      const int row = 4 * pss + (lane >> 4), chunk = lane & 15;
      const u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * kORow + chunk * 16);
      if (row0 + row < E)
        *reinterpret_cast<u32x4*>(a.o + (qo0 + row0 + row) * a.o_stride_t + head * a.o_stride_h + 8 * chunk) = v;
    }
    if (a.lse && h == 0 && m_row < E)
      a.lse[(qo0 + m_row) * a.lse_stride_t + head * a.lse_stride_h] = m_run[qb] * kLn2 + __logf(l);
    if (qb == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
#if RX_PW_STAMP
  PW_STAMP(5);   // boundary tiles, drains, epilogue
  if (lane == 0) {  // diagnostic build: the stamps REPLACE the first 32 bytes of the wave's first output row
    uint32_t* dbg = reinterpret_cast<uint32_t*>(a.o + (qo0 + qbase) * a.o_stride_t + head * a.o_stride_h);
    for (int i = 0; i < 8; ++i) dbg[i] = st_acc[i];
    dbg[8] = st_tiles;
    dbg[9] = (uint32_t)(__builtin_amdgcn_s_memtime() - st_t0);          // shader cycles of this wave
    dbg[10] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - st_r0);     // the same interval on the 100-MHz clock
    dbg[11] = (uint32_t)st_r0;                                          // start time (100-MHz ticks, low word)
    uint32_t hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[12] = hwid;
    dbg[13] = xcc;
  }
#endif
}

// what this kernel serves (everything else of head_dim 128 stays with rx_extend32.hip)
bool extend_pw_supports(const rx_extend_params* p) {
  if (p->head_dim != 128 || p->v_head_dim != 128 || p->kv.kv_fp8) return false;
  if (p->custom_mask || p->xai_temperature_len > 0 || p->unified_prefix_lens || p->q_pack > 1 || p->window_kv_offsets ||
      p->sliding_window_size > 0 || p->logit_cap > 0.f || p->v_scale != 1.0f)
    return false;
  const int64_t all = p->q_stride_t | p->q_stride_h | p->k_stride_t | p->k_stride_h | p->v_stride_t | p->v_stride_h |
                      p->kv.k_page_stride | p->kv.k_tok_stride | p->kv.k_head_stride | p->kv.v_page_stride |
                      p->kv.v_tok_stride | p->kv.v_head_stride | p->o_stride_t | p->o_stride_h;
  if (all % 8 != 0) return false;
  if ((((uintptr_t)p->q | (uintptr_t)p->k_extend | (uintptr_t)p->v_extend | (uintptr_t)p->kv.k_buf | (uintptr_t)p->kv.v_buf |
        (uintptr_t)p->o) & 15) != 0)
    return false;
  const bool linear = p->kv.page_size == 1 || (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                                               p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  if (!linear && (p->kv.page_size & (p->kv.page_size - 1)) != 0) return false;
  if (p->skip_prefix && p->skip_extend) return false;
  // one row offset serves the K and the V tensor (the offset table of the tile DMA)
  if (p->k_stride_t != p->v_stride_t) return false;
  if (!p->skip_prefix && (p->kv.k_tok_stride != p->kv.v_tok_stride || p->kv.k_page_stride != p->kv.v_page_stride)) return false;
  return true;
}

int launch_extend_pw(const rx_extend_params* p, hipStream_t s) {
  ExtPwArgs a;
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
  a.bs = p->bs; a.hq = p->num_q_heads; a.hkv = p->num_kv_heads;
  a.group = p->num_q_heads / p->num_kv_heads;
  a.mblocks = (p->max_extend_len + kPwRows - 1) / kPwRows;
  a.sm_scale = p->sm_scale; a.k_scale = p->k_scale;
  a.causal = p->is_causal; a.skip_prefix = p->skip_prefix; a.skip_extend = p->skip_extend;
  a.sinks = p->sinks;
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
#define RX_PW(TT)                                                                                              \
  do {                                                                                                         \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(extend_pw_kernel<TT>),    \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kPwLds);    \
    (void)attr;                                                                                                \
    hipLaunchKernelGGL((extend_pw_kernel<TT>), dim3(grid), dim3(256), kPwLds, s, a);                           \
  } while (0)
  if (p->dtype == RX_BF16) RX_PW(BF16);
  else RX_PW(F16);
#undef RX_PW
  return RX_OK;
}

}  // namespace rx
