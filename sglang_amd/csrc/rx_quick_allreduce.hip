// C3: the "quick" all-reduce -- a two-shot sum all-reduce for LARGE 16-bit messages (prefill-sized o_proj outputs: 16 Ki
// tokens x 8192 x 2 B = 256 MiB at BASELINE configs[3]) whose wire format is block-scaled INT8 / INT6 / INT4 (or plain 16
// bit), so that the xGMI links carry 1/2 .. 1/4 of the bytes.
//
// Reference: QuickAllReduce (srt/distributed/device_communicators/quick_all_reduce.py:41-267), taken by
// GroupCoordinator.all_reduce behind the custom all-reduce and ahead of NCCL (parallel_state.py:886-948); kernel
// kernels/aot/csrc/allreduce/quick_all_reduce.cuh (codecs :52-445, two-shot schedule :448-632), helpers
// quick_all_reduce_base.h.  The ARITHMETIC below is the reference's, operation for operation, because it defines the result:
//   * a message is cut into groups of 64 consecutive elements; a group's EVEN elements form one scale block and its ODD
//     elements another (the 16-bit pairs are processed as packed halves);
//   * block scale: m = the block's value of greatest magnitude WITH its sign (the maximum if |max| > |min|, else the minimum);
//     decode scale d = m * (-1 / R) with R = 128 / 32 / 8; encode scale e = rcp(d + eps); code = rint(clamp(x * e, -R, R - 1))
//     + R -- every step rounded to the 16-bit type (packed fp16 instructions, v_rcp_f16; or bf16 through fp32 with a round to
//     nearest even after every operation, as the HIP bf16 operators do);
//   * decode: (code - R) * d, rounded to the type;
//   * phase 1: every rank encodes its WHOLE input; rank r decodes all W versions of segment r and adds them in rank order
//     0 .. W-1 in the 16-bit type, starting from +0; phase 2: it encodes that sum and everybody decodes it -- so every rank
//     (the owner included) ends with the same bits;
//   * level FP moves the 16-bit values themselves: the result is the rank-order sum in the 16-bit type;
//   * bf16 input may be converted to fp16 first and back at the end (the reference's default: its fp16 kernels are faster).
// The CPU restatement under oracle/ (quick_allreduce) follows exactly this list.
//
// The SCHEDULE is this file's own (MI355X: xGMI is a full mesh of point-to-point links and remote WRITES are posted, remote
// reads are round trips -- so data is pushed, never pulled):
//   * a workgroup of 256 threads owns one 32-KiB tile of the message at a time (thread t: the 16-byte atoms t, t + 256, ..,
//     t + 1792; atoms [r * 8 / W, (r + 1) * 8 / W) are rank r's segment, so all W - 1 links carry an equal share of every tile);
//     at most 1024 workgroups (4 per CU) walk the tiles block-cyclically;
//   * phase 1: it encodes the tile once and writes segment r's payload into rank r's region -- slot (workgroup, source rank) --
//     then raises flag p1[workgroup][me] THERE; waits for its own W flags; reduces; encodes the sum ONCE and writes it to every
//     rank's phase-2 slot; flag p2; wait; decode; store.  Only payload and two 4-byte flags per (tile, peer) cross a link;
//   * a slot belongs to a WORKGROUP INDEX, not to a tile: the workgroup reuses it for its next tile.  That is safe without any
//     further handshake -- a peer has consumed my phase-1 data of tile k before it raises its phase-2 flag of k, which I wait
//     for before I start tile k + 1; and I write phase-2 data of k + 1 to a peer only after its phase-1 flag of k + 1, which
//     it raises after it has finished tile k.  The region is therefore 64 MiB + flags whatever the message size;
//   * flags carry the workgroup's tile counter ("colour"), kept in DEVICE memory and advanced by the kernel, so a launch
//     captured in a HIP graph replays correctly (the reference moved its colours on-device for the same reason,
//     quick_all_reduce.h:34-47) and no flag is ever reset; waits are "at least", bounded, and raise RX_DEVERR_AR_TIMEOUT.
// Never run across xGMI (the pool has one GPU per box): the multi-process tests time-slice one device, which proves the IPC
// mapping, the slot / flag protocol and the arithmetic, not link coherence -- the same caveat as rx_allreduce.hip.
#include "rx_common.h"

namespace rx {

constexpr int kQrThreads = 256;
constexpr int kQrAtoms = 8;                                   // 16-byte atoms per thread and tile
constexpr int kQrTileElems = kQrThreads * kQrAtoms * 8;       // 16384 elements = 32 KiB
constexpr int kQrSlotBytes = kQrTileElems * 2;                // one (workgroup, phase) slot: the FP level's payload
constexpr int kQrMaxBlocks = 1024;
constexpr int kQrMaxWorld = 8;
uint32_t ar_spin_limit();  // rx_allreduce.hip: 2^ar_spin_log2 polls (default 2^27) before a wait raises RX_DEVERR_AR_TIMEOUT

struct QrFlags {
  uint32_t p1[kQrMaxBlocks][kQrMaxWorld];  // p1[b][src]: src's workgroup b has written its phase-1 payload of tile-count `value`
  uint32_t p2[kQrMaxBlocks][kQrMaxWorld];
  uint32_t tiles[kQrMaxBlocks];            // this rank's workgroup b has finished this many tiles (written by that workgroup only)
};
__host__ __device__ constexpr int64_t qr_align(int64_t x) { return (x + 255) / 256 * 256; }
constexpr int64_t kQrDataOff = qr_align(sizeof(QrFlags));
constexpr int64_t kQrPhaseBytes = static_cast<int64_t>(kQrMaxBlocks) * kQrSlotBytes;
constexpr int64_t kQrRegionBytes = kQrDataOff + 2 * kQrPhaseBytes;

struct QrCtx {
  uint32_t magic;
  int rank, world;
  char* peers[kQrMaxWorld];
  int32_t* dev_err;
};
constexpr uint32_t kQrMagic = 0x51524331u;  // guards against a two-shot context handed to the quick entry (and back)

struct QrArgs {
  char* peers[kQrMaxWorld];
  int rank;
  const uint16_t* in;
  uint16_t* out;
  int64_t n;       // elements (multiple of 8)
  int64_t ntiles;
  int32_t* dev_err;
  int32_t fenced;  // option qr_fenced: release / acquire fences around the flags (see qr_signal)
  uint32_t spin_limit;  // polls before a wait gives up (option ar_spin_log2)
};

// ---- 16-bit arithmetic on packed pairs (one 32-bit register = elements 2 i, 2 i + 1) ---------------------------------------
// Constants of a level, as packed pairs of the type: -1 / R, the smallest positive value that keeps rcp finite-ish (the
// reference's 1e-7: fp16's smallest subnormal, bf16 0x33d7), -R, R - 1.
struct NumF16 {
  static constexpr uint32_t kEps = 0x00010001u;
  static __device__ __forceinline__ uint32_t neg_inv_range(int bits) { return bits == 8 ? 0xA000A000u : bits == 6 ? 0xA800A800u : 0xB000B000u; }
  static __device__ __forceinline__ uint32_t range_min(int bits) { return bits == 8 ? 0xD800D800u : bits == 6 ? 0xD000D000u : 0xC800C800u; }
  static __device__ __forceinline__ uint32_t range_max(int bits) { return bits == 8 ? 0x57F057F0u : bits == 6 ? 0x4FC04FC0u : 0x47004700u; }
  static __device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
  static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
  static __device__ __forceinline__ uint32_t vmax(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
  static __device__ __forceinline__ uint32_t vmin(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  }
  static __device__ __forceinline__ uint32_t rcp(uint32_t a) {  // v_rcp_f16 per half (what h2rcp is on this target)
    const _Float16 lo = __builtin_amdgcn_rcph(__builtin_bit_cast(_Float16, static_cast<uint16_t>(a & 0xffffu)));
    const _Float16 hi = __builtin_amdgcn_rcph(__builtin_bit_cast(_Float16, static_cast<uint16_t>(a >> 16)));
    return __builtin_bit_cast(uint16_t, lo) | static_cast<uint32_t>(__builtin_bit_cast(uint16_t, hi)) << 16;
  }
  // per half: the operand of greater magnitude, the SECOND one on a tie or a NaN
  static __device__ __forceinline__ uint32_t pick_abs(uint32_t a, uint32_t b) {
    const _Float16 al = __builtin_bit_cast(_Float16, static_cast<uint16_t>(a & 0x7fffu)), bl = __builtin_bit_cast(_Float16, static_cast<uint16_t>(b & 0x7fffu));
    const _Float16 ah = __builtin_bit_cast(_Float16, static_cast<uint16_t>((a >> 16) & 0x7fffu)), bh = __builtin_bit_cast(_Float16, static_cast<uint16_t>((b >> 16) & 0x7fffu));
    return ((al > bl ? a : b) & 0xffffu) | ((ah > bh ? a : b) & 0xffff0000u);
  }
  // code = rint(w) + R for w in [-R, R - 1]: adding 1024 + R lands in [1024, 2048), where fp16's spacing is 1 -- the add's
  // round-to-nearest-even IS the rint, and the sum's low mantissa bits are the code (the mirror image of the decode below)
  static __device__ __forceinline__ uint32_t to_code(uint32_t w, int bits) {
    const uint32_t magic = bits == 8 ? 0x64806480u : bits == 6 ? 0x64206420u : 0x64086408u;  // 1152, 1056, 1032
    return add(w, magic) & (bits == 8 ? 0x00FF00FFu : bits == 6 ? 0x003F003Fu : 0x000F000Fu);
  }
  static __device__ __forceinline__ uint32_t from_code(uint32_t code, int bits) {  // code - R, exact
    const uint32_t neg = bits == 8 ? 0xE480E480u : bits == 6 ? 0xE420E420u : 0xE408E408u;    // -1152, -1056, -1032
    return add(code | 0x64006400u, neg);
  }
};

struct NumBF16 {
  static constexpr uint32_t kEps = 0x33D733D7u;
  static __device__ __forceinline__ uint32_t neg_inv_range(int bits) { return bits == 8 ? 0xBC00BC00u : bits == 6 ? 0xBD00BD00u : 0xBE00BE00u; }
  static __device__ __forceinline__ uint32_t range_min(int bits) { return bits == 8 ? 0xC300C300u : bits == 6 ? 0xC200C200u : 0xC100C100u; }
  static __device__ __forceinline__ uint32_t range_max(int bits) { return bits == 8 ? 0x42FE42FEu : bits == 6 ? 0x41F841F8u : 0x40E040E0u; }
  static __device__ __forceinline__ float lo(uint32_t a) { return __builtin_bit_cast(float, a << 16); }
  static __device__ __forceinline__ float hi(uint32_t a) { return __builtin_bit_cast(float, a & 0xffff0000u); }
  static __device__ __forceinline__ uint32_t pack(float l, float h) { return pack2<BF16>(l, h); }  // round to nearest even
  static __device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b) { return pack(lo(a) * lo(b), hi(a) * hi(b)); }
  static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return pack(lo(a) + lo(b), hi(a) + hi(b)); }
  // __hmax / __hmin of the HIP bf16 header: a NaN loses, otherwise a > b ? a : b -- the SECOND operand on equality (+0 / -0)
  static __device__ __forceinline__ float max1(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }
  static __device__ __forceinline__ float min1(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
  static __device__ __forceinline__ uint32_t bits_of(float l, float h) {  // (operands are bf16 values: no rounding)
    return (__builtin_bit_cast(uint32_t, l) >> 16) | (__builtin_bit_cast(uint32_t, h) & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t vmax(uint32_t a, uint32_t b) { return bits_of(max1(lo(a), lo(b)), max1(hi(a), hi(b))); }
  static __device__ __forceinline__ uint32_t vmin(uint32_t a, uint32_t b) { return bits_of(min1(lo(a), lo(b)), min1(hi(a), hi(b))); }
  static __device__ __forceinline__ uint32_t rcp(uint32_t a) { return pack(1.0f / lo(a), 1.0f / hi(a)); }  // (IEEE division, then one rounding)
  static __device__ __forceinline__ uint32_t pick_abs(uint32_t a, uint32_t b) {
    return ((fabsf(lo(a)) > fabsf(lo(b)) ? a : b) & 0xffffu) | ((fabsf(hi(a)) > fabsf(hi(b)) ? a : b) & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t to_code(uint32_t w, int bits) {
    const int r = 1 << (bits - 1);
    return static_cast<uint32_t>(static_cast<int>(rintf(lo(w))) + r) | static_cast<uint32_t>(static_cast<int>(rintf(hi(w))) + r) << 16;
  }
  static __device__ __forceinline__ uint32_t from_code(uint32_t code, int bits) {  // code - R: an integer of at most 8 bits, exact in bf16
    const int r = 1 << (bits - 1);
    return bits_of(static_cast<float>(static_cast<int>(code & 0xffffu) - r), static_cast<float>(static_cast<int>(code >> 16) - r));
  }
};

// lane l <- lane l + N of its row of 16 (lanes whose source falls off the row keep their own value: never a group leader's
// chain, see group_scale)
template <int N>
__device__ __forceinline__ uint32_t qr_from_above(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x100 | N, 0xf, 0xf, false));
}

// One atom (8 elements as 4 packed pairs) of every lane of an 8-lane group = 64 consecutive elements -> the group's decode
// scale (packed: even block | odd block), on every lane.  Order of the comparisons as the reference's (group_abs_max,
// quick_all_reduce_base.h:268-300): inside the lane (0,1) (2,3) then the two; across lanes a binary tree towards the leader,
// own value first -- with bf16's "second operand on equality" rule the order decides the sign of an all-zero block's scale.
template <typename NT>
__device__ __forceinline__ uint32_t qr_group_scale(const u32x4 atom, int bits) {
  uint32_t mx = NT::vmax(NT::vmax(atom[0], atom[1]), NT::vmax(atom[2], atom[3]));
  uint32_t mn = NT::vmin(NT::vmin(atom[0], atom[1]), NT::vmin(atom[2], atom[3]));
  mx = NT::vmax(mx, qr_from_above<1>(mx));
  mn = NT::vmin(mn, qr_from_above<1>(mn));
  mx = NT::vmax(mx, qr_from_above<2>(mx));
  mn = NT::vmin(mn, qr_from_above<2>(mn));
  mx = NT::vmax(mx, qr_from_above<4>(mx));
  mn = NT::vmin(mn, qr_from_above<4>(mn));
  const uint32_t m = NT::pick_abs(mx, mn);
  const uint32_t d = NT::mul(m, NT::neg_inv_range(bits));
  return static_cast<uint32_t>(__shfl(static_cast<int>(d), (threadIdx.x & 63) & ~7));  // the leader's, to its seven followers
}

// ---- wire format of one ROW (the same atom index of all 256 threads), level by level ---------------------------------------
//   16: 256 x 16 B                                       (4096 B)
//    8: 256 x 8 B codes | 32 x 4 B scales                (2176 B)
//    6: 256 x 4 B low nibbles | 256 x 2 B top bits | 32 x 4 B scales   (1664 B)
//    4: 256 x 4 B codes | 32 x 4 B scales                (1152 B)
template <int BITS>
struct QrRow {
  static constexpr int kCodeBytes = BITS == 16 ? 16 : BITS == 8 ? 8 : 4;
  static constexpr int kTopAt = kQrThreads * kCodeBytes;                            // (BITS == 6 only)
  static constexpr int kScaleAt = kTopAt + (BITS == 6 ? kQrThreads * 2 : 0);
  static constexpr int kBytes = kScaleAt + (BITS == 16 ? 0 : (kQrThreads / 8) * 4);
};

struct QrPacked {
  uint32_t w[4];   // FP: the atom; 8: w[0..1]; 6: w[0] low nibbles, w[1] top bits (16 of them); 4: w[0]
  uint32_t scale;
};

template <typename NT, int BITS>
__device__ __forceinline__ QrPacked qr_encode(const u32x4 atom) {
  QrPacked p;
  if constexpr (BITS == 16) {
    p.w[0] = atom[0]; p.w[1] = atom[1]; p.w[2] = atom[2]; p.w[3] = atom[3];
    p.scale = 0;
    return p;
  } else {
    const uint32_t d = qr_group_scale<NT>(atom, BITS);
    const uint32_t e = NT::rcp(NT::add(d, NT::kEps));
    uint32_t c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      c[i] = NT::to_code(NT::vmin(NT::vmax(NT::mul(atom[i], e), NT::range_min(BITS)), NT::range_max(BITS)), BITS);
    p.scale = d;
    p.w[2] = p.w[3] = 0;
    if constexpr (BITS == 8) {
      p.w[0] = c[0] | c[1] << 8;
      p.w[1] = c[2] | c[3] << 8;
    } else if constexpr (BITS == 4) {
      p.w[0] = c[0] | c[1] << 4 | c[2] << 8 | c[3] << 12;
      p.w[1] = 0;
    } else {
      p.w[0] = (c[0] & 0x000F000Fu) | (c[1] & 0x000F000Fu) << 4 | (c[2] & 0x000F000Fu) << 8 | (c[3] & 0x000F000Fu) << 12;
      const uint32_t t = (c[0] >> 4 & 0x00030003u) | (c[1] >> 4 & 0x00030003u) << 2 | (c[2] >> 4 & 0x00030003u) << 4 |
                         (c[3] >> 4 & 0x00030003u) << 6;   // bits 0-7: the even elements' top bits, 16-23: the odd ones'
      p.w[1] = (t & 0xffu) | (t >> 8 & 0xff00u);
    }
    return p;
  }
}

// Region accesses go through buffer descriptors: one descriptor per rank's region (base in SGPRs), the slot / sub-slot / row
// offset as the instruction's scalar offset and `tid * width` as its only vector offset.  With flat 64-bit addresses hipcc
// hoists ~100 loop-invariant address pairs out of the tile loop (67 stores + 40 loads at INT6, W = 8: 250 VGPRs and scratch).
typedef __amdgpu_buffer_rsrc_t qr_rsrc;
__device__ __forceinline__ qr_rsrc qr_region(char* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, static_cast<int>(kQrRegionBytes), 0x00020000);
}
constexpr int kQrNt = 2;  // aux: non-temporal (streamed once; the region is uncached device memory anyway)

// (stores to a peer's region are made visible by the fence + release in front of the flag)
template <int BITS>
__device__ __forceinline__ void qr_store_row(qr_rsrc dst, int row, const QrPacked& p, int tid) {
  using R = QrRow<BITS>;
  if constexpr (BITS == 16) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{p.w[0], p.w[1], p.w[2], p.w[3]}, dst, tid * 16, row, kQrNt);
  } else {
    if constexpr (BITS == 8) __builtin_amdgcn_raw_buffer_store_b64(u32x2{p.w[0], p.w[1]}, dst, tid * 8, row, kQrNt);
    else __builtin_amdgcn_raw_buffer_store_b32(p.w[0], dst, tid * 4, row, kQrNt);
    if constexpr (BITS == 6) __builtin_amdgcn_raw_buffer_store_b16(static_cast<uint16_t>(p.w[1]), dst, tid * 2, row + R::kTopAt, kQrNt);
    if ((tid & 7) == 0) __builtin_amdgcn_raw_buffer_store_b32(p.scale, dst, (tid >> 3) * 4, row + R::kScaleAt, kQrNt);
  }
}

template <typename NT, int BITS>
__device__ __forceinline__ u32x4 qr_load_row(qr_rsrc src, int row, int tid) {
  using R = QrRow<BITS>;
  if constexpr (BITS == 16) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(src, tid * 16, row, kQrNt));
  } else {
    uint32_t c[4];
    if constexpr (BITS == 8) {
      const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(src, tid * 8, row, kQrNt));
      c[0] = w[0] & 0x00FF00FFu; c[1] = w[0] >> 8 & 0x00FF00FFu; c[2] = w[1] & 0x00FF00FFu; c[3] = w[1] >> 8 & 0x00FF00FFu;
    } else {
      const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(src, tid * 4, row, kQrNt);
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = w >> (4 * i) & 0x000F000Fu;
      if constexpr (BITS == 6) {
        const uint32_t t16 = __builtin_amdgcn_raw_buffer_load_b16(src, tid * 2, row + R::kTopAt, kQrNt);
        const uint32_t t = (t16 & 0xffu) | (t16 & 0xff00u) << 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] |= (t >> (2 * i) & 0x00030003u) << 4;
      }
    }
    const uint32_t d = __builtin_amdgcn_raw_buffer_load_b32(src, (tid >> 3) * 4, row + R::kScaleAt, kQrNt);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = NT::mul(NT::from_code(c[i], BITS), d);
    return o;
  }
}

// Publication of a slot (what makes the payload visible before the flag), default form:
//   every wave waits until ITS region stores have been ACKNOWLEDGED (vmcnt counts stores on gfx950; the region is uncached
//   memory -- rx_ar_alloc_region -- so an acknowledged store has left this GPU's caches for the memory it targets) ->
//   workgroup barrier -> the signalling lanes store the flag (a relaxed system-scope atomic: nothing is left to order).
// Consumption: the polling lanes spin on relaxed system-scope loads, the workgroup barrier, then the slot is read with
// non-temporal loads of uncached memory (never served from a cache of this GPU).
// Option qr_fenced = 1 is the memory-model form of the same handshake -- release store of the flag (buffer_wbl2 + wait), an
// acquire fence in every wave behind the wait (buffer_inv) -- for a platform where the acknowledgement argument should not
// hold; it is what the first version of this kernel did, and it is expensive because the write-back and the invalidate
// are whole-L2 operations issued twice per tile and workgroup.  Round 6, two processes on ONE GPU, 64 MiB fp16, ms per call
// (FP / INT4): __threadfence_system() in every wave + acquire loads in the spin 1.38 / 1.17; release store + one acquire
// fence per wave 0.53 / 0.46 (= qr_fenced); release store only 0.38 / 0.27; acquire fence only 0.30 / 0.29; neither (default)
// 0.127 / 0.120 -- at FP that is 6 bytes of HBM traffic per message byte and rank, 6.2 TB/s for the two ranks: the device's
// copy rate, i.e. the handshake is hidden.  (One GPU: no link was involved; see the file header.)
__device__ __forceinline__ void qr_stores_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0)
__device__ __forceinline__ void qr_signal(uint32_t* p, uint32_t v, bool fenced) {
  if (fenced) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool qr_wait(const uint32_t* p, uint32_t v, uint32_t limit) {
  for (uint32_t i = 0; i < limit; ++i) {
    if (static_cast<int32_t>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - v) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
__device__ __forceinline__ void qr_acquire(bool fenced) {
  if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope
}

__device__ __forceinline__ uint32_t qr_bf16x2_to_f16x2(uint32_t v) {
  const _Float16 l = static_cast<_Float16>(__builtin_bit_cast(float, v << 16)), h = static_cast<_Float16>(__builtin_bit_cast(float, v & 0xffff0000u));
  return __builtin_bit_cast(uint16_t, l) | static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h)) << 16;
}
__device__ __forceinline__ uint32_t qr_f16x2_to_bf16x2(uint32_t v) {
  return pack2<BF16>(static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(v & 0xffffu))),
                     static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(v >> 16))));
}

// NT: the arithmetic (NumF16 / NumBF16); CAST: the tensors are bf16 and travel as fp16 (NT = NumF16)
template <typename NT, int BITS, int W, bool CAST>
__global__ __launch_bounds__(kQrThreads, 4) void quick_allreduce_kernel(const QrArgs a) {
  constexpr int RA = kQrAtoms / W;              // atoms (= rows) of one rank's segment
  constexpr int kRow = QrRow<BITS>::kBytes;
  constexpr int kSub = RA * kRow;               // one source rank's share of a slot
  static_assert(W * kSub <= kQrSlotBytes, "slot geometry");
  const int b = blockIdx.x, tid = threadIdx.x, me = a.rank;
  __shared__ uint32_t colour_s;
  __shared__ int timeout_s;
  QrFlags* const my_flags = reinterpret_cast<QrFlags*>(a.peers[me]);
  if (tid == 0) {
    colour_s = __hip_atomic_load(&my_flags->tiles[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    timeout_s = 0;
  }
  __syncthreads();
  uint32_t colour = colour_s + 1;               // flags say "tile count >= colour"
  static_assert(kQrRegionBytes < (1ll << 31), "region offsets are 32-bit buffer offsets");
  const int p1_slot = static_cast<int>(kQrDataOff) + b * kQrSlotBytes;
  const int p2_slot = p1_slot + static_cast<int>(kQrPhaseBytes);
  const qr_rsrc mine = qr_region(a.peers[me]);
  const u32x4* in_v = reinterpret_cast<const u32x4*>(a.in);
  u32x4* out_v = reinterpret_cast<u32x4*>(a.out);
  const int64_t nv = a.n / 8;

  for (int64_t tile = b; tile < a.ntiles; tile += gridDim.x, ++colour) {
    // ---- the tile: atoms past the end of the message are zeros (and join their group's scale as zeros)
    u32x4 atom[kQrAtoms];
    const int64_t v0 = tile * (kQrThreads * kQrAtoms) + tid;
#pragma unroll
    for (int i = 0; i < kQrAtoms; ++i) {
      const int64_t v = v0 + i * kQrThreads;
      atom[i] = v < nv ? in_v[v] : u32x4{0u, 0u, 0u, 0u};
      if constexpr (CAST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atom[i][j] = qr_bf16x2_to_f16x2(atom[i][j]);
      }
    }
    // ---- phase 1: segment r -> rank r
#pragma unroll
    for (int r = 0; r < W; ++r) {
      const qr_rsrc dst = qr_region(a.peers[r]);
#pragma unroll
      for (int k = 0; k < RA; ++k) qr_store_row<BITS>(dst, p1_slot + me * kSub + k * kRow, qr_encode<NT, BITS>(atom[r * RA + k]), tid);
    }
    qr_stores_done();
    __syncthreads();
    if (tid < W) {
      qr_signal(&reinterpret_cast<QrFlags*>(a.peers[tid])->p1[b][me], colour, a.fenced != 0);
      if (!qr_wait(&my_flags->p1[b][tid], colour, a.spin_limit)) timeout_s = 1;
    }
    __syncthreads();
    if (timeout_s) break;
    qr_acquire(a.fenced != 0);
    // ---- my segment: the W versions, decoded and added in rank order in the 16-bit type
    u32x4 acc[RA];
#pragma unroll
    for (int k = 0; k < RA; ++k) acc[k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < W; ++r) {
#pragma unroll
      for (int k = 0; k < RA; ++k) {
        const u32x4 x = qr_load_row<NT, BITS>(mine, p1_slot + r * kSub + k * kRow, tid);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = NT::add(acc[k][j], x[j]);
      }
    }
    // ---- phase 2: the sum, encoded once, to everybody (myself included: every rank decodes the same payload)
#pragma unroll
    for (int k = 0; k < RA; ++k) {
      const QrPacked p = qr_encode<NT, BITS>(acc[k]);
#pragma unroll
      for (int r = 0; r < W; ++r) qr_store_row<BITS>(qr_region(a.peers[r]), p2_slot + me * kSub + k * kRow, p, tid);
    }
    qr_stores_done();
    __syncthreads();
    if (tid < W) {
      qr_signal(&reinterpret_cast<QrFlags*>(a.peers[tid])->p2[b][me], colour, a.fenced != 0);
      if (!qr_wait(&my_flags->p2[b][tid], colour, a.spin_limit)) timeout_s = 1;
    }
    __syncthreads();
    if (timeout_s) break;
    qr_acquire(a.fenced != 0);
#pragma unroll
    for (int r = 0; r < W; ++r) {
#pragma unroll
      for (int k = 0; k < RA; ++k) atom[r * RA + k] = qr_load_row<NT, BITS>(mine, p2_slot + r * kSub + k * kRow, tid);
    }
#pragma unroll
    for (int i = 0; i < kQrAtoms; ++i) {
      const int64_t v = v0 + i * kQrThreads;
      if constexpr (CAST) {
#pragma unroll
        for (int j = 0; j < 4; ++j) atom[i][j] = qr_f16x2_to_bf16x2(atom[i][j]);
      }
      if (v < nv) out_v[v] = atom[i];
    }
  }
  if (tid == 0) {
    if (timeout_s) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    __hip_atomic_store(&my_flags->tiles[b], colour - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// test support: v_rcp_f16 of every fp16 bit pattern (the encode scale of the fp16 codecs is this instruction's output, which
// the ISA specifies to 1 ulp only -- the oracle takes the table instead of assuming a rounding)
__global__ void rcp_f16_table_kernel(uint16_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 65536u) out[i] = __builtin_bit_cast(uint16_t, __builtin_amdgcn_rcph(__builtin_bit_cast(_Float16, static_cast<uint16_t>(i))));
}

template <typename NT, int BITS, bool CAST>
static void qr_launch_world(const QrArgs& a, int world, unsigned grid, hipStream_t s) {
  if (world == 2) hipLaunchKernelGGL((quick_allreduce_kernel<NT, BITS, 2, CAST>), dim3(grid), dim3(kQrThreads), 0, s, a);
  else if (world == 4) hipLaunchKernelGGL((quick_allreduce_kernel<NT, BITS, 4, CAST>), dim3(grid), dim3(kQrThreads), 0, s, a);
  else hipLaunchKernelGGL((quick_allreduce_kernel<NT, BITS, 8, CAST>), dim3(grid), dim3(kQrThreads), 0, s, a);
}
template <typename NT, bool CAST>
static void qr_launch_level(const QrArgs& a, int level, int world, unsigned grid, hipStream_t s) {
  switch (level) {
    case RX_QR_INT8: qr_launch_world<NT, 8, CAST>(a, world, grid, s); break;
    case RX_QR_INT6: qr_launch_world<NT, 6, CAST>(a, world, grid, s); break;
    case RX_QR_INT4: qr_launch_world<NT, 4, CAST>(a, world, grid, s); break;
    default: qr_launch_world<NT, 16, CAST>(a, world, grid, s); break;
  }
}

}  // namespace rx

using namespace rx;

extern "C" {

int64_t rx_qr_region_bytes(void) { return kQrRegionBytes; }

int rx_qr_init(rx_qr_ctx** ctx_out, int rank, int world, void* const* peer_regions, int32_t* dev_err) {
  RX_REQUIRE(ctx_out && peer_regions && dev_err, "rx_qr_init: null pointer");
  // (quick_all_reduce.cu:10-18: 2, 4 or 8 ranks -- a tile's eight atoms split evenly)
  RX_REQUIRE((world == 2 || world == 4 || world == 8) && rank >= 0 && rank < world, "rx_qr_init: rank %d / world %d (2, 4 or 8 ranks)",
             rank, world);
  auto* c = new QrCtx();
  c->magic = kQrMagic;
  c->rank = rank;
  c->world = world;
  c->dev_err = dev_err;
  for (int i = 0; i < world; ++i) {
    if (!peer_regions[i]) {
      delete c;
      return fail(RX_ERR_INVALID_ARG, "rx_qr_init: peer region %d is null", i);
    }
    c->peers[i] = static_cast<char*>(peer_regions[i]);
  }
  *ctx_out = reinterpret_cast<rx_qr_ctx*>(c);
  return RX_OK;
}

int rx_quick_allreduce(rx_qr_ctx* ctx, const void* in, void* out, int64_t count, int dtype, int quant_level,
                       int cast_bf16_to_fp16, void* stream) {
  RX_RANGE("rx_quick_allreduce");
  RX_REQUIRE(ctx && in && out, "rx_quick_allreduce: null pointer");
  auto* c = reinterpret_cast<QrCtx*>(ctx);
  RX_REQUIRE(c->magic == kQrMagic, "rx_quick_allreduce: not a context of rx_qr_init");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_quick_allreduce: dtype %d", dtype);
  RX_REQUIRE(quant_level >= RX_QR_FP && quant_level <= RX_QR_INT4, "rx_quick_allreduce: quant_level %d", quant_level);
  RX_REQUIRE(count >= 0 && count % 8 == 0, "rx_quick_allreduce: count %lld must be a multiple of 8", (long long)count);
  RX_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "rx_quick_allreduce: in/out must be 16-byte aligned");
  if (count == 0) return RX_OK;
  QrArgs a{};
  for (int i = 0; i < c->world; ++i) a.peers[i] = c->peers[i];
  a.rank = c->rank;
  a.in = static_cast<const uint16_t*>(in);
  a.out = static_cast<uint16_t*>(out);
  a.n = count;
  a.ntiles = (count + kQrTileElems - 1) / kQrTileElems;
  a.dev_err = c->dev_err;
  a.fenced = options().qr_fenced;
  a.spin_limit = ar_spin_limit();
  const int cap = options().qr_max_blocks;  // (tests: a small grid walks many tiles per workgroup on a small message)
  const unsigned grid = static_cast<unsigned>(std::min<int64_t>(a.ntiles, cap > 0 && cap < kQrMaxBlocks ? cap : kQrMaxBlocks));
  auto s = static_cast<hipStream_t>(stream);
  const char* lv = quant_level == RX_QR_INT8 ? "8" : quant_level == RX_QR_INT6 ? "6" : quant_level == RX_QR_INT4 ? "4" : "16";
  if (dtype == RX_F16) {
    note_dispatch("quick_allreduce_kernel<rx::NumF16, %s, %d, false>", lv, c->world);
    qr_launch_level<NumF16, false>(a, quant_level, c->world, grid, s);
  } else if (cast_bf16_to_fp16) {
    note_dispatch("quick_allreduce_kernel<rx::NumF16, %s, %d, true>", lv, c->world);
    qr_launch_level<NumF16, true>(a, quant_level, c->world, grid, s);
  } else {
    note_dispatch("quick_allreduce_kernel<rx::NumBF16, %s, %d, false>", lv, c->world);
    qr_launch_level<NumBF16, false>(a, quant_level, c->world, grid, s);
  }
  return check_launch("rx_quick_allreduce");
}

int rx_qr_destroy(rx_qr_ctx* ctx) {
  auto* c = reinterpret_cast<QrCtx*>(ctx);
  if (c && c->magic != kQrMagic) return fail(RX_ERR_INVALID_ARG, "rx_qr_destroy: not a context of rx_qr_init");
  delete c;
  return RX_OK;
}

int rx_rcp_f16_table(uint16_t* dev_out_65536, void* stream) {
  RX_REQUIRE(dev_out_65536, "rx_rcp_f16_table: null pointer");
  hipLaunchKernelGGL(rcp_f16_table_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), dev_out_65536);
  return check_launch("rx_rcp_f16_table");
}

}  // extern "C"
