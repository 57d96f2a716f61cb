// Shared host/device helpers for libradix_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <type_traits>

#include "../../include/radix_hip.h"

namespace rx {

// thread-local last-error string (rx_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

// ---- dispatch switches and the dispatch record ---------------------------------------------------------------
// Everything that picks a kernel instance beyond the call's own arguments lives in ONE struct, filled once when the
// library is loaded (defaults; RX_OPT_<NAME> in the environment at load time overrides one) and changed only through
// rx_set_option() -- the launch path reads plain ints, never the environment.  Tests flip switches through
// rx_set_option and learn which instance a call took from rx_last_dispatch().
struct Options {
  int ext32_autopack = 1;     // long causal GQA-4 / GQA-8 extends at D = 128 pack their rows by themselves
  int ext32_small_wg = -1;    // D = 128 workgroup form: -1 by the tile estimate, 0 always eight waves, 1 always four
  int ext32_small_wg_tiles = 28;  // ... the estimate below which the four-wave form runs (unpacked calls)
  int ext32_pack_min_len = 1;     // self-packing from this many new tokens (longest request) up ...
  int ext32_pack_min_tiles = 4;   // ... and this tile estimate; packed PLAIN rows take eight waves from here
  int ext32_pack_min_wgs = -1;    // ... while the packed grid holds this many workgroups (-1: the device's CU count)
  int ext32_pack4_tiles = 24;    // ... and FOUR waves (two workgroups per CU) below this tile estimate
  int ext32_plain = 1;        // PLAIN instances (features compile-time off) when the call uses none of them
  int ext32_uni = 1;          // the unified (deterministic) extend on its own instance: pipelined, row-deterministic (0: the general instance's masked body)
  int roctx = 0;              // roctx ranges around the entry points' launches (RX_RANGE)
  int ext32_count_redo = 0;   // debug: the bench-shaped packed call runs the COUNTING instance (rx_debug_counters)
  int ext64 = 0;              // dev builds (RX_WITH_EXT64=1) only: PLAIN eight-wave calls take tools/probe/rx_extend64.hip's kernel; no effect in the product library
  int extend_16x16_d128 = 0;  // the 16x16x32 kernel of rx_extend.hip for plain D = 128 calls (A/B of the two shapes)
  int extend_d256 = 1;        // the AGPR / LDS-DMA template (256, 192, 96, 64) where it supports the call
  int extend_d256_min_rows = 1;    // ... for short extends (<= 128 rows of the longest request) from this many rows up, given >= 4 estimated tiles
  int extend_d256_at128 = 0;  // ... instantiated at 128 / 128 (A/B against rx_extend32)
  int extend_d256_at64 = 1;   // ... at 64 / 64 (0: extend_mfma_kernel)
  int extend_d256_at96 = 1;   // ... at 96 / 96 (0: extend_nd_kernel)
  int extend_nd = 1;          // extend_nd_kernel for the other head dims
  int extend_nd_big = 1;      // its eight-wave form for Dk > 128
  int extend_mla = 1;         // the latent (576 / 512) extend kernel
  int extend_mla_shared_v = 1;  // V^T read from the K image when v aliases k[..., :512]
  int decode_mla8_t64 = 0;    // ... in its 64-token-tile form (every wave owns 16 tokens of the tile; 0: the 32-token K-split form; A/B: DESIGN 4.1b)
  int merge_in_kernel_max_mb = 4;      // stage 2 inside the stage-1 kernel while the partials are at most this many MB ...
  int merge_in_kernel_max_mb_mla = 4;  // ... and for latent (576 / 512) rows
  int ar_spin_log2 = 27;      // all-reduce kernels: a flag wait gives up (RX_DEVERR_AR_TIMEOUT) after 2^this polls (~1-2 us each); bench.py's first xGMI contact lowers it
  int ar_blocks = 0;          // peer-to-peer all-reduce kernels: blocks per launch (0: by message size; at most 256)
  int ar_fenced = 0;          // peer-to-peer all-reduce kernels: 1 = system fences / release / acquire around their flags (rx_allreduce.hip)
  int qr_fenced = 0;          // quick all-reduce: 1 = release store / acquire fence around its flags instead of acknowledged uncached stores (rx_quick_allreduce.hip)
  int qr_max_blocks = 0;      // quick all-reduce: cap on its grid (0: 1024 = 4 workgroups per CU); tests walk many tiles per workgroup on small messages
  int decode_mla8_dma = 1;    // fp8 latent rows through the LDS-DMA kernel (0: upcast-while-staging form)
};
Options& options();

// ---- optional tooling hooks (round 5; SURVEY 5: named ranges around the library's launches, an argument dump at the C-ABI
// boundary as sglang/kernel_api_logging.py:52-60 has for the reference's kernel API) ----------------------------------
// RX_RANGE("rx_decode_attn"): a roctx range around the entry point's launches when option `roctx` is on (RX_OPT_ROCTX=1 at
// load, or rx_set_option): rocprofv3 --marker-trace then names store / metadata / decode / extend in the timeline.
// libroctx64.so is dlopen'ed on first use; absent, the option does nothing.
struct RangeGuard {
  bool on;
  explicit RangeGuard(const char* name);
  ~RangeGuard();
};
#define RX_RANGE(name) ::rx::RangeGuard rx_range_guard_(name)
// RX_DUMP_DIR (read once at load): an entry point that returns a non-zero status leaves <dir>/rx_<entry>_<pid>_<n>.txt (status,
// rx_last_error) and .bin (the raw parameter struct: tools/decode_dump.py prints its fields through the ctypes binding).
int dump_on_error(const char* entry, int status, const void* params, size_t bytes);
// CU count of the CURRENT device (cached per device id: a process may drive GPUs of different sizes, and the first call
// may come before its hipSetDevice -- ADVICE r4).  256 when the query fails.
int device_cu_count();
// name of the kernel instance the calling thread's last rx_extend_attn / rx_decode_attn launched (rx_last_dispatch)
void note_dispatch(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RX_OK;
}

#define RX_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return ::rx::fail(RX_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;  // CDNA4 wavefront

__device__ __forceinline__ int64_t load_idx(const void* p, int64_t i, bool is64) {
  return is64 ? reinterpret_cast<const int64_t*>(p)[i]
              : static_cast<int64_t>(reinterpret_cast<const int32_t*>(p)[i]);
}

// ---- 16-bit float helpers -------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;  // builtin's type
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef u32x2 __attribute__((aligned(4))) u32x2_a4;  // an 8-byte load from an address that is only 4-byte aligned (int32 pairs at odd i)
// p[i] and p[i + 1] of an int32 or int64 array, branch-free: two 8-byte loads issued back to back (a load behind an
// `is64 ?` branch is waited for at the join, so two load_idx calls cost two dependent round trips).  int32: the first
// load holds both entries and the second re-reads the same address (no byte past p[i + 1] is touched).
__device__ __forceinline__ void load_idx_pair(const void* p, int64_t i, bool is64, int64_t& v0, int64_t& v1) {
  const char* b = reinterpret_cast<const char*>(p);
  const u32x2 x = *reinterpret_cast<const u32x2_a4*>(b + (is64 ? i * 8 : i * 4));
  const u32x2 y = *reinterpret_cast<const u32x2_a4*>(b + (is64 ? (i + 1) * 8 : i * 4));
  v0 = is64 ? static_cast<int64_t>((static_cast<uint64_t>(x[1]) << 32) | x[0]) : static_cast<int64_t>(static_cast<int32_t>(x[0]));
  v1 = is64 ? static_cast<int64_t>((static_cast<uint64_t>(y[1]) << 32) | y[0]) : static_cast<int64_t>(static_cast<int32_t>(x[1]));
}


struct BF16 {
  using vec8 = bf16x8;
  using scalar = __bf16;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // ds_read_b64_tr_b16: 4 rows x 16 cols block per 16 lanes, delivered column-major
  static __device__ __forceinline__ u32x2 ds_read_tr(const void* lds_addr) {
    auto p = (__attribute__((address_space(3))) bf16x4*)(uintptr_t)(uint32_t)(uintptr_t)lds_addr;
    bf16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
    return __builtin_bit_cast(u32x2, r);
  }
  static __device__ __forceinline__ float to_f32(uint16_t b) {
    return __builtin_bit_cast(float, static_cast<uint32_t>(b) << 16);
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    __bf16 h = static_cast<__bf16>(f);  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, h);
  }
};

struct F16 {
  using vec8 = f16x8;
  using scalar = _Float16;
  static __device__ __forceinline__ f32x4 mfma(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ u32x2 ds_read_tr(const void* lds_addr) {
    auto p = (__attribute__((address_space(3))) fp16x4_t*)(uintptr_t)(uint32_t)(uintptr_t)lds_addr;
    fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    return __builtin_bit_cast(u32x2, r);
  }
  static __device__ __forceinline__ float to_f32(uint16_t b) {
    return static_cast<float>(__builtin_bit_cast(_Float16, b));
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f));
  }
};

// template arguments as c++filt prints them: the dispatch record names an instance the way its symbol does
template <typename T>
inline const char* tname() {
  if constexpr (std::is_same_v<T, BF16>) return "rx::BF16";
  else if constexpr (std::is_same_v<T, F16>) return "rx::F16";
  else if constexpr (std::is_same_v<T, int64_t>) return "long";
  else return "int";
}
inline const char* tbool(bool b) { return b ? "true" : "false"; }

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// eight OCP fp8 e4m3fn bytes -> eight 16-bit floats of T (exact: both bf16 and f16 hold every e4m3
// value).  One v_cvt_scalef32_pk_{bf16,f16}_fp8 per pair, scale 1.0.
template <typename T>
__device__ __forceinline__ u32x4 fp8x8_to_16(u32x2 raw) {
  u32x4 out;
  if constexpr (__is_same(typename T::scalar, __bf16)) {
    out[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw[0], 1.0f, false));
    out[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw[0], 1.0f, true));
    out[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw[1], 1.0f, false));
    out[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw[1], 1.0f, true));
  } else {
    out[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw[0], 1.0f, false));
    out[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw[0], 1.0f, true));
    out[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw[1], 1.0f, false));
    out[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(raw[1], 1.0f, true));
  }
  return out;
}

// two fp32 -> one dword of two 16-bit floats (RNE).  The vector convert lets hipcc emit ONE
// v_cvt_pk_bf16_f32 per pair (two scalar casts cost cvt + cvt + shift + or).
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  if constexpr (sizeof(typename T::scalar) == 2 && __is_same(typename T::scalar, __bf16)) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
  } else {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));
  }
}

// KV slot -> element offset with 32-bit operands (slots and per-token strides are < 2^31):
// one v_mad_u64_u32 instead of a 64x64-bit multiply.
__device__ __forceinline__ int64_t mul_u32(int64_t slot, int64_t stride) {
  return static_cast<int64_t>(static_cast<uint64_t>(static_cast<uint32_t>(slot)) *
                              static_cast<uint32_t>(stride));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- stage 2 inside the stage-1 kernel ("the last workgroup to arrive merges") --------------------------------
// With rx_decode_params.merge_counters the kv-split kernels do not need the stage-2 launch: every workgroup of a
// (request, head block) publishes its partial, bumps the block's counter, and the one that sees `live - 1` resets the
// counter for the next launch and merges all `live` partials of its head block.  The partials cross XCDs, whose L2s
// are not coherent with each other.  Agent-scope release / acquire FENCES would do (buffer_wbl2 + buffer_inv), but
// they write back and invalidate a whole L2 per workgroup: measured 67 -> 142 us on the MLA shape and 121 -> 750 us
// at bs 64 x 2 k.  Instead only the partials themselves take the slow path: device-scope (sc0 sc1) write-through
// stores and device-scope loads, plus the counter's device-scope atomic; everything else keeps its cached accesses.
__device__ __forceinline__ void store_dev(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_dev(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// One round trip for a chunk of 8 splits: the 8 LSEs (two 16-B loads) and the 8 partial rows' 16 B of this thread, all
// device-scope, issued together and waited for once.  (One asm statement: between a load and its wait hipcc must not
// touch the destination registers, and it cannot know that.)
__device__ __forceinline__ void load_dev_chunk8(const float* lse8, const float* const (&row)[8], f32x4& la, f32x4& lb,
                                                f32x4 (&x)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %10, off sc0 sc1\n\tglobal_load_dwordx4 %1, %10, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %11, off sc0 sc1\n\tglobal_load_dwordx4 %3, %12, off sc0 sc1\n\t"
      "global_load_dwordx4 %4, %13, off sc0 sc1\n\tglobal_load_dwordx4 %5, %14, off sc0 sc1\n\t"
      "global_load_dwordx4 %6, %15, off sc0 sc1\n\tglobal_load_dwordx4 %7, %16, off sc0 sc1\n\t"
      "global_load_dwordx4 %8, %17, off sc0 sc1\n\tglobal_load_dwordx4 %9, %18, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(la), "=&v"(lb), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]),
        "=&v"(x[7])
      : "v"(lse8), "v"(row[0]), "v"(row[1]), "v"(row[2]), "v"(row[3]), "v"(row[4]), "v"(row[5]), "v"(row[6]), "v"(row[7])
      : "memory");
}
__device__ __forceinline__ void load_dev_lse8(const float* lse8, f32x4& la, f32x4& lb) {
  asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(la), "=&v"(lb)
               : "v"(lse8)
               : "memory");
}
// Returns true in the workgroup that has to merge (workgroup-uniform; contains two barriers).
__device__ __forceinline__ bool split_arrive_is_last(int32_t* counter, int live) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through partial stores have completed ...
  __syncthreads();                                  // ... and so have every other thread's, before the count moves
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (old == live - 1);
    if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// _decode_softmax_reducev_fwd (decode_attention.py:731-805) for ONE block of `nheads` (<= 16) consecutive q heads of
// one request, by the 256 threads of the calling workgroup; same arithmetic as decode_merge_kernel (rx_decode.hip).
// logits / lse point at the block's first head: partial (q, s) is logits[(q * max_splits + s) * dv ...], lse[q *
// max_splits + s]; o at the block's first head of the request.
// R: output chunks (4 columns of one head) a thread has in flight per device-scope round trip.  The merging workgroup is
// alone on the critical path of its request (and the launch's tail, when it is the last one): with one chunk per
// round trip the 16 x 512 MLA block took 8 dependent ~2 us trips (round 2's 67.6 -> 69.5 us); R = 4 makes it two.
// The loads are buffer loads with the sc0 sc1 policy bits (aux 17 = device scope, as load_dev_chunk8's) issued through
// the builtin, so hipcc counts them itself and ANY number can be in flight (an asm statement stops at 30 operands).
template <typename T, int R = 1>
__device__ __forceinline__ void merge_splits_in_kernel(const float* logits, const float* lse, int nheads, int dv, int live,
                                                       int max_splits, const float* sinks /* block's first head, or NULL */,
                                                       float v_scale, uint16_t* o, int64_t o_stride_h) {
  // requires max_splits % 8 == 0 and 16-byte aligned buffers (the host enables the in-kernel form only then)
  const int dv4 = dv >> 2;
  const int items = nheads * dv4;
  if (live <= 8) {
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(logits), 0, nheads * max_splits * dv * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lse), 0, nheads * max_splits * 4, 0x00020000);
    for (int i0 = threadIdx.x; i0 < items; i0 += 256 * R) {
      f32x4 la[R], lb[R], x[R][8];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = min(i0 + 256 * r, items - 1);
        const int q = i / dv4, d = (i % dv4) * 4;
        la[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(re, q * max_splits * 4, 0, 17));
        lb[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(re, q * max_splits * 4 + 16, 0, 17));
#pragma unroll
        for (int j = 0; j < 8; ++j)
          x[r][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, ((q * max_splits + min(j, live - 1)) * dv + d) * 4, 0, 17));
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = i0 + 256 * r;
        if (i >= items) break;
        const int q = i / dv4, d = (i % dv4) * 4;
        float e_max = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < live) e_max = fmaxf(e_max, j < 4 ? la[r][j] : lb[r][j - 4]);
        float e_sum = 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < live) {  // in stage 2's order
            const float w = __expf((j < 4 ? la[r][j] : lb[r][j - 4]) - e_max);
            acc += w * x[r][j];
            e_sum += w;
          }
        }
        if (sinks) e_sum += __expf(sinks[q] - e_max);
        const float inv = v_scale / e_sum;
        u32x2 pk;
        pk[0] = pack2<T>(acc[0] * inv, acc[1] * inv);
        pk[1] = pack2<T>(acc[2] * inv, acc[3] * inv);
        *reinterpret_cast<u32x2*>(o + q * o_stride_h + d) = pk;
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < items; i += 256) {  // more than 8 live splits: two passes, as stage 2 (the maximum over ALL splits first)
    const int q = i / dv4, d = (i % dv4) * 4;
    const float* l = lse + q * max_splits;
    const float* lp = logits + static_cast<int64_t>(q) * max_splits * dv + d;
    float e_max = -INFINITY;
    for (int s0 = 0; s0 < live; s0 += 8) {
      f32x4 la, lb;
      load_dev_lse8(l + s0, la, lb);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s0 + j < live) e_max = fmaxf(e_max, j < 4 ? la[j] : lb[j - 4]);
    }
    float e_sum = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < live; s0 += 8) {
      const float* row[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) row[j] = lp + static_cast<int64_t>(min(s0 + j, live - 1)) * dv;
      f32x4 la, lb, x[8];
      load_dev_chunk8(l + s0, row, la, lb, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (s0 + j < live) {  // in stage 2's order
          const float w = __expf((j < 4 ? la[j] : lb[j - 4]) - e_max);
          acc += w * x[j];
          e_sum += w;
        }
      }
    }
    if (sinks) e_sum += __expf(sinks[q] - e_max);
    const float inv = v_scale / e_sum;
    u32x2 pk;
    pk[0] = pack2<T>(acc[0] * inv, acc[1] * inv);
    pk[1] = pack2<T>(acc[2] * inv, acc[3] * inv);
    *reinterpret_cast<u32x2*>(o + q * o_stride_h + d) = pk;
  }
}

// max over the four lanes {l, l^16, l^32, l^48} that hold one query's scores in the S^T accumulator
// layout.  gfx950 half/row swaps (v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute
// round trips through the LDS crossbar: no lgkmcnt wait on the softmax critical path.  Inline asm
// because hipcc (ROCm 7.2) folds fmax(r[0], r[1]) of the builtin's result pair away; the s_nop's are
// the VALU-write -> permlane-read wait states (cdna_hip_programming.md T21).
__device__ __forceinline__ float quad_row_max(float x) {
  float a = x, b = x;
  asm volatile(
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1\n\t"
      "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1"
      : "+v"(a), "+v"(b));
  return a;
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// relative-position score bias (relative_bias_score_mod, kernels/ops/attention/score_mod.py:44-56): element r of a
// (query token, head) row, fp32 or the call's 16-bit dtype
// ... and four consecutive elements r0 .. r0 + 3 in ONE load.  The address is only element-aligned (r0 is a distance of
// positions): gfx950 takes under-aligned global loads, and hipcc emits global_load_dwordx2 / dwordx4 for these types.
typedef u32x2 __attribute__((aligned(2))) u32x2_a2;
typedef f32x4 __attribute__((aligned(4))) f32x4_a4;
template <typename T>
__device__ __forceinline__ void load_bias4(const void* row, int32_t is_f32, int32_t r0, float (&out)[4]) {
  if (is_f32) {
    const f32x4_a4 v = *reinterpret_cast<const f32x4_a4*>(static_cast<const float*>(row) + r0);
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
  } else {
    const u32x2_a2 v = *reinterpret_cast<const u32x2_a2*>(static_cast<const uint16_t*>(row) + r0);
    out[0] = T::to_f32(static_cast<uint16_t>(v[0] & 0xFFFFu)); out[1] = T::to_f32(static_cast<uint16_t>(v[0] >> 16));
    out[2] = T::to_f32(static_cast<uint16_t>(v[1] & 0xFFFFu)); out[3] = T::to_f32(static_cast<uint16_t>(v[1] >> 16));
  }
}
template <typename T>
__device__ __forceinline__ float load_bias(const void* row, int32_t is_f32, int32_t r) {
  return is_f32 ? static_cast<const float*>(row)[r] : T::to_f32(static_cast<const uint16_t*>(row)[r]);
}

}  // namespace rx
