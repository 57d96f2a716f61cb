// C1: sum all-reduce of the row-parallel o_proj output over one node's GPUs, written for the xGMI
// full mesh (every GPU has a direct link to every peer) and for decode-sized messages (2 MiB at
// bs 256 x hidden 4096 x bf16).
//
// Reference: RowParallelLinear.forward (srt/layers/linear.py:1606-1627) ->
// tensor_model_parallel_all_reduce (srt/distributed/communication_op.py:18-20) ->
// GroupCoordinator.all_reduce (srt/distributed/parallel_state.py:622-732), which dispatches to a
// custom peer-to-peer all-reduce for small messages and to NCCL otherwise.  This file is the custom
// one; sglang_amd/parallel.py keeps RCCL as the default and fallback.
//
// Why not a ring: xGMI is point to point, a ring uses ONE of the seven links per step and needs
// 2 (W-1) latency-bound steps.  Here every rank talks to all peers at once (two-shot direct):
//   phase 0  copy the input into this rank's IPC-shared staging buffer; tell every peer
//   phase 1  reduce-scatter: rank r sums chunk r of all W staging buffers (fp32) -> its result buffer
//   phase 2  all-gather: every rank reads the W reduced chunks into `out`
// each GPU moves 2 (W-1)/W of the message over W-1 links in parallel, with two flag exchanges.
//
// Synchronisation is per block index, never grid wide: block b of rank r only ever waits for block b
// of its peers (it reads exactly the slices those blocks wrote), so no co-residency is assumed.
// Flags are monotonically increasing call numbers in the receiver's region; staging / result buffers
// alternate by call parity, so a rank may start call g+1 while a slow peer still reads call g's data
// (a buffer is reused at g+2, after the peer's "ready" of g+1 proved it finished g).  Spins are
// bounded: on timeout the kernel raises RX_DEVERR_AR_TIMEOUT in the context's error word and returns.
#include "rx_common.h"

#include <cstring>

namespace rx {

constexpr int kArMaxWorld = 8;
constexpr int kArBlocks = 64;      // blocks per rank (= independent flag lanes)
constexpr int kArThreads = 256;
constexpr uint32_t kArSpinLimit = 1u << 27;

// layout of one rank's shared region:  [flags][stage 0][stage 1][result 0][result 1]
struct ArFlags {
  // ready[b][src]: src's block b has staged its input of call `value`; done[b][src]: ... reduced its chunk
  uint32_t ready[kArBlocks][kArMaxWorld];
  uint32_t done[kArBlocks][kArMaxWorld];
};

struct ArCtx {
  int rank, world;
  int64_t max_bytes;       // per-call message limit (bytes)
  char* peers[kArMaxWorld];  // every rank's region mapped into this process (own = local pointer)
  uint32_t call;             // host-side call counter
  int32_t* dev_err;          // device error word (RX_DEVERR_*)
};

__host__ __device__ inline int64_t ar_align(int64_t x) { return (x + 255) / 256 * 256; }
__host__ __device__ inline int64_t ar_region_bytes(int64_t max_bytes) {
  return ar_align(sizeof(ArFlags)) + 4 * ar_align(max_bytes);
}

struct ArArgs {
  char* peers[kArMaxWorld];
  int rank, world;
  int64_t max_bytes;
  const uint16_t* in;
  uint16_t* out;
  int64_t n;  // elements
  uint32_t call;
  int32_t* dev_err;
};

__device__ __forceinline__ void ar_signal(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool ar_wait(const uint32_t* p, uint32_t v) {
  for (uint32_t i = 0; i < kArSpinLimit; ++i) {
    // calls are numbered 1, 2, 3, ...: "at least v" (a fast peer may already be a call ahead)
    if (static_cast<int32_t>(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - v) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// 8 elements (16 B) per thread step; chunk c = elements [c * per, min((c+1) * per, n)), per % 8 == 0
template <typename T>
__global__ __launch_bounds__(kArThreads) void allreduce_two_shot_kernel(const ArArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, W = a.world, r = a.rank;
  const int64_t stage_off = ar_align(sizeof(ArFlags)) + (a.call & 1) * ar_align(a.max_bytes);
  const int64_t result_off = ar_align(sizeof(ArFlags)) + (2 + (a.call & 1)) * ar_align(a.max_bytes);
  const int64_t nv = a.n / 8;                     // 16-byte vectors
  const int64_t per = (nv + W - 1) / W;           // vectors per chunk
  __shared__ int timeout_s;
  if (tid == 0) timeout_s = 0;

  // ---- phase 0: stage my input (block b takes every kArBlocks-th group of 256 vectors of each chunk)
  u32x4* my_stage = reinterpret_cast<u32x4*>(a.peers[r] + stage_off);
  const u32x4* in_v = reinterpret_cast<const u32x4*>(a.in);
  for (int c = 0; c < W; ++c) {
    const int64_t lo = c * per, hi = min(lo + per, nv);
    for (int64_t i = lo + b * kArThreads + tid; i < hi; i += kArBlocks * kArThreads) my_stage[i] = in_v[i];
  }
  __threadfence_system();
  __syncthreads();
  if (tid < W && tid != r)
    ar_signal(&reinterpret_cast<ArFlags*>(a.peers[tid])->ready[b][r], a.call);
  if (tid < W && tid != r) {
    if (!ar_wait(&reinterpret_cast<ArFlags*>(a.peers[r])->ready[b][tid], a.call)) timeout_s = 1;
  }
  __syncthreads();
  if (timeout_s) {
    if (tid == 0) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    return;
  }

  // ---- phase 1: reduce my chunk over all ranks' staging buffers (fp32 accumulate, rank order fixed
  // so that every rank computes bit-identical sums)
  {
    const int64_t lo = r * per, hi = min(lo + per, nv);
    u32x4* my_res = reinterpret_cast<u32x4*>(a.peers[r] + result_off);
    u32x4* out_v = reinterpret_cast<u32x4*>(a.out);
    for (int64_t i = lo + b * kArThreads + tid; i < hi; i += kArBlocks * kArThreads) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int p = 0; p < W; ++p) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.peers[p] + stage_off) + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += T::to_f32(static_cast<uint16_t>(v[j] & 0xffffu));
          acc[2 * j + 1] += T::to_f32(static_cast<uint16_t>(v[j] >> 16));
        }
      }
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack2<T>(acc[2 * j], acc[2 * j + 1]);
      my_res[i] = o;
      out_v[i] = o;
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < W && tid != r)
    ar_signal(&reinterpret_cast<ArFlags*>(a.peers[tid])->done[b][r], a.call);
  if (tid < W && tid != r) {
    if (!ar_wait(&reinterpret_cast<ArFlags*>(a.peers[r])->done[b][tid], a.call)) timeout_s = 1;
  }
  __syncthreads();
  if (timeout_s) {
    if (tid == 0) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    return;
  }

  // ---- phase 2: gather the other ranks' reduced chunks
  {
    u32x4* out_v = reinterpret_cast<u32x4*>(a.out);
    for (int c = 0; c < W; ++c) {
      if (c == r) continue;
      const int64_t lo = c * per, hi = min(lo + per, nv);
      const u32x4* res = reinterpret_cast<const u32x4*>(a.peers[c] + result_off);
      for (int64_t i = lo + b * kArThreads + tid; i < hi; i += kArBlocks * kArThreads)
        out_v[i] = __builtin_nontemporal_load(res + i);
    }
  }
}

}  // namespace rx

using namespace rx;

extern "C" {

int64_t rx_ar_region_bytes(int64_t max_bytes) { return max_bytes > 0 ? ar_region_bytes(max_bytes) : -1; }

int rx_ar_alloc_region(int64_t bytes, void** dev_ptr_out) {
  RX_REQUIRE(bytes > 0 && dev_ptr_out, "rx_ar_alloc_region: bad arguments");
  // uncached: peers poll the flags and read the data while this GPU's kernel is still running, so
  // nothing of the region may sit in a non-coherent L2 line
  hipError_t e = hipExtMallocWithFlags(dev_ptr_out, static_cast<size_t>(bytes), hipDeviceMallocUncached);
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipExtMallocWithFlags(uncached, %lld): %s", (long long)bytes,
                                   hipGetErrorString(e));
  e = hipMemset(*dev_ptr_out, 0, static_cast<size_t>(bytes));
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipMemset: %s", hipGetErrorString(e));
  return RX_OK;
}

int rx_ar_free_region(void* dev_ptr) {
  hipError_t e = hipFree(dev_ptr);
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipFree: %s", hipGetErrorString(e));
  return RX_OK;
}

int rx_ipc_get_handle(void* dev_ptr, void* handle_out_64b) {
  RX_REQUIRE(dev_ptr && handle_out_64b, "rx_ipc_get_handle: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out_64b), dev_ptr);
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
  return RX_OK;
}

int rx_ipc_open_handle(const void* handle_64b, void** dev_ptr_out) {
  RX_REQUIRE(handle_64b && dev_ptr_out, "rx_ipc_open_handle: null pointer");
  hipIpcMemHandle_t h;
  memcpy(&h, handle_64b, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(dev_ptr_out, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  return RX_OK;
}

int rx_ipc_close_handle(void* dev_ptr) {
  hipError_t e = hipIpcCloseMemHandle(dev_ptr);
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
  return RX_OK;
}

int rx_ar_init(rx_ar_ctx** ctx_out, int rank, int world, void* const* peer_regions, int64_t max_bytes,
               int32_t* dev_err) {
  RX_REQUIRE(ctx_out && peer_regions && dev_err, "rx_ar_init: null pointer");
  RX_REQUIRE(world >= 2 && world <= kArMaxWorld && rank >= 0 && rank < world,
             "rx_ar_init: rank %d / world %d (2..%d ranks)", rank, world, kArMaxWorld);
  RX_REQUIRE(max_bytes > 0 && max_bytes % 256 == 0, "rx_ar_init: max_bytes must be a positive multiple of 256");
  auto* c = new ArCtx();
  c->rank = rank;
  c->world = world;
  c->max_bytes = max_bytes;
  c->call = 0;
  c->dev_err = dev_err;
  for (int i = 0; i < world; ++i) {
    if (!peer_regions[i]) {
      delete c;
      return fail(RX_ERR_INVALID_ARG, "rx_ar_init: peer region %d is null", i);
    }
    c->peers[i] = static_cast<char*>(peer_regions[i]);
  }
  *ctx_out = reinterpret_cast<rx_ar_ctx*>(c);
  return RX_OK;
}

int rx_allreduce(rx_ar_ctx* ctx, const void* in, void* out, int64_t count, int dtype, void* stream) {
  RX_REQUIRE(ctx && in && out, "rx_allreduce: null pointer");
  auto* c = reinterpret_cast<ArCtx*>(ctx);
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_allreduce: dtype %d", dtype);
  RX_REQUIRE(count >= 0 && count % 8 == 0, "rx_allreduce: count %lld must be a multiple of 8", (long long)count);
  RX_REQUIRE(count * 2 <= c->max_bytes, "rx_allreduce: %lld bytes exceed the context's %lld", (long long)count * 2,
             (long long)c->max_bytes);
  RX_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "rx_allreduce: in/out must be 16-byte aligned");
  if (count == 0) return RX_OK;
  ArArgs a;
  for (int i = 0; i < c->world; ++i) a.peers[i] = c->peers[i];
  a.rank = c->rank;
  a.world = c->world;
  a.max_bytes = c->max_bytes;
  a.in = static_cast<const uint16_t*>(in);
  a.out = static_cast<uint16_t*>(out);
  a.n = count;
  a.call = ++c->call;  // every rank issues the same sequence of calls, so the counters agree
  a.dev_err = c->dev_err;
  auto s = static_cast<hipStream_t>(stream);
  if (dtype == RX_BF16)
    hipLaunchKernelGGL(allreduce_two_shot_kernel<BF16>, dim3(kArBlocks), dim3(kArThreads), 0, s, a);
  else
    hipLaunchKernelGGL(allreduce_two_shot_kernel<F16>, dim3(kArBlocks), dim3(kArThreads), 0, s, a);
  return check_launch("rx_allreduce");
}

int rx_ar_destroy(rx_ar_ctx* ctx) {
  delete reinterpret_cast<ArCtx*>(ctx);
  return RX_OK;
}

}  // extern "C"
