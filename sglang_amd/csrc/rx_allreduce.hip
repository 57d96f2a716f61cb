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
//
// HIP-graph safe: the call number lives in DEVICE memory (one counter per block index in the rank's own
// region, read and bumped by that block), not in the launch arguments -- a captured launch replays with the
// same arguments and still sees call numbers 1, 2, 3, ... (the reference's custom all-reduce keeps its signal
// counters on the device for the same reason, parallel_state.py:560-620).  One context serves ONE stream of
// ordered launches; a caller that reduces on two streams (sglang_amd/parallel.py: main + side stream) uses one
// context per stream.
//
// allreduce_rmsnorm_kernel (GroupCoordinator.fused_allreduce_rmsnorm, parallel_state.py:748-878):
//   residual_out = all_reduce(x) + residual ;  out = rmsnorm(residual_out) * weight
// in the same two-shot schedule, chunked by ROWS so that a rank owns whole rows: phase 1 reduces its rows,
// adds the residual and normalises them; phase 2 gathers the other ranks' residual_out rows and normalises
// them locally while they pass through registers -- the normalised rows never cross xGMI (a separate
// RMSNorm launch would re-read 2 x T x H bytes from HBM; shipping `out` as well would double the gather).
#include "rx_common.h"

#include <cstring>

namespace rx {

constexpr int kArMaxWorld = 8;
constexpr int kArBlocks = 256;     // most blocks per rank (= independent flag lanes); a launch uses ar_grid() of them
constexpr int kArThreads = 256;
constexpr uint32_t kArSpinLimit = 1u << 27;

// layout of one rank's shared region:  [flags][stage 0][stage 1][result 0][result 1]
struct ArFlags {
  // ready[b][src]: src's block b has staged its input of call `value`; done[b][src]: ... reduced its chunk
  uint32_t ready[kArBlocks][kArMaxWorld];
  uint32_t done[kArBlocks][kArMaxWorld];
  uint32_t calls[kArBlocks];  // this rank's own call counter, one per block index (written by that block only)
};

struct ArCtx {
  int rank, world;
  int64_t max_bytes;       // per-call message limit (bytes)
  char* peers[kArMaxWorld];  // every rank's region mapped into this process (own = local pointer)
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
  int32_t* dev_err;
  // fused residual-add + RMSNorm form: x = in [rows, hidden]; out = normalised, out_res = residual stream
  const uint16_t* residual;
  const uint16_t* weight;
  uint16_t* out_res;
  int32_t rows, hidden;
  float eps;
  int32_t fenced;  // option ar_fenced
  uint32_t spin_limit;  // polls before a wait gives up (option ar_spin_log2; default 2^27)
  int32_t active;       // blocks that take part in this call (ar_grid); the others only count it
};

// The flag handshake (round 6; measured on rx_quick_allreduce.hip, which documents the numbers): a block's stores into the
// shared regions are ACKNOWLEDGED before its flag goes out -- every wave waits vmcnt(0) (stores count on gfx950; the regions
// are uncached memory, an acknowledged store is out of this GPU's caches), workgroup barrier, relaxed system-scope flag
// store -- and consumed by relaxed polling, barrier, non-temporal loads of uncached memory.  Option ar_fenced = 1: the
// memory-model form (system fence in every wave, release store, acquire fence behind the wait), what this file did until
// round 6: the whole-L2 write-back / invalidate behind those fences were most of a small message's latency (two processes
// on one GPU, 2 MiB: 42 us -> see DESIGN 7).
__device__ __forceinline__ void ar_publish_begin(bool fenced) {
  if (fenced) __threadfence_system();
  else __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
}
__device__ __forceinline__ void ar_signal(uint32_t* p, uint32_t v, bool fenced) {
  if (fenced) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool ar_wait(const uint32_t* p, uint32_t v, uint32_t limit) {
  for (uint32_t i = 0; i < limit; ++i) {
    // calls are numbered 1, 2, 3, ...: "at least v" (a fast peer may already be a call ahead)
    if (static_cast<int32_t>(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - v) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
__device__ __forceinline__ void ar_acquire(bool fenced) {
  if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// call number of this launch for block b: previous + 1, kept in the rank's own region
__device__ __forceinline__ uint32_t ar_next_call(const ArArgs& a, int b, uint32_t* sh) {
  if (threadIdx.x == 0) {
    uint32_t* c = &reinterpret_cast<ArFlags*>(a.peers[a.rank])->calls[b];
    const uint32_t g = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __hip_atomic_store(c, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *sh = g;
  }
  __syncthreads();
  return *sh;
}

// 8 elements (16 B) per thread step; chunk c = elements [c * per, min((c+1) * per, n)), per % 8 == 0
// ONE_SHOT (rx_allreduce_det; the reference's deterministic all-reduce on AMD forces its one-stage kernel,
// device_communicators/custom_all_reduce.py:294-301, kernels/aot/csrc/allreduce/deterministic_all_reduce.hip:1-14): after
// the staging exchange EVERY rank reduces ALL chunks itself -- fp32, rank order 0 .. W-1, one rounding -- straight into
// `out`; no second flag exchange, no gather.  (The two-shot form sums in the same fixed order; what the one-shot form adds
// is the reference's structure: no rank's result depends on another rank's reduction.)
// WT: the world size as a compile-time constant (2, 4, 6, 8): the W loads of one reduction step are then issued together
// instead of one per trip of a run-time loop (uncached reads are ~2 us round trips; round 6, two processes on one GPU, 2 MiB:
// one-shot 44 -> see DESIGN 7).
template <typename T, bool ONE_SHOT, int WT>
__global__ __launch_bounds__(kArThreads) void allreduce_two_shot_kernel(const ArArgs a) {
  constexpr int W = WT;
  const int b = blockIdx.x, tid = threadIdx.x, r = a.rank;
  __shared__ uint32_t call_s;
  __shared__ int timeout_s;
  if (tid == 0) timeout_s = 0;
  const uint32_t call = ar_next_call(a, b, &call_s);
  // Every launch has kArBlocks blocks so that ALL per-block call counters advance together (the staging / result buffers
  // alternate by the call's parity, which must be one value per launch); only the first `active` of them -- the same number
  // on every rank, a function of the message size -- move data and exchange flags.
  const int nb = a.active;
  if (b >= nb) return;
  const int64_t stage_off = ar_align(sizeof(ArFlags)) + (call & 1) * ar_align(a.max_bytes);
  const int64_t result_off = ar_align(sizeof(ArFlags)) + (2 + (call & 1)) * ar_align(a.max_bytes);
  const int64_t nv = a.n / 8;                     // 16-byte vectors
  const int64_t per = (nv + W - 1) / W;           // vectors per chunk

  // ---- phase 0: stage my input (block b takes every nb-th group of 256 vectors of each chunk)
  u32x4* my_stage = reinterpret_cast<u32x4*>(a.peers[r] + stage_off);
  const u32x4* in_v = reinterpret_cast<const u32x4*>(a.in);
  for (int c = 0; c < W; ++c) {
    const int64_t lo = c * per, hi = min(lo + per, nv);
#pragma unroll 4
    for (int64_t i = lo + b * kArThreads + tid; i < hi; i += nb * kArThreads) my_stage[i] = in_v[i];
  }
  const bool fenced = a.fenced != 0;
  ar_publish_begin(fenced);
  __syncthreads();
  if (tid < W && tid != r)
    ar_signal(&reinterpret_cast<ArFlags*>(a.peers[tid])->ready[b][r], call, fenced);
  if (tid < W && tid != r) {
    if (!ar_wait(&reinterpret_cast<ArFlags*>(a.peers[r])->ready[b][tid], call, a.spin_limit)) timeout_s = 1;
  }
  __syncthreads();
  if (timeout_s) {
    if (tid == 0) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    return;
  }
  ar_acquire(fenced);

  // ---- phase 1: reduce my chunk over all ranks' staging buffers (fp32 accumulate, rank order fixed
  // so that every rank computes bit-identical sums)
  for (int c = ONE_SHOT ? 0 : r; c < (ONE_SHOT ? W : r + 1); ++c) {  // (one-shot: every chunk, by the block that staged its slices)
    const int64_t lo = c * per, hi = min(lo + per, nv);
    u32x4* my_res = reinterpret_cast<u32x4*>(a.peers[r] + result_off);
    u32x4* out_v = reinterpret_cast<u32x4*>(a.out);
#pragma unroll 2
    for (int64_t i = lo + b * kArThreads + tid; i < hi; i += nb * kArThreads) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u32x4 vs[W];
#pragma unroll
      for (int p = 0; p < W; ++p) vs[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.peers[p] + stage_off) + i);
#pragma unroll
      for (int p = 0; p < W; ++p) {   // (rank order 0 .. W-1: the sum's order is part of the contract)
        const u32x4 v = vs[p];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += T::to_f32(static_cast<uint16_t>(v[j] & 0xffffu));
          acc[2 * j + 1] += T::to_f32(static_cast<uint16_t>(v[j] >> 16));
        }
      }
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack2<T>(acc[2 * j], acc[2 * j + 1]);
      if constexpr (!ONE_SHOT) my_res[i] = o;
      out_v[i] = o;
    }
  }
  if constexpr (ONE_SHOT) return;  // (the staging buffer of this parity is reused at call + 2, behind the peers' ready(call + 1))
  ar_publish_begin(fenced);
  __syncthreads();
  if (tid < W && tid != r)
    ar_signal(&reinterpret_cast<ArFlags*>(a.peers[tid])->done[b][r], call, fenced);
  if (tid < W && tid != r) {
    if (!ar_wait(&reinterpret_cast<ArFlags*>(a.peers[r])->done[b][tid], call, a.spin_limit)) timeout_s = 1;
  }
  __syncthreads();
  if (timeout_s) {
    if (tid == 0) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    return;
  }
  ar_acquire(fenced);

  // ---- phase 2: gather the other ranks' reduced chunks
  if constexpr (!ONE_SHOT) {
    u32x4* out_v = reinterpret_cast<u32x4*>(a.out);
    for (int c = 0; c < W; ++c) {
      if (c == r) continue;
      const int64_t lo = c * per, hi = min(lo + per, nv);
      const u32x4* res = reinterpret_cast<const u32x4*>(a.peers[c] + result_off);
#pragma unroll 4
      for (int64_t i = lo + b * kArThreads + tid; i < hi; i += nb * kArThreads)
        out_v[i] = __builtin_nontemporal_load(res + i);
    }
  }
}

// ---- fused all-reduce + residual add + RMSNorm ----------------------------------------------------------------
// Rows [c * per, (c + 1) * per) are rank c's chunk; row j of a chunk (local index l) belongs to block l % nb
// in every phase on every rank, so block b still only depends on block b of its peers.  A row is H / 8 16-byte
// vectors, thread t takes vectors t, t + 256, ... (at most kArVpt of them: H <= 16384).
constexpr int kArVpt = 8;

__device__ __forceinline__ float ar_block_sum(float x, float* sh) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  __syncthreads();  // sh may still be read by the previous row
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// one residual_out row held in registers as rounded 16-bit values `ro`: out[row] = ro * rstd * weight
template <typename T>
__device__ __forceinline__ void ar_norm_row(const ArArgs& a, const u32x4 (&ro)[kArVpt], int nvec, int64_t row,
                                            float* sh) {
  const int tid = threadIdx.x;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < kArVpt; ++k) {
    const int v = tid + k * kArThreads;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = T::to_f32(static_cast<uint16_t>(ro[k][j] & 0xffffu));
        const float hi = T::to_f32(static_cast<uint16_t>(ro[k][j] >> 16));
        ss += lo * lo + hi * hi;
      }
    }
  }
  const float rstd = rsqrtf(ar_block_sum(ss, sh) / static_cast<float>(a.hidden) + a.eps);
  u32x4* out_v = reinterpret_cast<u32x4*>(a.out) + row * nvec;
  const u32x4* w_v = reinterpret_cast<const u32x4*>(a.weight);
#pragma unroll
  for (int k = 0; k < kArVpt; ++k) {
    const int v = tid + k * kArThreads;
    if (v < nvec) {
      const u32x4 w = w_v[v];
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = T::to_f32(static_cast<uint16_t>(ro[k][j] & 0xffffu)) * rstd *
                         T::to_f32(static_cast<uint16_t>(w[j] & 0xffffu));
        const float hi = T::to_f32(static_cast<uint16_t>(ro[k][j] >> 16)) * rstd *
                         T::to_f32(static_cast<uint16_t>(w[j] >> 16));
        o[j] = pack2<T>(lo, hi);
      }
      out_v[v] = o;
    }
  }
}

template <typename T, int WT>
__global__ __launch_bounds__(kArThreads) void allreduce_rmsnorm_kernel(const ArArgs a) {
  constexpr int W = WT;
  const int b = blockIdx.x, tid = threadIdx.x, r = a.rank;
  __shared__ uint32_t call_s;
  __shared__ int timeout_s;
  __shared__ float red_s[4];
  if (tid == 0) timeout_s = 0;
  const uint32_t call = ar_next_call(a, b, &call_s);
  const int nb = a.active;   // (see allreduce_two_shot_kernel)
  if (b >= nb) return;
  const int64_t stage_off = ar_align(sizeof(ArFlags)) + (call & 1) * ar_align(a.max_bytes);
  const int64_t result_off = ar_align(sizeof(ArFlags)) + (2 + (call & 1)) * ar_align(a.max_bytes);
  const int nvec = a.hidden / 8;
  const int64_t per = (a.rows + W - 1) / W;  // rows per chunk

  // ---- phase 0: stage my rows
  {
    u32x4* my_stage = reinterpret_cast<u32x4*>(a.peers[r] + stage_off);
    const u32x4* in_v = reinterpret_cast<const u32x4*>(a.in);
    for (int c = 0; c < W; ++c) {
      const int64_t lo = c * per, hi = min(lo + per, static_cast<int64_t>(a.rows));
      for (int64_t row = lo + b; row < hi; row += nb)
        for (int v = tid; v < nvec; v += kArThreads) my_stage[row * nvec + v] = in_v[row * nvec + v];
    }
  }
  const bool fenced = a.fenced != 0;
  ar_publish_begin(fenced);
  __syncthreads();
  if (tid < W && tid != r) ar_signal(&reinterpret_cast<ArFlags*>(a.peers[tid])->ready[b][r], call, fenced);
  if (tid < W && tid != r) {
    if (!ar_wait(&reinterpret_cast<ArFlags*>(a.peers[r])->ready[b][tid], call, a.spin_limit)) timeout_s = 1;
  }
  __syncthreads();
  if (timeout_s) {
    if (tid == 0) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    return;
  }
  ar_acquire(fenced);

  // ---- phase 1: my rows = sum over ranks (fp32, fixed order) -> 16-bit, + residual -> 16-bit, normalise
  {
    const int64_t lo = r * per, hi = min(lo + per, static_cast<int64_t>(a.rows));
    u32x4* my_res = reinterpret_cast<u32x4*>(a.peers[r] + result_off);
    u32x4* res_out = reinterpret_cast<u32x4*>(a.out_res);
    const u32x4* resid = reinterpret_cast<const u32x4*>(a.residual);
    for (int64_t row = lo + b; row < hi; row += nb) {
      u32x4 ro[kArVpt];
#pragma unroll
      for (int k = 0; k < kArVpt; ++k) {
        const int v = tid + k * kArThreads;
        if (v < nvec) {
          float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          u32x4 xs[W];
#pragma unroll
          for (int p = 0; p < W; ++p)
            xs[p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.peers[p] + stage_off) + row * nvec + v);
#pragma unroll
          for (int p = 0; p < W; ++p) {   // (rank order: the sum's order is part of the contract)
            const u32x4 x = xs[p];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[2 * j] += T::to_f32(static_cast<uint16_t>(x[j] & 0xffffu));
              acc[2 * j + 1] += T::to_f32(static_cast<uint16_t>(x[j] >> 16));
            }
          }
          const u32x4 rs = resid[row * nvec + v];
          u32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // the split path's two roundings: all_reduce output in 16 bits, then the residual add in 16 bits
            const uint32_t s2 = pack2<T>(acc[2 * j], acc[2 * j + 1]);
            o[j] = pack2<T>(
                T::to_f32(static_cast<uint16_t>(s2 & 0xffffu)) + T::to_f32(static_cast<uint16_t>(rs[j] & 0xffffu)),
                T::to_f32(static_cast<uint16_t>(s2 >> 16)) + T::to_f32(static_cast<uint16_t>(rs[j] >> 16)));
          }
          ro[k] = o;
          my_res[row * nvec + v] = o;
          res_out[row * nvec + v] = o;
        }
      }
      ar_norm_row<T>(a, ro, nvec, row, red_s);
    }
  }
  ar_publish_begin(fenced);
  __syncthreads();
  if (tid < W && tid != r) ar_signal(&reinterpret_cast<ArFlags*>(a.peers[tid])->done[b][r], call, fenced);
  if (tid < W && tid != r) {
    if (!ar_wait(&reinterpret_cast<ArFlags*>(a.peers[r])->done[b][tid], call, a.spin_limit)) timeout_s = 1;
  }
  __syncthreads();
  if (timeout_s) {
    if (tid == 0) atomicOr(a.dev_err, RX_DEVERR_AR_TIMEOUT);
    return;
  }
  ar_acquire(fenced);

  // ---- phase 2: the other ranks' residual_out rows: copy and normalise on the way through
  {
    u32x4* res_out = reinterpret_cast<u32x4*>(a.out_res);
    for (int c = 0; c < W; ++c) {
      if (c == r) continue;
      const int64_t lo = c * per, hi = min(lo + per, static_cast<int64_t>(a.rows));
      const u32x4* src = reinterpret_cast<const u32x4*>(a.peers[c] + result_off);
      for (int64_t row = lo + b; row < hi; row += nb) {
        u32x4 ro[kArVpt];
#pragma unroll
        for (int k = 0; k < kArVpt; ++k) {
          const int v = tid + k * kArThreads;
          if (v < nvec) {
            ro[k] = __builtin_nontemporal_load(src + row * nvec + v);
            res_out[row * nvec + v] = ro[k];
          }
        }
        ar_norm_row<T>(a, ro, nvec, row, red_s);
      }
    }
  }
}

uint32_t ar_spin_limit() {  // (rx_quick_allreduce.hip uses it too)
  const int lg = options().ar_spin_log2;
  return lg >= 10 && lg <= 31 ? 1u << lg : kArSpinLimit;
}

// Blocks of a launch, the same on every rank (a function of the element count and a process-wide option only): block b
// only ever talks to block b of its peers, so any grid up to kArBlocks works.  Option ar_blocks forces a count.
// Round 6, two processes on one GPU, us per call at 32 / 64 / 128 / 256 blocks: two-shot 256 KiB 6.9 / 7.0 / 7.1 / 7.2, 2 MiB
// 19.6 / 13.2 / 10.0 / 8.6, 8 MiB 55 / 32 / 21 / 16; fused RMSNorm 2 MiB 31.6 / 19.1 / 13.4 / 13.2 -- one pass of 256 threads
// over a rank's chunk per block, between 32 and kArBlocks blocks.
static int ar_grid(int64_t vectors, int world) {
  const int forced = options().ar_blocks;
  if (forced > 0) return forced < kArBlocks ? forced : kArBlocks;
  const int64_t per = (vectors + world - 1) / world;                  // vectors of one rank's chunk
  const int64_t nb = (per + kArThreads - 1) / kArThreads;
  return static_cast<int>(nb < 32 ? 32 : (nb > kArBlocks ? kArBlocks : nb));
}

template <bool ONE_SHOT>
static void launch_two_shot(const ArArgs& a, int dtype, hipStream_t s) {
#define RX_AR_W(TT, WW) hipLaunchKernelGGL((allreduce_two_shot_kernel<TT, ONE_SHOT, WW>), dim3(kArBlocks), dim3(kArThreads), 0, s, a)
#define RX_AR_T(TT)                                                  \
  do {                                                               \
    switch (a.world) {                                               \
      case 2: RX_AR_W(TT, 2); break;                                 \
      case 3: RX_AR_W(TT, 3); break;                                 \
      case 4: RX_AR_W(TT, 4); break;                                 \
      case 5: RX_AR_W(TT, 5); break;                                 \
      case 6: RX_AR_W(TT, 6); break;                                 \
      case 7: RX_AR_W(TT, 7); break;                                 \
      default: RX_AR_W(TT, 8); break;                                \
    }                                                                \
  } while (0)
  if (dtype == RX_BF16) RX_AR_T(BF16);
  else RX_AR_T(F16);
#undef RX_AR_T
#undef RX_AR_W
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
  // The zeroing must have HAPPENED before the handle leaves this process: hipMemset of device memory is asynchronous, and a
  // peer that maps the region may raise a flag in it before this GPU queue has run the fill -- which then wipes the flag
  // and the owner waits for it until the timeout (seen with four processes time-slicing one GPU: the owner's queue got its
  // turn after a peer's first kernel; round 6).
  e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(RX_ERR_LAUNCH, "hipDeviceSynchronize: %s", hipGetErrorString(e));
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

static ArArgs ar_args(const ArCtx* c) {
  ArArgs a{};
  for (int i = 0; i < c->world; ++i) a.peers[i] = c->peers[i];
  a.rank = c->rank;
  a.world = c->world;
  a.max_bytes = c->max_bytes;
  a.dev_err = c->dev_err;
  a.fenced = options().ar_fenced;
  a.spin_limit = ar_spin_limit();
  return a;
}

int rx_allreduce(rx_ar_ctx* ctx, const void* in, void* out, int64_t count, int dtype, void* stream) {
  RX_RANGE("rx_allreduce");
  RX_REQUIRE(ctx && in && out, "rx_allreduce: null pointer");
  auto* c = reinterpret_cast<ArCtx*>(ctx);
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_allreduce: dtype %d", dtype);
  RX_REQUIRE(count >= 0 && count % 8 == 0, "rx_allreduce: count %lld must be a multiple of 8", (long long)count);
  RX_REQUIRE(count * 2 <= c->max_bytes, "rx_allreduce: %lld bytes exceed the context's %lld", (long long)count * 2,
             (long long)c->max_bytes);
  RX_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "rx_allreduce: in/out must be 16-byte aligned");
  if (count == 0) return RX_OK;
  ArArgs a = ar_args(c);
  a.in = static_cast<const uint16_t*>(in);
  a.out = static_cast<uint16_t*>(out);
  a.n = count;
  auto s = static_cast<hipStream_t>(stream);
  a.active = ar_grid(a.n / 8, a.world);
  launch_two_shot<false>(a, dtype, s);
  return check_launch("rx_allreduce");
}

int rx_allreduce_det(rx_ar_ctx* ctx, const void* in, void* out, int64_t count, int dtype, void* stream) {
  RX_RANGE("rx_allreduce_det");
  RX_REQUIRE(ctx && in && out, "rx_allreduce_det: null pointer");
  auto* c = reinterpret_cast<ArCtx*>(ctx);
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_allreduce_det: dtype %d", dtype);
  RX_REQUIRE(count >= 0 && count % 8 == 0, "rx_allreduce_det: count %lld must be a multiple of 8", (long long)count);
  RX_REQUIRE(count * 2 <= c->max_bytes, "rx_allreduce_det: %lld bytes exceed the context's %lld", (long long)count * 2,
             (long long)c->max_bytes);
  RX_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "rx_allreduce_det: in/out must be 16-byte aligned");
  if (count == 0) return RX_OK;
  ArArgs a = ar_args(c);
  a.in = static_cast<const uint16_t*>(in);
  a.out = static_cast<uint16_t*>(out);
  a.n = count;
  auto s = static_cast<hipStream_t>(stream);
  a.active = ar_grid(a.n / 8, a.world);
  launch_two_shot<true>(a, dtype, s);
  return check_launch("rx_allreduce_det");
}

int rx_allreduce_rmsnorm(rx_ar_ctx* ctx, const void* in, const void* residual_in, const void* weight, void* out,
                         void* residual_out, int64_t rows, int64_t hidden, float eps, int dtype, void* stream) {
  RX_RANGE("rx_allreduce_rmsnorm");
  RX_REQUIRE(ctx && in && residual_in && weight && out && residual_out, "rx_allreduce_rmsnorm: null pointer");
  auto* c = reinterpret_cast<ArCtx*>(ctx);
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_allreduce_rmsnorm: dtype %d", dtype);
  RX_REQUIRE(rows >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 8 * kArThreads * kArVpt,
             "rx_allreduce_rmsnorm: hidden %lld must be a multiple of 8, at most %d", (long long)hidden,
             8 * kArThreads * kArVpt);
  RX_REQUIRE(rows * hidden * 2 <= c->max_bytes, "rx_allreduce_rmsnorm: %lld bytes exceed the context's %lld",
             (long long)(rows * hidden * 2), (long long)c->max_bytes);
  RX_REQUIRE((((uintptr_t)in | (uintptr_t)residual_in | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)residual_out) & 15) == 0,
             "rx_allreduce_rmsnorm: pointers must be 16-byte aligned");
  RX_REQUIRE(out != in && out != residual_in && out != residual_out, "rx_allreduce_rmsnorm: out must not alias an input");
  if (rows == 0) return RX_OK;
  ArArgs a = ar_args(c);
  a.in = static_cast<const uint16_t*>(in);
  a.residual = static_cast<const uint16_t*>(residual_in);
  a.weight = static_cast<const uint16_t*>(weight);
  a.out = static_cast<uint16_t*>(out);
  a.out_res = static_cast<uint16_t*>(residual_out);
  a.rows = static_cast<int32_t>(rows);
  a.hidden = static_cast<int32_t>(hidden);
  a.eps = eps;
  a.n = rows * hidden;
  auto s = static_cast<hipStream_t>(stream);
  a.active = ar_grid(a.n / 8, a.world);
#define RX_ARN_W(TT, WW) hipLaunchKernelGGL((allreduce_rmsnorm_kernel<TT, WW>), dim3(kArBlocks), dim3(kArThreads), 0, s, a)
#define RX_ARN_T(TT)                                                  \
  do {                                                                \
    switch (a.world) {                                                \
      case 2: RX_ARN_W(TT, 2); break;                                 \
      case 3: RX_ARN_W(TT, 3); break;                                 \
      case 4: RX_ARN_W(TT, 4); break;                                 \
      case 5: RX_ARN_W(TT, 5); break;                                 \
      case 6: RX_ARN_W(TT, 6); break;                                 \
      case 7: RX_ARN_W(TT, 7); break;                                 \
      default: RX_ARN_W(TT, 8); break;                                \
    }                                                                 \
  } while (0)
  if (dtype == RX_BF16) RX_ARN_T(BF16);
  else RX_ARN_T(F16);
#undef RX_ARN_T
#undef RX_ARN_W
  return check_launch("rx_allreduce_rmsnorm");
}

int rx_ar_destroy(rx_ar_ctx* ctx) {
  delete reinterpret_cast<ArCtx*>(ctx);
  return RX_OK;
}

}  // extern "C"
