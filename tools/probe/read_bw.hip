// Dev probe: read-only streaming bandwidth of the chip (the ceiling a KV-streaming kernel is measured against).
// hipcc --offload-arch=gfx950 -O3 tools/probe/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) float f4;

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f4* __restrict__ p, size_t n4, float* out) {
  f4 acc = {0, 0, 0, 0};
  const size_t stride = static_cast<size_t>(gridDim.x) * 256 * UNROLL;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = (i + u * 256 < n4) ? __builtin_nontemporal_load(p + i + u * 256) : f4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

// the same bytes as runs of `run` bytes in a shuffled order: one 256-thread block reads one run (16 B per thread and
// step), the run index of block-iteration j is a multiplicative hash -- the access pattern of a paged KV pool
__global__ __launch_bounds__(256) void read_runs_kernel(const char* __restrict__ p, size_t nruns, int run, float* out) {
  f4 acc = {0, 0, 0, 0};
  for (size_t j = blockIdx.x; j < nruns; j += gridDim.x) {
    const size_t r = (j * 2654435761ull) % nruns;  // nruns is a power of two and the multiplier odd: a permutation
    const char* base = p + r * run;
    for (int off = threadIdx.x * 16; off < run; off += 256 * 16 * 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = (off + u * 4096 < run) ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + off + u * 4096)) : f4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += v[u];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

// one run per WAVE and step, all of its 16-byte loads issued before the first is used, next run's loads issued before
// this run's are consumed (two runs in flight per wave): with 8 waves per SIMD-pair that is the bytes-in-flight regime of
// the decode kernel
template <int LOADS>  // run = LOADS * 1024 B
__global__ __launch_bounds__(256) void read_runs_wave_kernel(const char* __restrict__ p, size_t nruns, float* out) {
  f4 acc = {0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const size_t wave = static_cast<size_t>(blockIdx.x) * 4 + (threadIdx.x >> 6), nwaves = static_cast<size_t>(gridDim.x) * 4;
  f4 cur[LOADS], nxt[LOADS];
  auto issue = [&](size_t j, f4 (&v)[LOADS]) {
    const size_t r = (j * 2654435761ull) % nruns;
    const char* base = p + r * (LOADS * 1024) + lane * 16;
#pragma unroll
    for (int u = 0; u < LOADS; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + u * 1024));
  };
  size_t j = wave;
  if (j < nruns) issue(j, cur);
  for (; j < nruns; j += nwaves) {
    if (j + nwaves < nruns) issue(j + nwaves, nxt);
#pragma unroll
    for (int u = 0; u < LOADS; ++u) acc += cur[u];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) cur[u] = nxt[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

// the decode kernel's shape: a wave owns SETS register sets of 16 runs of 1 KiB... no: of LOADS 16-byte loads per lane
// (LOADS KiB per set, as 4-KiB shuffled runs); a set is re-issued only `gap` sleep units after its data arrived (the QK^T
// MFMAs and the V -> LDS writes that free the registers) and `work` more units pass before the wave needs the next set
// (softmax + PV).  SETS = 1, LOADS = 16 is today's 16-bit kernel.
template <int LOADS, int SETS, int ROWMAP>
__global__ __launch_bounds__(256) void decode_like_kernel(const char* __restrict__ p, size_t nruns, int gap, int work, float* out) {
  f4 acc = {0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const size_t wave = static_cast<size_t>(blockIdx.x) * 4 + (threadIdx.x >> 6), nwaves = static_cast<size_t>(gridDim.x) * 4;
  f4 v[SETS][LOADS];
  auto issue = [&](size_t j, f4 (&x)[LOADS]) {  // tile j of this wave = LOADS / 4 shuffled 4-KiB runs
#pragma unroll
    for (int q = 0; q < LOADS / 4; ++q) {
      const size_t r = ((j * (LOADS / 4) + q) * 2654435761ull) % nruns;
      // ROWMAP: the decode kernel's lane map -- a run is 16 rows of 256 B, lane (row = lane & 15, g = lane >> 4) takes
      // 16 B at g * 16 + u * 64 of its row, so one instruction touches 64 B of each of 16 rows; else 1 KiB linear
      // ROWMAP 2: whole 128-byte lines -- lane (row8 = lane >> 3, c = lane & 7) takes 16 B at c * 16 of one HALF of a
      // row: instruction u covers the half (u & 1) of rows 8 (u >> 1) .. + 7
      const char* base = ROWMAP == 1   ? p + r * 4096 + (lane & 15) * 256 + (lane >> 4) * 16
                         : ROWMAP == 2 ? p + r * 4096 + (lane >> 3) * 256 + (lane & 7) * 16
                                       : p + r * 4096 + lane * 16;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        x[q * 4 + u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(
            base + (ROWMAP == 1 ? u * 64 : ROWMAP == 2 ? (u & 1) * 128 + (u >> 1) * 2048 : u * 1024)));
    }
  };
  const size_t ntiles = nruns / (LOADS / 4);
  size_t j = wave;
#pragma unroll
  for (int s = 0; s < SETS; ++s)
    if (j + s * nwaves < ntiles) issue(j + s * nwaves, v[s]);
  for (; j < ntiles; j += SETS * nwaves) {
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
      if (j + s * nwaves >= ntiles) break;
#pragma unroll
      for (int u = 0; u < LOADS; ++u) acc += v[s][u];          // waits for the set
      for (int i = 0; i < gap; ++i) __builtin_amdgcn_s_sleep(8);   // 8 * 64 cycles
      if (j + (s + SETS) * nwaves < ntiles) issue(j + (s + SETS) * nwaves, v[s]);
      for (int i = 0; i < work; ++i) __builtin_amdgcn_s_sleep(8);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

template <int LOADS, int SETS, int ROWMAP = 0>
static void bench_decode_like(const char* d, size_t bytes, float* o, hipEvent_t e0, hipEvent_t e1, int gap, int work) {
  const size_t nruns = bytes / 4096;
  const int blocks = 256 * 2;  // 8 waves per CU, as the decode kernel
  for (int rep = 0; rep < 2; ++rep) decode_like_kernel<LOADS, SETS, ROWMAP><<<blocks, 256>>>(d, nruns, gap, work, o);
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) decode_like_kernel<LOADS, SETS, ROWMAP><<<blocks, 256>>>(d, nruns, gap, work, o);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("decode-like%s, 8 waves/CU, %d set(s) of %2d KiB per wave, gap %d work %d (x512 cycles): %.2f TB/s\n",
         ROWMAP == 1 ? " (64 B of 16 rows per instruction)" : ROWMAP == 2 ? " (128 B of 8 rows per instruction)" : "", SETS, LOADS, gap, work, bytes * 5 / (ms * 1e-3) / 1e12);
}

template <int LOADS>
static void bench_wave(const char* d, size_t bytes, float* o, hipEvent_t e0, hipEvent_t e1) {
  const size_t nruns = bytes / (LOADS * 1024);
  const int blocks = 256 * 4;  // 16 waves per CU
  for (int rep = 0; rep < 2; ++rep) read_runs_wave_kernel<LOADS><<<blocks, 256>>>(d, nruns, o);
  hipEventRecord(e0);
  for (int rep = 0; rep < 5; ++rep) read_runs_wave_kernel<LOADS><<<blocks, 256>>>(d, nruns, o);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("shuffled runs of %6d B, one run per wave, two in flight: %.2f TB/s\n", LOADS * 1024, bytes * 5 / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = 4ull << 30;
  f4* d;
  float* o;
  hipMalloc(&d, bytes);
  hipMalloc(&o, 4);
  hipMemset(d, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    for (int rep = 0; rep < 2; ++rep) read_kernel<8><<<blocks, 256>>>(d, bytes / 16, o);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) read_kernel<8><<<blocks, 256>>>(d, bytes / 16, o);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("read-only, %5d blocks x 256 threads x 8 x 16 B in flight: %.2f TB/s\n", blocks, bytes * 5 / (ms * 1e-3) / 1e12);
  }
  for (int run : {1024, 2048, 4096, 16384, 32768, 262144}) {
    const size_t nruns = bytes / run;
    const int blocks = 8192;
    for (int rep = 0; rep < 2; ++rep) read_runs_kernel<<<blocks, 256>>>(reinterpret_cast<const char*>(d), nruns, run, o);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) read_runs_kernel<<<blocks, 256>>>(reinterpret_cast<const char*>(d), nruns, run, o);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("shuffled runs of %6d B: %.2f TB/s\n", run, bytes * 5 / (ms * 1e-3) / 1e12);
  }
  bench_wave<1>(reinterpret_cast<const char*>(d), bytes, o, e0, e1);
  bench_wave<2>(reinterpret_cast<const char*>(d), bytes, o, e0, e1);
  bench_wave<4>(reinterpret_cast<const char*>(d), bytes, o, e0, e1);
  bench_wave<8>(reinterpret_cast<const char*>(d), bytes, o, e0, e1);
  bench_wave<16>(reinterpret_cast<const char*>(d), bytes, o, e0, e1);
  const char* dc = reinterpret_cast<const char*>(d);
  for (int gw : {0, 2}) {
    bench_decode_like<16, 1>(dc, bytes, o, e0, e1, gw, gw);
    bench_decode_like<16, 1, 1>(dc, bytes, o, e0, e1, gw, gw);
    bench_decode_like<16, 1, 2>(dc, bytes, o, e0, e1, gw, gw);
    bench_decode_like<16, 2>(dc, bytes, o, e0, e1, gw, gw);
  }
  return 0;
}
