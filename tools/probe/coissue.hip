// Do the two waves of a SIMD overlap one's MFMAs with the other's VALU?  512-thread blocks (waves w and w + 4 share a
// SIMD), one block per CU.  mode 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA, 4-7 VALU; 3: every wave
// alternates 8 MFMAs / 32 VALU (in phase); 4: the same, waves 4-7 start with the VALU half (out of phase).
// hipcc --offload-arch=gfx950 -O3 tools/probe/coissue.hip -o tools/probe/coissue && tools/probe/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA8(acc, a, b)                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                            \
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i_ & 1]) : "v"(a), "v"(b));
#define VALU32(x)                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_)                           \
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i_ & 7]) : "v"(c1), "v"(c2));
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int w = threadIdx.x >> 6;
  f32x16 acc[2] = {};
  bf16x8 a = {}, b = {};
  float x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  const float c1 = 0.999f, c2 = 0.001f;
  const bool second = w >= 4;
  for (int it = 0; it < iters; ++it) {
    if (mode == 0 || (mode == 2 && !second)) { MFMA8(acc, a, b) MFMA8(acc, a, b) }
    else if (mode == 1 || (mode == 2 && second)) { VALU32(x) VALU32(x) }
    else if (mode == 3 || (mode == 4 && !second)) { MFMA8(acc, a, b) VALU32(x) }
    else { VALU32(x) MFMA8(acc, a, b) }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 100, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration and wave: modes 0: 16 MFMA (32 cyc each); 1: 64 VALU; 2: either; 3 / 4: 8 MFMA + 32 VALU
    printf("mode %d: %.3f ms, %.1f ns per iteration\n", mode, ms, ms * 1e6 / iters);
  }
  return 0;
}
