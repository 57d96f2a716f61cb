// Dev probe: what one LDS-DMA piece (1 KiB per wave instruction) costs the ISSUING wave, by form and by what the wave
// does between pieces.  hipcc --offload-arch=gfx950 -O3 tools/probe/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
//   form 0: global_load_lds_dwordx4, M0 saved / set / restored around every piece (the recipe of the guide)
//   form 1: global_load_lds_dwordx4, M0 set per piece, never restored
//   form 2: buffer_load_dwordx4 ... lds (32-bit per-lane offset against a buffer resource), M0 set per piece
//   form 4: nothing (the gap alone); form 5: global_load_lds_dwordx4 with M0 set ONCE (every piece to the same LDS address)
//   form 3: plain global_load_dwordx4 into registers + ds_write_b128 (the register-staged path), for reference
// gap: groups of 4 dependent v_fma between two pieces; waves: waves per workgroup issuing (1 or 4).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ void dma_full(const void* g, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma_norestore(const void* g, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma_buffer(u32x4 rsrc, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, 0 offen lds" : : "s"(rsrc), "v"(voff), "s"(lds) : "memory", "m0");
}

template <int FORM>
__global__ __launch_bounds__(256) void probe(const char* src, size_t bytes, int pieces, int gap, int waves, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w >= waves) return;
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)) + w * 16384;
  u32x4 rsrc;
  {
    const uint64_t b = reinterpret_cast<uint64_t>(src);
    rsrc[0] = static_cast<uint32_t>(b);
    rsrc[1] = static_cast<uint32_t>(b >> 32);
    rsrc[2] = static_cast<uint32_t>(bytes);
    rsrc[3] = 0x00020000;  // raw buffer, dword data format not needed for dwordx4
  }
  float x = lane * 1e-3f;
  // rows of 1152 B gathered 16 B per lane, like a latent KV tile
  const size_t span = bytes / 2;
  size_t off = (static_cast<size_t>(blockIdx.x) * 7919 * 1152 + w * 4096 + lane * 16) % span;
  if constexpr (FORM == 5) asm volatile("s_mov_b32 m0, %0" : : "s"(__builtin_amdgcn_readfirstlane(lds0)) : "memory", "m0");
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int p = 0; p < pieces; ++p) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (p & 15) * 1024);
    if constexpr (FORM == 0) dma_full(src + off, dst);
    else if constexpr (FORM == 1) dma_norestore(src + off, dst);
    else if constexpr (FORM == 2) dma_buffer(rsrc, static_cast<uint32_t>(off), dst);
    else if constexpr (FORM == 3) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + off);
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(dst + lane * 16) = v;
    } else if constexpr (FORM == 5) {
      asm volatile("global_load_lds_dwordx4 %0, off" : : "v"(src + off) : "memory");  // M0 set once, before the loop
    }
    off = (off + 1024 * 37) % span;
#pragma unroll 1
    for (int i = 0; i < gap; ++i)
      asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(x));
    if ((p & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // at most two batches of 8 in flight
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) out[blockIdx.x * 4 + w] = static_cast<uint32_t>(t1 - t0);
  if (x == 12345.f) out[0] = 0;
}

int main() {
  const size_t bytes = 64ull << 20;  // L2 / MALL resident after the first pass
  char* src;
  uint32_t* out;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  hipMalloc(&out, 256 * 4 * 4);
  const int pieces = 512;
  std::vector<uint32_t> h(256 * 4);
  for (int form : {4, 0, 1, 5, 2})
    for (int waves : {1, 4})
      for (int gap : {0, 10, 20, 40}) {
        for (int rep = 0; rep < 2; ++rep) {
          hipMemset(out, 0, 256 * 4 * 4);
          switch (form) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 65536, 0, src, bytes, pieces, gap, waves, out); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 65536, 0, src, bytes, pieces, gap, waves, out); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 65536, 0, src, bytes, pieces, gap, waves, out); break;
            case 4: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 65536, 0, src, bytes, pieces, gap, waves, out); break;
            case 5: hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 65536, 0, src, bytes, pieces, gap, waves, out); break;
            default: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 65536, 0, src, bytes, pieces, gap, waves, out); break;
          }
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), out, 256 * 4 * 4, hipMemcpyDeviceToHost);
        double sum = 0;
        int n = 0;
        for (int b = 0; b < 256; ++b)
          for (int w = 0; w < waves; ++w) { sum += h[b * 4 + w]; ++n; }
        const double per = sum / n / pieces;
        printf("form %d waves %d gap %3d: %7.1f ticks per piece (gap alone ~%d cycles), %.1f B/tick/CU\n", form, waves, gap, per,
               gap * 4, 1024.0 * waves / per);
      }
  return 0;
}
