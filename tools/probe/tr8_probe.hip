// Discover the data movement of ds_read_b64_tr_b8 (gfx950) empirically: LDS rows of 64 bytes hold byte = 16 * row + col
// (row < 16, col < 16 used); lane 2q + p of every 16-lane group supplies the address of row q, cols 8p .. 8p+7 (the
// hypothesis); print what every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void probe(uint8_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint8_t)(((i / 64) * 16 + (i % 64)) & 0xff);
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  int q, p;
  if (mode == 0) { q = i >> 1; p = i & 1; }        // lane 2q+p -> row q, cols 8p..
  else { q = i & 7; p = i >> 3; }                   // lane 8p+q -> row q, cols 8p..
  const uint8_t* addr = lds + (8 * g + q) * 64 + 8 * p;
  auto ptr = (__attribute__((address_space(3))) v2i*)(uintptr_t)(uint32_t)(uintptr_t)addr;
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32(ptr);
  reinterpret_cast<v2i*>(out)[lane] = r;
}
int main() {
  uint8_t* d; hipMalloc(&d, 64 * 8);
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode);
    uint8_t h[512]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 8; ++j) printf(" r%02d.c%02d", h[l * 8 + j] >> 4, h[l * 8 + j] & 15);
      printf("\n");
    }
  }
  return 0;
}
