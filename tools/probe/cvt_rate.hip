// Dev probe: issue rate of the fp8 -> bf16 upcast (v_cvt_scalef32_pk_bf16_fp8) against a plain VALU op and a
// bit-twiddling alternative, one wave, s_memtime around 64 x 16 back-to-back independent instructions.
// hipcc --offload-arch=gfx950 -O3 tools/probe/cvt_rate.hip -o /tmp/cvt_rate && /tmp/cvt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void probe(uint32_t* out, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) {
    if (MODE == 0) {
      REP16(asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %4, 1.0\n v_cvt_scalef32_pk_bf16_fp8 %1, %5, 1.0 op_sel:[1,0,0]\n"
                         "v_cvt_scalef32_pk_bf16_fp8 %2, %6, 1.0\n v_cvt_scalef32_pk_bf16_fp8 %3, %7, 1.0 op_sel:[1,0,0]"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if (MODE == 1) {
      REP16(asm volatile("v_add_u32 %0, %4, %5\n v_add_u32 %1, %5, %6\n v_add_u32 %2, %6, %7\n v_add_u32 %3, %7, %4"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if (MODE == 2) {
      REP16(asm volatile("v_perm_b32 %0, %4, %5, %6\n v_pk_ashrrev_i16 %1, 4, %5\n v_and_b32 %2, %6, %7\n v_perm_b32 %3, %7, %4, %5"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if (MODE == 3) {
      uint64_t w0, w1;
      REP16(asm volatile("v_cvt_pk_f32_fp8 %0, %2\n v_cvt_pk_f32_fp8 %1, %3\n"
                         : "=v"(w0), "=v"(w1) : "v"(a0), "v"(a1));)
      r0 ^= (uint32_t)w0; r1 ^= (uint32_t)w1;
    } else {
      REP16(asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %5\n v_exp_f32 %2, %6\n v_exp_f32 %3, %7"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = (uint32_t)(t1 - t0); }
  out[1 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3;
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 4 * 128);
  const char* names[] = {"cvt_scalef32_pk_bf16_fp8", "v_add_u32", "perm/pk_ashr/and mix", "cvt_pk_f32_fp8", "v_exp_f32"};
  for (int m = 0; m < 5; ++m) {
    uint32_t h = 0;
    for (int rep = 0; rep < 2; ++rep) {
      switch (m) {
        case 0: probe<0><<<1, 64>>>(d, 1); break;
        case 1: probe<1><<<1, 64>>>(d, 1); break;
        case 2: probe<2><<<1, 64>>>(d, 1); break;
        case 3: probe<3><<<1, 64>>>(d, 1); break;
        default: probe<4><<<1, 64>>>(d, 1); break;
      }
      hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    }
    const int n = 64 * 16 * (m == 3 ? 2 : 4);
    printf("%-28s %u ticks for %d instr = %.3f ticks/instr\n", names[m], h, n, (double)h / n);
  }
  return 0;
}
