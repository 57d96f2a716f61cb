// Dev harness (not part of libradix_hip.so): single instances of rx::extend_mfma32_kernel -- the bench's PLAIN / PKC = 4 /
// bf16 / int64 / paged one -- in the variants under study (template parameter VAR of the kernel), callable side by side
// from tools/ext32_ab.py for interleaved A/B timing in one process (cdna_hip_programming.md rule 24).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared <rx_extend32.hip's flags> -I include -I sglang_amd/csrc \
//         tools/probe/ext32_dev.hip -o tools/probe/libext32_dev.so
#include "rx_extend32_kernel.inc"

#ifndef RX_DEV_VARIANTS
#define RX_DEV_VARIANTS 0
#endif

namespace rx {
char* err_buf() { static thread_local char b[8]; return b; }
int fail(int code, const char*, ...) { return code; }

template <int VAR>
static int launch_var(const rx_extend_params* p, hipStream_t s) {
  Ext32Args a = make_ext32_args(p);
  const int grp = p->num_q_heads / p->num_kv_heads;
  a.q_pack = grp;
  a.hq = p->num_kv_heads;
  a.group = 1;
  a.mblocks = (p->max_extend_len * a.q_pack + 8 * 32 - 1) / (8 * 32);
  a.kv_fp8 = 0;
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  constexpr unsigned kLds = 2 * kBufBytes;
  static_assert(VAR == 0, "add a trailing `int VAR = 0` template parameter to the kernel to study variants");
  auto kern = extend_mfma32_kernel<BF16, int64_t, false, false, 8, false, true, 4>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  (void)attr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
}  // namespace rx

extern "C" int rx_dev_extend32(const rx_extend_params* p, int variant, void* stream) {
  auto s = static_cast<hipStream_t>(stream);
  if (p->num_q_heads != 4 * p->num_kv_heads || p->dtype != RX_BF16 || !p->kv_indices_is_i64) return -2;
  switch (variant) {
    case 0: return rx::launch_var<0>(p, s);
#define RX_V(n) case n: return rx::launch_var<n>(p, s);
    RX_DEV_VARIANT_CASES
#undef RX_V
    default: return -2;
  }
}
