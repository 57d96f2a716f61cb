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

__device__ unsigned g_dev_work[16];

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
  constexpr unsigned kLds = ext32_lds_bytes<8>();
  static_assert(VAR == 0 || VAR == 1 || VAR == 2, "add a trailing `int VAR = 0` template parameter to the kernel to study variants");
  if constexpr (VAR == 2) {  // packed rows on FOUR waves: 128-row blocks, two workgroups per CU
    a.mblocks = (p->max_extend_len * a.q_pack + 4 * 32 - 1) / (4 * 32);
    constexpr unsigned kLds4 = ext32_lds_bytes<4>();
    auto kern = extend_mfma32_kernel<BF16, int64_t, false, false, 4, false, true, 4>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds4);
    (void)attr;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(a.bs) * a.hq * a.mblocks), dim3(256), kLds4, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if constexpr (VAR == 1) {  // the resident-workgroup form (rx::extend_mfma32_persist_kernel)
    auto kern = extend_mfma32_persist_kernel<BF16, int64_t, false, 4>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    (void)attr;
    unsigned* work = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&work), HIP_SYMBOL(g_dev_work)) != hipSuccess) return -4;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const unsigned g = std::min<unsigned>(grid, static_cast<unsigned>(cus)) & ~7u;
    if (g < 8) return -5;
    hipLaunchKernelGGL(kern, dim3(g), dim3(512), kLds, s, a, work);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  auto kern = extend_mfma32_kernel<BF16, int64_t, false, false, 8, false, true, 4>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  (void)attr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), kLds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
}  // namespace rx

#ifdef RX_EXT32_TIMELINE
extern "C" int rx_dev_stamps(void* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rx::rx_ext32_timeline), sizeof(unsigned long long) * 8 * nblocks) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int rx_dev_extend32(const rx_extend_params* p, int variant, void* stream) {
  auto s = static_cast<hipStream_t>(stream);
  // (a call without a prefix has an empty index list: the int64 instance never reads it)
  if (p->num_q_heads != 4 * p->num_kv_heads || p->dtype != RX_BF16 || (!p->kv_indices_is_i64 && p->kv_indices != nullptr)) return -2;
  switch (variant) {
    case 0: return rx::launch_var<0>(p, s);
#define RX_V(n) case n: return rx::launch_var<n>(p, s);
    RX_DEV_VARIANT_CASES
#undef RX_V
    default: return -2;
  }
}
