// Dev harness (not part of libradix_hip.so): the bench's instance of rx::extend_mfma64_kernel (PKC = 4, bf16, int64,
// paged) with parts of its pipelined run REMOVED by the kernel's dev template parameter VAR (a bit mask: 1 softmax VALU,
// 2 fragment reads, 4 staging, 8 tile barrier, 16 PV MFMAs, 32 QK^T MFMAs) -- what each part costs a lone wave per SIMD.
// Built and timed by DEV=64 tools/ext32_ab.py (interleaved A/B in one process).
#include "rx_extend64_kernel.inc"

namespace rx {
char* err_buf() { static thread_local char b[8]; return b; }
int fail(int code, const char*, ...) { return code; }

template <int VAR>
static int launch_var(const rx_extend_params* p, hipStream_t s) {
  Ext32Args a = make_ext32_args(p);
  a.q_pack = p->num_q_heads / p->num_kv_heads;
  a.hq = p->num_kv_heads;
  a.group = 1;
  a.mblocks = (p->max_extend_len * a.q_pack + 255) / 256;
  a.kv_fp8 = 0;
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  constexpr int QB = (VAR & 256) ? 1 : 2;  // VAR bit 256: the eight-wave form (one 32-row block per wave)
  auto kern = extend_mfma64_kernel<BF16, int64_t, false, 4, QB, VAR & 255>;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds64);
  (void)attr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512 / QB), kLds64, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
}  // namespace rx

extern "C" int rx_dev_extend64(const rx_extend_params* p, int variant, void* stream) {
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
