// K7, head_dim 128: launcher of rx::extend_mfma64_kernel (rx_extend64_kernel.inc -- the template lives in an include
// file so that a dev translation unit can instantiate ONE variant for ISA inspection and A/B timing).
#include "rx_extend64_kernel.inc"

namespace rx {

// launcher: PLAIN calls only (rx_extend32.hip decides); a has q_pack / hq / group / mblocks set for 256-row blocks
template <int PKC, int QB>
static void launch64_pk(const Ext32Args& a, bool bf16, bool idx64, bool linear, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  note_dispatch("extend_mfma64_kernel<%s, %s, %s, %d, %d, 0>", bf16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int", tbool(linear),
                PKC, QB);
#define RX_E64(TT, IT, LIN)                                                                                      \
  do {                                                                                                           \
    auto kern = extend_mfma64_kernel<TT, IT, LIN, PKC, QB>;                                                        \
    static const hipError_t attr =                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds64); \
    (void)attr;                                                                                                  \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512 / QB), kLds64, s, a);                                             \
  } while (0)
#define RX_E64_LIN(TT, IT) \
  do { if (linear) RX_E64(TT, IT, true); else RX_E64(TT, IT, false); } while (0)
#define RX_E64_IDX(TT) \
  do { if (idx64) RX_E64_LIN(TT, int64_t); else RX_E64_LIN(TT, int32_t); } while (0)
  if (bf16) RX_E64_IDX(BF16);
  else RX_E64_IDX(F16);
#undef RX_E64_IDX
#undef RX_E64_LIN
#undef RX_E64
}

// (the template's QB = 1 form -- the same pipeline on eight waves of 32 rows -- is instantiated by tools/probe/ext64_dev.hip only:
// it needs 64 more registers than a wave of a 512-thread workgroup has and spills)
void launch_extend64(const Ext32Args& a, bool bf16, bool idx64, bool linear, hipStream_t s) {
  if (a.q_pack == 4) launch64_pk<4, 2>(a, bf16, idx64, linear, s);
  else if (a.q_pack == 8) launch64_pk<8, 2>(a, bf16, idx64, linear, s);
  else launch64_pk<0, 2>(a, bf16, idx64, linear, s);
}

}  // namespace rx
