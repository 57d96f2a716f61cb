// K7, head_dim 128: launcher of rx::extend_mfma32_kernel (rx_extend32_kernel.inc -- the kernel template lives in an
// include file so that tools/probe/ext32_dev.hip can instantiate single variants side by side for A/B timing).
#include "rx_extend32_kernel.inc"

namespace rx {

// debug counters of the one counting instance (option ext32_count_redo): [0] softmax blocks of pipelined tiles, [1] redone
__device__ unsigned long long g_ext32_counters[2];

// launcher used by rx_extend.hip for head_dim == v_head_dim == 128
template <int NW, bool KV8, bool PLAIN, int PKC = 0>
static void launch32_nw(const Ext32Args& a, bool bf16, bool idx64, bool linear, bool vs, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  constexpr unsigned kLds = ext32_lds_bytes<NW>();  // 74 KiB (+ 68 KiB of epilogue rows with eight waves): dynamic
  note_dispatch("extend_mfma32_kernel<%s, %s, %s, %s, %d, %s, %s, %d>", bf16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int",
                tbool(linear), tbool(!PLAIN && vs), NW, tbool(KV8), tbool(PLAIN), PKC);
#define RX_E32(TT, IT, LIN, VS)                                                                        \
  do {                                                                                                 \
    auto kern = extend_mfma32_kernel<TT, IT, LIN, VS, NW, KV8, PLAIN, PKC>;                            \
    static const hipError_t attr = hipFuncSetAttribute(                                                \
        reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);       \
    (void)attr;                                                                                        \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), kLds, s, a);                                   \
  } while (0)
#define RX_E32_VS(TT, IT, LIN)                                          \
  do {                                                                  \
    if constexpr (PLAIN) RX_E32(TT, IT, LIN, false);                    \
    else if (vs) RX_E32(TT, IT, LIN, true);                             \
    else RX_E32(TT, IT, LIN, false);                                    \
  } while (0)
#define RX_E32_LIN(TT, IT) \
  do { if (linear) RX_E32_VS(TT, IT, true); else RX_E32_VS(TT, IT, false); } while (0)
#define RX_E32_IDX(TT) \
  do { if (idx64) RX_E32_LIN(TT, int64_t); else RX_E32_LIN(TT, int32_t); } while (0)
  if (bf16) RX_E32_IDX(BF16);
  else RX_E32_IDX(F16);
#undef RX_E32_IDX
#undef RX_E32_LIN
#undef RX_E32_VS
#undef RX_E32
}

// the unified (deterministic) instance: extend_mfma32_uni_kernel<T, IdxT, LINEAR, NW>
template <int NW>
static void launch32_uni(const Ext32Args& a, bool bf16, bool idx64, bool linear, hipStream_t s) {
  const unsigned grid = static_cast<unsigned>(a.bs) * a.hq * a.mblocks;
  constexpr unsigned kLds = ext32_lds_bytes<NW>();
  note_dispatch("extend_mfma32_uni_kernel<%s, %s, %s, %d>", bf16 ? "rx::BF16" : "rx::F16", idx64 ? "long" : "int", tbool(linear), NW);
#define RX_U32(TT, IT, LIN)                                                                            \
  do {                                                                                                 \
    auto kern = extend_mfma32_uni_kernel<TT, IT, LIN, NW>;                                             \
    static const hipError_t attr = hipFuncSetAttribute(                                                \
        reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);       \
    (void)attr;                                                                                        \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), kLds, s, a);                                   \
  } while (0)
#define RX_U32_LIN(TT, IT) \
  do { if (linear) RX_U32(TT, IT, true); else RX_U32(TT, IT, false); } while (0)
#define RX_U32_IDX(TT) \
  do { if (idx64) RX_U32_LIN(TT, int64_t); else RX_U32_LIN(TT, int32_t); } while (0)
  if (bf16) RX_U32_IDX(BF16);
  else RX_U32_IDX(F16);
#undef RX_U32_IDX
#undef RX_U32_LIN
#undef RX_U32
}

#ifdef RX_WITH_EXT64
void launch_extend64(const Ext32Args& a, bool bf16, bool idx64, bool linear, hipStream_t s);  // tools/probe/rx_extend64.hip (dev builds only)
#endif

int launch_extend32(const rx_extend_params* p, hipStream_t s) {
  const Options& opt = options();
  Ext32Args a = make_ext32_args(p);
  a.kv_fp8 = p->kv.kv_fp8;
  const bool vsc = p->v_scale != 1.0f;
  // (PLAIN instances compute ONE row offset for the K and the V load: both sides must have the same strides)
  const bool same_kv = p->kv.k_page_stride == p->kv.v_page_stride && p->kv.k_tok_stride == p->kv.v_tok_stride &&
                       p->k_stride_t == p->v_stride_t;
  // (PLAIN instances run EVERY tile on the pipelined body -- boundary tiles with masked selects -- whose staging addresses a
  // page by shift and mask: a linear pool or a power-of-two page)
  const bool plain_any = !a.kv_fp8 && !vsc && !a.unified_prefix && !a.custom_mask && a.window <= 0 &&
                         a.xai_len <= 0 && !(a.logit_cap > 0.f) && !a.bias && same_kv && opt.ext32_plain &&
                         (p->kv.page_size == 1 || a.page_size < 0 ||
                          (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride && p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride));
  // tiles a workgroup walks, estimated from the host-side hints: (mean prefix + half the longest extend) / 64
  const int est_tiles = (p->avg_kv_len_hint + p->max_extend_len / 2) / kTok;
  // Causal extends of a GQA-4 / GQA-8 model pack by themselves (bit-identical results): a 256-row block is then 64 or 32
  // tokens of the whole group -- its diagonal one boundary tile instead of four, a kv head's tiles staged once for the
  // group -- on the eight-wave PLAIN loop.  Round 3 did this for long extends over long prefixes only; round 4
  // (tools/extend_forms.py, bare operator, TFLOP/s unpacked four-wave / unpacked eight-wave / packed): the packed form
  // wins wherever a workgroup walks a few tiles -- no prefix + 2 k new tokens 812 / 710 / 895, + 512: 415 / 354 / 467;
  // 512 + 512: 731 / 683 / 789; 2 k + 128: 917 / 783 / 964; 2 k + 64: 634 / 454 / 981; 4 k + 32: 407 / 252 / 872;
  // config 3's 3584 + 512: 995 / 1033 / 1065 -- and loses only where there is next to nothing to do (no prefix + 256:
  // 280 / 224 / 260).  Options: ext32_autopack = 0 turns it off, ext32_pack_min_len / ext32_pack_min_tiles /
  // ext32_pack_min_wgs move the gates.
  // Packing trades workgroups for work: it is taken only while the packed grid still covers the chip (1 request x 32 k +
  // 64 tokens: 32 per-head workgroups 636 us, 8 packed ones 993 -- a batch that small needs the parallelism more).
  const int grp = p->num_kv_heads > 0 ? p->num_q_heads / p->num_kv_heads : 1;
  const int cus = device_cu_count();  // (of the current device; rx_set_option / options() are process-wide and not thread-safe)
  if (opt.ext32_autopack && a.q_pack == 1 && plain_any && (grp == 4 || grp == 8) && p->num_q_heads == grp * p->num_kv_heads &&
      p->is_causal && !p->skip_extend && p->max_extend_len >= opt.ext32_pack_min_len && est_tiles >= opt.ext32_pack_min_tiles &&
      static_cast<int64_t>(a.bs) * p->num_kv_heads * ((static_cast<int64_t>(p->max_extend_len) * grp + 255) / 256) >=
          (opt.ext32_pack_min_wgs < 0 ? cus : opt.ext32_pack_min_wgs))
    a.q_pack = grp;
  if (a.q_pack > 1) {  // the grid's heads are KV heads; their rows carry the q heads of the group
    a.hq = p->num_kv_heads;
    a.group = 1;
  }
  const bool linear = p->kv.page_size == 1 ||
                      (p->kv.k_page_stride == p->kv.page_size * p->kv.k_tok_stride &&
                       p->kv.v_page_stride == p->kv.page_size * p->kv.v_tok_stride);
  // Workgroup size.  Unpacked rows: one 256-query workgroup per CU (8 waves) is best when a (request, head, query block)
  // walks many tiles (config 3: 60 tiles, 1033 vs 995 TFLOP/s); with few tiles the per-workgroup prologue / epilogue
  // dominate and two 128-query workgroups per CU overlap them (no prefix, 2 Ki new tokens: 710 vs 812; 512 + 128: 513 vs
  // 679).  Packed rows on the PLAIN loop (the caller's q_pack or the self-packing above) take eight waves from
  // ext32_pack_min_tiles up.  Option ext32_small_wg: 0 / 1 force a form.
  const bool packed_plain = plain_any && (a.q_pack == 4 || a.q_pack == 8);
  // Either way eight waves only while their grid covers the chip: 2 requests x 8 k + 512 tokens are 128 eight-wave
  // workgroups (661 TFLOP/s) or 256 four-wave ones (862).
  const int64_t rows = static_cast<int64_t>(p->max_extend_len) * a.q_pack;
  const bool thin_grid = rows > 128 && static_cast<int64_t>(a.bs) * a.hq * ((rows + 255) / 256) <
                                           (opt.ext32_pack_min_wgs < 0 ? cus : opt.ext32_pack_min_wgs);
  // ... and a call whose rows fit ONE four-wave block anyway (<= 128: the same grid in both forms) takes four waves only
  // when two such workgroups per CU can co-reside; below that eight waves stage every tile twice as fast (split-KV
  // verify of 2 requests x 4 k + 8 draft tokens, 256 workgroups: 36.3 -> 32.2 us).
  const int min_wgs = opt.ext32_pack_min_wgs < 0 ? cus : opt.ext32_pack_min_wgs;
  const bool one_block = rows <= 128;
  const bool few_small = one_block && static_cast<int64_t>(a.bs) * a.hq < 2 * static_cast<int64_t>(min_wgs);
  const bool small_wg = opt.ext32_small_wg < 0
                            ? (!few_small && (thin_grid || est_tiles < (packed_plain ? opt.ext32_pack_min_tiles : opt.ext32_small_wg_tiles)))
                            : opt.ext32_small_wg != 0;
  // Packed rows on FEW tiles (round 5): four waves of the same PLAIN loop, 128-row blocks -- two workgroups share a CU and one
  // runs tiles while the other is in its prologue, diagonal tile or epilogue (per 256-row block: 3.8 + 1.0 us, + ~3 us until
  // the CU's next workgroup starts, against 1.74 us a tile).  tools/ext32_ab.py, eight vs four waves, TFLOP/s: no prefix +
  // 512 new tokens 412 / 468, + 1 k 603 / 650, + 2 k 814 / 831; 512 + 512 725 / 755; 2 k + 128 (33 tiles) 937 / 944 median,
  // 995 / 971 best.  Option ext32_pack4_tiles moves the gate (0: never).
  const bool pack4 = packed_plain && (small_wg || (opt.ext32_small_wg < 0 && !few_small && est_tiles < opt.ext32_pack4_tiles));
  const int nw = (small_wg || pack4) ? 4 : 8;
  a.mblocks = (p->max_extend_len * a.q_pack + nw * 32 - 1) / (nw * 32);
  const bool bf = p->dtype == RX_BF16, i64 = p->kv_indices_is_i64 != 0;
  const bool plain = plain_any && a.q_pack == 1;
#ifdef RX_WITH_EXT64
  // Option ext64 (dev builds, RX_WITH_EXT64=1; measured 0.83x of the eight-wave kernel on the config-3 chunk): PLAIN
  // eight-wave calls (packed or not) as the same 256-row blocks on FOUR waves of 64 rows, one per SIMD (tools/probe/rx_extend64.hip)
  if (opt.ext64 && nw == 8 && (packed_plain || plain) && !(a.page_size >= 0 && !linear)) {
    launch_extend64(a, bf, i64, linear, s);
    return RX_OK;
  }
#endif
  if (opt.ext32_count_redo && packed_plain && nw == 8 && a.q_pack == 4 && bf && i64 && !linear) {  // the counting instance
    unsigned long long* ctr = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&ctr), HIP_SYMBOL(g_ext32_counters)) != hipSuccess)
      return fail(RX_ERR_LAUNCH, "rx_extend_attn: no address for the debug counters");
    auto kern = extend_mfma32_count_kernel<BF16, int64_t, false, false, 8, false, true, 4>;
    constexpr unsigned kLds = ext32_lds_bytes<8>();
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    (void)attr;
    note_dispatch("extend_mfma32_count_kernel<rx::BF16, long, false, false, 8, false, true, 4>");
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(a.bs) * a.hq * a.mblocks), dim3(512), kLds, s, a, ctr);
    return RX_OK;
  }
  if (packed_plain) {  // packed rows on the PLAIN loop (GQA 4 / 8)
    if (nw == 8) {
      if (a.q_pack == 4) launch32_nw<8, false, true, 4>(a, bf, i64, linear, false, s);
      else launch32_nw<8, false, true, 8>(a, bf, i64, linear, false, s);
    } else {
      if (a.q_pack == 4) launch32_nw<4, false, true, 4>(a, bf, i64, linear, false, s);
      else launch32_nw<4, false, true, 8>(a, bf, i64, linear, false, s);
    }
    return RX_OK;
  }
  // The one-stage (unified) extend of deterministic inference with no other extra: its own instance, every tile on the
  // pipelined body in its row-deterministic form (rx_extend32_kernel.inc, UNI).  Option ext32_uni = 0: the general instance
  // (every tile on the three-phase masked body, as round 5).
  if (opt.ext32_uni && a.unified_prefix && !a.kv_fp8 && !vsc && !a.custom_mask && a.window <= 0 && a.xai_len <= 0 &&
      !(a.logit_cap > 0.f) && !a.bias && a.q_pack == 1 && same_kv && !p->skip_prefix && (linear || a.page_size < 0)) {
    a.sm_scale *= a.k_scale;  // (every key is a pool row: one scale)
    a.k_scale = 1.0f;
    if (small_wg) launch32_uni<4>(a, bf, i64, linear, s);
    else launch32_uni<8>(a, bf, i64, linear, s);
    return RX_OK;
  }
  if (small_wg) {
    if (plain) launch32_nw<4, false, true>(a, bf, i64, linear, false, s);
    else if (a.kv_fp8) launch32_nw<4, true, false>(a, bf, i64, linear, vsc, s);
    else launch32_nw<4, false, false>(a, bf, i64, linear, vsc, s);
  } else {
    if (plain) launch32_nw<8, false, true>(a, bf, i64, linear, false, s);
    else if (a.kv_fp8) launch32_nw<8, true, false>(a, bf, i64, linear, vsc, s);
    else launch32_nw<8, false, false>(a, bf, i64, linear, vsc, s);
  }
  return RX_OK;
}

}  // namespace rx

int rx_debug_counters(uint64_t* out2, int reset) {
  RX_REQUIRE(out2, "rx_debug_counters: null output");
  unsigned long long v[2] = {0, 0};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(rx::g_ext32_counters), sizeof(v)) != hipSuccess)  // (synchronises with the device)
    return rx::fail(RX_ERR_LAUNCH, "rx_debug_counters: %s", hipGetErrorString(hipGetLastError()));
  out2[0] = v[0];
  out2[1] = v[1];
  if (reset) {
    const unsigned long long z[2] = {0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(rx::g_ext32_counters), z, sizeof(z)) != hipSuccess)
      return rx::fail(RX_ERR_LAUNCH, "rx_debug_counters: reset failed");
  }
  return RX_OK;
}
