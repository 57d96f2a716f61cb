/*
 * radix_hip.h -- C ABI of libradix_hip.so: the MI355X (gfx950) RadixAttention hot path.
 *
 * Every entry point replaces ONE kernel/operator of the reference (bytedance-iaas/sglang,
 * paths relative to python/sglang/); the citation next to each prototype names it.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer owned by the caller
 *     unless marked HOST; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - calls are stream-ordered, re-entrant, allocate nothing, never synchronise the host,
 *     and are therefore hipGraph-capturable.
 *   - return value: 0 (RX_OK) or a negative rx_status; rx_last_error() gives a
 *     thread-local message for the last failing call.  Nothing throws.
 *   - strides are in ELEMENTS of the tensor's dtype unless the name ends in _bytes.
 *   - slot 0 / page 0 is the reserved padding sink of the reference's pools
 *     (srt/mem_cache/allocator/token.py:41-46, paged.py:329-337).
 */
#ifndef RADIX_HIP_H_
#define RADIX_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 16 (round 6): the quick all-reduce (rx_qr_region_bytes, rx_qr_init, rx_quick_allreduce, rx_qr_destroy) + rx_rcp_f16_table.
 * 15 (round 6): rx_fused_fp8_qkv_kv_cache, rx_pool_alloc_extend_rows, rx_allreduce_det (deterministic fixed-order reduce),
 * rx_decode_units + rx_decode_params.unit_desc / unit_first_slots (appended);
 * the experimental rx_extend64 kernel left the product library (dev builds: RX_WITH_EXT64=1).
 * 14 (round 5): score_bias* appended to rx_decode_params / rx_extend_params (score_mod = relative_bias_score_mod).
 * 13 (round 5): rx_clock_probe.
 * 12 (round 5): rx_split_items_guarded, rx_debug_counters, rx_draft_decode_kv_indices.
 * 11 (round 4): rx_qknorm_rope_store_kv (fused QK-norm + RoPE + store).
 * 10 (round 4): rx_decode_params.rope_* (fused RoPE of the latent decode).
 * 9 (round 4): rx_last_dispatch, rx_set_option / rx_get_option.
 * 8 (round 3): rx_split_items; rx_decode_params.split_items / split_items_count / split_items_cap /
 * split_items_wgs_per_cu and extra_index / extra_rows; rx_num_kv_splits_balanced gained wg_target_mixed. */
#define RX_ABI_VERSION 16

typedef enum rx_status {
  RX_OK = 0,
  RX_ERR_INVALID_ARG = -1,  /* null pointer, negative size, misaligned buffer ... */
  RX_ERR_UNSUPPORTED = -2,  /* shape / dtype the gfx950 kernels do not cover      */
  RX_ERR_LAUNCH = -3        /* hipLaunchKernel / hipGetLastError reported failure */
} rx_status;

typedef enum rx_dtype { RX_BF16 = 0, RX_F16 = 1 } rx_dtype; /* dtype of q / o / new k, v */

/* Device-side error word bits (optional `err_flag`, int32, caller-zeroed): the kernels OR
 * these in instead of the reference's always-on device assert (kvcache.cuh:209). */
#define RX_DEVERR_SLOT_OOB 1   /* a KV slot index fell outside [0, size_limit) */
#define RX_DEVERR_AR_TIMEOUT 2 /* rx_allreduce gave up waiting for a peer's flag */

int rx_version(void);
/* sizeof of the parameter structs this library was built with: which = 0 rx_kv_layout, 1 rx_decode_params,
 * 2 rx_extend_params (-1 otherwise).  A binding compares them with its own struct definitions at load time, so a
 * stale library next to newer host code (or the reverse) fails loudly instead of reading a shifted layout. */
int64_t rx_abi_sizeof(int which);
const char* rx_last_error(void);

/* ---- dispatch switches and introspection (round 4; no reference counterpart: the reference picks its Triton
 * configs inside extend_attention_fwd / decode_attention_fwd, extend_attention.py:664-812, decode_attention.py:968-1044).
 * rx_extend_attn / rx_decode_attn choose a kernel instance from the call's shapes.  rx_last_dispatch() names the
 * instance the calling thread's last such call launched, spelled as the instance's own symbol demangles --
 * "extend_mfma32_kernel<rx::BF16, long, false, false, 8, false, true, 4>", optionally followed by "|" and run-time
 * facts ("|paged,g4"); "" before the first call.  tests/test_dispatch_coverage.py lists the attention-kernel instances
 * in the library's symbol table and fails if one of them has no parity case that provably ran it.
 * The few process-wide switches that override the default choice (A/B of kernel forms) are named ints: set through
 * rx_set_option, read once from RX_OPT_<NAME> at load -- the launch path itself never reads the environment.
 * Names: ext32_autopack, ext32_small_wg, ext32_plain, ext32_count_redo, extend_16x16_d128, extend_d256, extend_d256_at128,
 * extend_d256_at64, extend_d256_at96, extend_nd, extend_nd_big, extend_mla, extend_mla_shared_v, decode_mla8_dma, decode_mla8_t64,
 * merge_in_kernel_max_mb[_mla], ext32_uni (round 6), roctx; ar_fenced / qr_fenced (the all-reduce kernels' flag handshake
 * in its fenced form), ar_spin_log2 (polls before a
 * flag wait gives up), ar_blocks (blocks of the decode-sized all-reduce kernels; 0 = by message size), qr_max_blocks (the quick all-reduce's grid cap); ext64 only acts in a dev build (RX_WITH_EXT64=1).
 * NOT THREAD-SAFE: the switches are plain process-wide ints that the launch path reads without synchronisation.  Set them
 * before other threads launch (tests and A/B tools flip them from the one thread that also launches); a set that races with
 * a launch on another thread gives that launch the old or the new value, nothing is torn, but no ordering is promised. */
const char* rx_last_dispatch(void);
/* Debug counters (round 5; no reference counterpart).  With option ext32_count_redo = 1 the GQA-4 packed eight-wave D = 128
 * extend call (bf16, int64 indices, paged pool: the bench's instance) runs a counting twin of its kernel: out2[0] = 32-token
 * softmax blocks its pipelined tiles processed, out2[1] = how many of them took the sum check's redo (a block whose lane sum
 * exceeded 4096 against the standing reference max), summed over all such launches since the last reset.  Synchronises. */
int rx_debug_counters(uint64_t* out2, int reset);
/* Clock probe (round 5; no reference counterpart).  Launches ONE wave on `stream` that sleeps for spin_us microseconds of the
 * constant-rate clock and then writes out2_dev[0] = shader cycles (s_memtime), out2_dev[1] = 100-MHz ticks (s_memrealtime)
 * it saw pass (device memory, 16 bytes): launched on a side stream next to a kernel under test, cycles / ticks * 100 is
 * the shader clock in MHz that kernel sustains -- bench.py quotes the D = 128 extend launch against the MFMA peak at THAT clock
 * beside the nominal one (the launch is power-capped on random operands: DESIGN 4.2). */
int rx_clock_probe(uint64_t* out2_dev, int32_t spin_us, void* stream);
int rx_set_option(const char* name, int value);
int rx_get_option(const char* name, int* value);

/* ---- K1: KV store -------------------------------------------------------------------
 * store_cache (kernels/ops/kvcache/kvcache.py:57-110) -> store_kvcache
 * (kernels/jit/csrc/elementwise/kvcache.cuh:189-219), called from
 * MHATokenToKVPool._store_kv_layer (srt/mem_cache/memory_pool.py:2383-2430).
 *   k_cache[loc[i]] = k[i];  v_cache[loc[i]] = v[i]   (byte rows)
 * Rows with loc == skip_index (reference default 0) are not written; pass -1 to disable.
 * loc outside [0, size_limit) is not written and sets RX_DEVERR_SLOT_OOB in *err_flag.
 * Row byte counts and all strides must be multiples of 4 (reference: kvcache.py:38-45). */
int rx_store_kv(const void* k, const void* v, void* k_cache, void* v_cache, const void* loc,
                int64_t n, int64_t k_row_bytes, int64_t v_row_bytes, int64_t k_stride_bytes,
                int64_t v_stride_bytes, int64_t kc_stride_bytes, int64_t vc_stride_bytes,
                int loc_is_i64, int64_t size_limit, int64_t skip_index, int32_t* err_flag,
                void* stream);

/* ---- K2: ragged kv-index build --------------------------------------------------------
 * create_flashinfer_kv_indices_triton (kernels/ops/kvcache/kv_indices.py:8-46) plus the
 * cumsum of TritonAttnBackend._fill_kv_indptr_and_indices (triton_backend.py:386-404).
 *   kv_indptr[0] = 0; kv_indptr[i+1] = kv_indptr[i] + lens[i]            (int32)
 *   kv_indices[kv_indptr[i] + j] = req_to_token[req_pool_indices[i]][kv_start[i] + j]
 * kv_start may be NULL (= 0).  req_to_token is int32[*, row_stride]. */
int rx_build_kv_indices(const int32_t* req_to_token, int64_t row_stride,
                        const void* req_pool_indices, int pool_idx_is_i64, const void* lens,
                        int lens_is_i64, const int32_t* kv_start, int32_t* kv_indptr_out,
                        void* kv_indices_out, int out_is_i64, int bs, void* stream);

/* ---- K8 metadata: the unified kv list of the one-stage extend (round 5, deterministic inference) ----
 * build_unified_kv_indices (kernels/ops/attention/extend_attention.py:193-238, copy kernel :135-190), called from
 * TritonAttnBackend._forward_extend_unified (srt/layers/attention/triton_backend.py:1572-1712).
 *   prefix_lens[i]          = prefix_kv_indptr[i+1] - prefix_kv_indptr[i]                       (int32, may be NULL)
 *   unified_kv_indptr[0]    = 0;  [i+1] = [i] + prefix_lens[i] + extend_seq_lens[i]            (int32[bs+1])
 *   unified_kv_indices[unified_kv_indptr[i] + j] = j < prefix_lens[i]
 *        ? prefix_kv_indices[prefix_kv_indptr[i] + j] : extend_kv_indices[extend_start_loc[i] + j - prefix_lens[i]]   (int64)
 * prefix_kv_indices may be NULL when every prefix is empty.  max_tokens_per_request (> 0: an upper estimate of
 * prefix + extend of one request, 0 = unknown) only sizes the grid. */
int rx_build_unified_kv_indices(const int32_t* prefix_kv_indptr, const void* prefix_kv_indices, int prefix_is_i64,
                                const void* extend_start_loc, int start_is_i64, const void* extend_seq_lens,
                                int lens_is_i64, const void* extend_kv_indices, int extend_is_i64, int bs,
                                int64_t max_tokens_per_request, int32_t* unified_kv_indptr, int64_t* unified_kv_indices,
                                int32_t* prefix_lens, void* stream);

/* ---- EAGLE multi-step draft decode: the per-step page tables of the top-k branches (round 5) ----------------------
 * generate_draft_decode_kv_indices (kernels/ops/speculative/cache_locs.py:56-141), launched by
 * TritonMultiStepDraftBackend.common_template (srt/layers/attention/triton_backend.py:1929-1945).  For step i (iters = i + 1),
 * request b, branch k < topk, with n = seq_lens[b] and row = req_to_token[req_pool_indices[b]]:
 *   off = sum(seq_lens[:b]) * topk + b * iters * topk + k * (n + iters)
 *   kv_indices[i][off + j] = row[j], j < n;   kv_indices[i][off + n + j] = row[start + j], j < iters, where
 *   start = n + k * num_steps (page_size == 1 or topk == 1), else n / page * page + k * ceil((n % page + num_steps) / page) * page + n % page
 *   kv_indptr[i][z] = sum(positions[:z]) + z * iters for z = b * topk + k, with z = 0 standing for num_seqs * topk
 * (kv_indptr[i][0] is not written: the caller's buffer holds 0 there).  kv_indices rows are kv_indices_stride apart (int64 or
 * int32 words), kv_indptr rows kv_indptr_stride (>= num_seqs * topk + 1).  One launch, no host sync. */
int rx_draft_decode_kv_indices(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int pool_idx_is_i64,
                               const void* seq_lens, int seq_lens_is_i64, const void* positions, int positions_is_i64,
                               int num_seqs, int topk, int num_steps, int page_size, void* kv_indices, int kv_indices_is_i64,
                               int64_t kv_indices_stride, int32_t* kv_indptr, int64_t kv_indptr_stride, void* stream);

/* ---- K3: kv-split scheduler -----------------------------------------------------------
 * get_num_kv_splits_triton (kernels/ops/attention/metadata.py:11-60).
 * out is int32[num_seq * num_group]. */
int rx_num_kv_splits(const void* seq_lens, int seq_lens_is_i64, int num_seq, int num_group,
                     int num_head, int num_kv_head, int max_kv_splits, int device_core_count,
                     int32_t* out, void* stream);

/* MI355X-native split schedule (not a reference kernel: the reference's K3 formula above stays available
 * and bit-exact).  Measured on gfx950 (tools/decode_sweep.py): the decode kernel is fastest with one to two
 * workgroups per CU in total, e.g. bs 1 x ctx 32k: 8 splits (K3's answer) 74 us, 32 splits 38 us;
 * bs 64 x ctx 2k: 3 splits 110 us, 1 split 102 us.  Every request gets
 *   S = clamp(ceil(cu_count / (bs * wg_per_request)), 1, max_kv_splits), further capped at
 *   ceil(seq_len / min_tokens_per_split) so that short requests are not shredded;
 * wg_per_request = num_kv_heads * ceil(group / 16) (the kernel's workgroups per request and split). */
int rx_num_kv_splits_native(const void* seq_lens, int seq_lens_is_i64, int bs, int wg_per_request,
                            int cu_count, int max_kv_splits, int min_tokens_per_split, int32_t* out,
                            void* stream);

/* Length-aware form of the native schedule (round 2): a batch is rarely uniform, and ONE long request in a batch of
 * short ones is the whole kernel's tail when every request gets the same split count (64 requests, one of 32 k tokens
 * and 63 of 1 k: 516 us unsplit, 139 us with the long one cut 16 ways).  With total = sum of the lengths,
 *   t* = max(min_tokens_per_split, ceil(total * wg_per_request / wg_target))      (an even share of the batch)
 *   out[b] = 1 if 2 * len_b <= 3 * t*,  else min(max_kv_splits, ceil(len_b / t*)).
 * One launch, no host sync; max_kv_splits (the scratch's split slots) caps the count, so a host-side guess of the
 * largest count can never be overrun.
 * wg_target_mixed (round 3; 0 or <= wg_target: off): when that first pass leaves the batch MIXED -- some requests
 * cut, some whole, i.e. workgroups of different sizes -- the counts are re-derived with wg_target_mixed in place of
 * wg_target (the budget of the live-pairs grid: three workgroups per CU, all resident at once), and if rounding up
 * overshoots it (pairs * wg_per_request > wg_target_mixed) t* is scaled up by that ratio once.  A uniform batch keeps
 * the first pass: fewer, larger workgroups are faster there (4 x 16 k: 54 us at 512 workgroups, 56 at 768).
 * wg_target_mixed = -1: the ROUNDS rule instead, for the kernel's usual two workgroups per CU (what a graph-replayed
 * step runs): if the first pass of a mixed batch needs R1 = ceil(pairs * wg_per_request / wg_target) >= 2 rounds of
 * workgroups, the split requests are cut into pieces of p = R * a tokens, a = the mean length of the unsplit requests
 * (rounded up; never below the first pass's t*), R = the smallest of 1..4 with ceil(workgroups(p) / wg_target) <= R: a long request's pieces then end
 * with the last round of the short ones.  R1 = 1 keeps the first pass (all resident: the even share fills the slots).
 * The FILL rule (round 4; only with wg_target_mixed != 0, i.e. the live-pairs schedules; CUs = wg_target / 2) comes
 * before all of the above: a NEAR-UNIFORM batch (2 max(len) <= 3 mean(len) over the live requests) whose whole-request
 * workgroups number blocks = live * wg_per_request with 0.7 CUs <= blocks < 3 CUs takes ONE count S for everybody --
 * S = 1 when blocks <= CUs (the launch is bound by HBM as a whole: a split only adds its merge), otherwise the smallest
 * S <= min(max_kv_splits, 6) with blocks * S >= 0.85 * ceil(blocks * S / CUs) * CUs (the pieces fill whole rounds of
 * CUs: the launch lasts as long as the CU with the most bytes), or the best-filling S if none reaches 0.85 -- and
 * out[b] = max(1, min(S, len_b / 256)).  Below 0.7 CUs blocks of a near-uniform batch, where everybody is cut: if the even
 * share's count n = ceil(mean(len) / t*) gives workgroups that fill < 85 % of whole rounds of CUs (or fewer than 1.5 per
 * CU), the nearest count (n - 1, n + 1, n - 2, ...; >= 2, pieces >= min_tokens_per_split) that fills replaces it for
 * everybody; otherwise the even share stands.  With 0 < wg_target_mixed <= wg_target (kernels without the live-pairs grid whose
 * workgroups are latency-bound one by one: the MLA kernels) the fill rule is instead: near-uniform and blocks >= 0.8 CUs
 * -> every count is 1. */
int rx_num_kv_splits_balanced(const void* seq_lens, int seq_lens_is_i64, int bs, int wg_per_request, int wg_target,
                              int max_kv_splits, int min_tokens_per_split, int wg_target_mixed, int32_t* out,
                              void* stream);

/* The work-item table of a split schedule (rx_decode_params.split_items): for i over the requests in `order` (a
 * permutation, e.g. longest first; NULL = identity) and s < num_kv_splits[order[i]]: items[2 k] = order[i],
 * items[2 k + 1] = s, k counting up; count[0] = number of pairs written (clamped to cap: pairs beyond it are NOT
 * written and count[0] still says how many there would be -- the caller sizes cap from bs * max slots or from its host
 * copy of the lengths).  One launch of one block, no host sync. */
int rx_split_items(const int32_t* num_kv_splits, const int32_t* order, int bs, int32_t* items, int32_t* count, int cap,
                   void* stream);

/* The same table for a caller that sized cap from a BOUND (graph replay: the counts are refilled on the device, the
 * grid is fixed at capture): if the schedule holds more live pairs than cap (>= bs, checked), the schedule itself is
 * replaced -- every num_kv_splits[b] becomes 1, the table becomes the bs whole-request pairs in `order`, count[0] = bs
 * and overflow[0] (optional, sticky: only ever set) = 1.  The decode launch that follows reads a consistent schedule:
 * slower than the one asked for, never a request without its workgroups. */
int rx_split_items_guarded(int32_t* num_kv_splits, const int32_t* order, int bs, int32_t* items, int32_t* count, int cap,
                           int32_t* overflow, void* stream);

/* ---- KV buffer addressing shared by decode / extend ------------------------------------
 * element offset of (slot, kv_head) = (slot / page_size) * page_stride
 *                                   + (slot % page_size) * tok_stride + kv_head * head_stride
 * NHD [slots, Hkv, D] (memory_pool.py:2030-2041): page_stride = page_size * Hkv*D,
 *     tok_stride = Hkv*D, head_stride = D.
 * HND [pages, Hkv, page, D] (memory_pool.py:2032-2036): page_stride = Hkv*page*D,
 *     tok_stride = D, head_stride = page*D.
 * (mirrors _extract_kv_strides, kernels/ops/attention/decode_attention.py:39-88) */
typedef struct rx_kv_layout {
  const void* k_buf;
  const void* v_buf;
  int32_t page_size;
  int64_t k_page_stride, k_tok_stride, k_head_stride;
  int64_t v_page_stride, v_tok_stride, v_head_stride;
  /* 0: the pool holds q's 16-bit dtype.  1: OCP fp8 e4m3fn bytes (--kv-cache-dtype fp8_e4m3:
   * MHATokenToKVPool store_dtype uint8, srt/mem_cache/memory_pool.py:2043-2094,2305-2381; MLA
   * latent rows :4046-4138): strides are in elements = bytes, kernels upcast to q's dtype on load
   * (exact) and fold k_scale / v_scale as the reference does (decode_attention.py:499,603). */
  int32_t kv_fp8;
} rx_kv_layout;


/* K1 for pools whose slot is not one contiguous row (HND [pages, Hkv, page, D]): 16-bit k [n, Hkv*Dk]
 * / v [n, Hkv*Dv] rows (token strides in elements) scattered to the rx_kv_layout addresses.  The
 * reference uses torch index_put for this layout (srt/mem_cache/memory_pool.py:2372-2379). */
int rx_store_kv_layout(const void* k, const void* v, const rx_kv_layout* lay /* HOST */,
                       const void* loc, int64_t n, int num_kv_heads, int head_dim, int v_head_dim,
                       int64_t k_stride_t, int64_t v_stride_t, int loc_is_i64, int64_t size_limit,
                       int64_t skip_index, int32_t* err_flag, void* stream);

/* Quantising K1/K12: 16-bit k [n, Hkv*Dk] / v [n, Hkv*Dv] rows -> fp8 e4m3fn bytes at the
 * rx_kv_layout addresses (lay->kv_fp8 must be 1).  Per element: t = x / scale rounded to the source
 * dtype (the reference's in-place cache_k.div_(k_scale), memory_pool.py:2334-2343; skipped when
 * scale == 1), then round-to-nearest-even to e4m3fn (.to(float8_e4m3fn)); |t| > 448 saturates
 * where torch produces NaN.  The MLA two-tensor write (set_mla_kv_buffer_triton[_fp8_quant],
 * kernels/ops/kvcache/mla_buffer.py; memory_pool.py:4046-4110) is the Hkv = 1 case with
 * k = nope[n,512] -> row bytes [0,512) and v = rope[n,64] -> v_buf = k_buf + 512. */
int rx_store_kv_fp8(const void* k, const void* v, const rx_kv_layout* lay /* HOST */,
                    const void* loc, int64_t n, int num_kv_heads, int head_dim, int v_head_dim,
                    int64_t k_stride_t, int64_t v_stride_t, int src_dtype /* rx_dtype */,
                    float k_scale, float v_scale, int loc_is_i64, int64_t size_limit,
                    int64_t skip_index, int32_t* err_flag, void* stream);

/* fused_fp8_qkv_kv_cache (kernels/ops/kvcache/fused_fp8_qkv_kv_cache.py:35-80; kernel
 * kernels/jit/csrc/attention/fused_fp8_qkv_kv_cache.cuh:56-91; caller trtllm_mha_backend.py:764-790): fp8 e4m3fn
 * quantisation of the step's K / V rows into the paged pool AND, when q != NULL, of the q rows into a dense
 * q_out [n, q_dim] -- one launch.  Per element y = float(x) * (1.0f / *scale), saturated to +-448, round-to-nearest-even
 * (the reference kernel's static_cast<fp8_e4m3_t>; q uses scale 1).  NOTE the arithmetic differs from rx_store_kv_fp8's
 * (set_kv_buffer divides and rounds to the source dtype first): the two reference operators differ the same way.
 * k_scale / v_scale: DEVICE fp32 scalars (NULL = 1.0), as the reference passes them.  No slot is skipped (the reference
 * writes slot 0 too); a slot outside [0, size_limit) is dropped and flagged in err_flag. */
int rx_fused_fp8_qkv_kv_cache(const void* q /* or NULL */, const void* k, const void* v, void* q_out /* NULL iff q is */,
                              const rx_kv_layout* lay /* HOST; kv_fp8 = 1 */, const void* cache_loc, int loc_is_i64,
                              const float* k_scale, const float* v_scale, int64_t n, int q_dim, int num_kv_heads,
                              int head_dim, int v_head_dim, int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t,
                              int src_dtype /* rx_dtype */, int64_t size_limit, int32_t* err_flag, void* stream);

/* K12 read side: get_mla_kv_buffer_triton (kernels/ops/kvcache/mla_buffer.py; caller
 * MLATokenToKVPool.get_mla_kv_buffer, memory_pool.py:4117-4138).  Gathers latent rows
 *   nope_out[i, :] = rows[loc[i], 0:nope_cols],  rope_out[i, :] = rows[loc[i], nope_cols:nope_cols+rope_cols]
 * into dense [n, cols] outputs of dst_dtype (rx_dtype).  The pool is that same 16-bit dtype
 * (pure copy) or fp8 e4m3fn (kv_fp8 = 1: exact upcast).  Column counts are multiples of 8. */
int rx_get_mla_kv(const void* kv_buf, int64_t row_stride /* elements */, int kv_fp8, const void* loc,
                  int loc_is_i64, int64_t n, int nope_cols, int rope_cols, void* nope_out,
                  void* rope_out, int dst_dtype, int64_t size_limit, int32_t* err_flag, void* stream);

/* ---- K4/K5/K6: decode attention ---------------------------------------------------------
 * decode_attention_fwd (kernels/ops/attention/decode_attention.py:968-1044): stage 1
 * _fwd_grouped_kernel_stage1 (:383-608) / _fwd_kernel_stage1 (:96-281), stage 2
 * _fwd_kernel_stage2 (:731-805); caller TritonAttnBackend.forward_decode
 * (srt/layers/attention/triton_backend.py:1714-1864).
 *
 * Token -> slot lookup, one of:
 *   (a) kv_indptr (int32[bs+1]) + kv_indices (int32 or int64)  -- the reference contract;
 *   (b) kv_indices == NULL: req_to_token (int32[*, req_row_stride]) + req_pool_indices +
 *       seq_lens, read in-kernel (what decode_attention_cpu does, aot/csrc/cpu/decode.cpp:1586);
 *       saves the 8 MiB/step kv_indices round trip.
 * Split-KV: request b is cut into num_kv_splits[b] (<= max_kv_splits) pieces of
 * cdiv(cdiv(seq,splits),32)*32 tokens exactly as the reference (:466-472); partials go to
 * attn_logits fp32[bs,Hq,max_kv_splits,Dv] / attn_lse fp32[bs,Hq,max_kv_splits] and are
 * merged by the stage-2 kernel.  num_kv_splits == NULL or max_kv_splits == 1 runs a single
 * pass that writes `o` directly (no scratch traffic). */
typedef struct rx_decode_params {
  const void* q; /* [bs, Hq, Dk] */
  void* o;       /* [bs, Hq, Dv] */
  int64_t q_stride_t, q_stride_h, o_stride_t, o_stride_h;
  rx_kv_layout kv;
  const int32_t* kv_indptr; /* mode (a) */
  const void* kv_indices;   /* mode (a); NULL selects mode (b) */
  int32_t kv_indices_is_i64;
  const int32_t* req_to_token; /* mode (b) */
  int64_t req_row_stride;
  const void* req_pool_indices;
  int32_t req_pool_indices_is_i64;
  const void* seq_lens;
  int32_t seq_lens_is_i64;
  const int32_t* num_kv_splits; /* int32[bs] or NULL */
  int32_t max_kv_splits;
  float* attn_logits;
  float* attn_lse;
  int32_t bs, num_q_heads, num_kv_heads, head_dim, v_head_dim;
  float sm_scale, k_scale, v_scale, logit_cap;
  const float* sinks; /* fp32[Hq] or NULL (stage 2, :796-798) */
  int32_t dtype;      /* rx_dtype of q / kv / o */
  /* Grok temperature (decode_attention.py:156-160,212-213): L > 0 multiplies request b's scores by
   * log2(seq_len_b - 1) / log2(L) when seq_len_b - 1 > L (after scale and cap); <= 0 = off */
  int32_t xai_temperature_len;
  /* Shared-prefix (cascade) decode, see rx_shared_prefix_plan.  kv_start (int32[bs] or NULL, lookup mode
   * (b) only): request b attends tokens [kv_start[b], seq_len_b) of its req_to_token row; split sizes and
   * the stage-2 merge use that suffix length, the Grok temperature still uses seq_len_b.
   * extra_o / extra_lse: num_extra_partials more partial results per (request, head) -- dense
   * [num_extra, bs, Hq, Dv] of dtype (already divided by their own softmax sums, NOT multiplied by v_scale)
   * and natural-log LSEs fp32 [num_extra, bs, Hq] (-inf = empty partial, row ignored) -- that stage 2
   * merges with the kv splits; with max_kv_splits == 1 the D = 64 / 128 kernel folds them into its own
   * epilogue (no fp32 partials, no second launch), the other kernels need max_kv_splits > 1. */
  const int32_t* kv_start;
  const void* extra_o;
  const float* extra_lse;
  int32_t num_extra_partials;
  /* Several shared prefixes in one batch (one per radix-tree node, round 3): the extra partials hold rows for the
   * MEMBERS of the groups only, in group order.  extra_index (int32[bs] or NULL = the identity over bs rows): row of
   * request b inside one partial, < 0 = b belongs to no group (no extras merged); extra_rows: rows per partial. */
  const int32_t* extra_index;
  int32_t extra_rows;
  /* 0 = stage 1 then stage 2 (default); 1 = stage 1 only (partials to attn_logits / attn_lse);
   * 2 = stage 2 only.  Lets a caller produce the extra partials on another stream while stage 1 runs
   * and join before the merge.  1 / 2 need max_kv_splits > 1. */
  int32_t stages;
  /* Optional: stage 2 inside the stage-1 kernel.  int32[>= bs * num_q_heads], device memory, ZERO before the first
   * call; the kernels leave it zero.  With it (and stages == 0, no extra partials, max_kv_splits a multiple of 8, 16-byte
   * aligned attn_logits / attn_lse, at most 4 MiB of partials, the D = 64 / 128 or MLA kernel) the last workgroup of a (request, head block) to finish merges that block's kv-split partials
   * itself and no second kernel is launched -- the same arithmetic as stage 2 (_decode_softmax_reducev_fwd,
   * decode_attention.py:731-805).  One buffer serves one stream of calls (not two concurrent ones); after a faulted
   * launch the caller zeroes it again.  NULL: stage 2 is its own launch, as in the reference. */
  int32_t* merge_counters;
  /* Optional: the step's KV store fused into the kernel.  k_new / v_new = the new token's rows of every request,
   * dtype [bs, Hkv, D] with element strides (token, head); request b's newest position (seq_len_b - 1, whose slot the
   * req_to_token row / kv_indices already names) is then READ from them and WRITTEN to that slot by the kernel -- K1
   * (store_cache, memory_pool.py:2383-2430) without its own launch, as the reference's CPU kernel does
   * (decode_attention_cpu, aot/csrc/cpu/decode.cpp).  Needs the D = 64 / 128 kernel on a 16-bit pool, at most 16 q heads
   * per kv head (one workgroup per row) and a call that runs stage 1 (stages 0 or 1; kv_start and extra partials are
   * fine: the newest token is the last of the attended suffix); anything else is an error (store with rx_store_kv*
   * first).  NULL: the pool already holds the token. */
  const void* k_new;
  const void* v_new;
  int64_t k_new_stride_t, k_new_stride_h, v_new_stride_t, v_new_stride_h;
  /* Optional: int32[bs], a permutation of 0..bs-1 -- the order in which the D = 64 / 128 kernel's workgroups take the
   * requests (e.g. argsort of the lengths, descending: the last round of workgroups of a ragged batch is then its short
   * requests).  Results do not depend on it.  NULL: request b is block b. */
  const int32_t* request_order;
  /* Optional (0 = bs * max_kv_splits): an upper estimate of the (request, split) pairs that really write a partial.
   * With the length-aware schedule most requests of a large batch have ONE split and write none (direct output), so
   * the 4-MiB bound of the in-kernel stage 2 is taken on this count instead of on the split slots (with split_items
   * below and no hint, on split_items_cap).  A performance hint only: results do not depend on it. */
  int32_t partial_pairs_hint;
  /* Optional (round 3; D = 64 / 96 / 128 / 256 kernel): the LIVE (request, split) pairs of the split schedule,
   * compacted by rx_split_items: int32 pairs {request, split}.  Without it the grid is bs x max_kv_splits split slots
   * with the split the slowest block dimension, and a request whose count is below the slots leaves dead workgroups
   * behind -- in a batch with ONE long request (64 requests: one of 32 k tokens cut 22 ways, 63 of 1 k with one split)
   * every live workgroup of split s >= 1 sits behind 504 dead ones in dispatch order, and the long request's last
   * splits start tens of microseconds late.  With the table, block (q block, kv head, item i) takes pair i: live work
   * only, longest requests first; split_items_cap sizes the grid (>= the device-side count; pairs beyond the count
   * exit at once -- they are at the END of the grid, which is what a graph-replayed step with an upper bound needs). */
  const int32_t* split_items;       /* int32[2 * split_items_cap], device */
  const int32_t* split_items_count; /* int32[1], device: live pairs */
  int32_t split_items_cap;
  /* 0 / 2: the kernel's usual register budget (two workgroups per CU).  3: the three-per-CU instance (plain D = 128
   * 16-bit kernel without k_new only; ignored elsewhere) -- for a MIXED batch whose schedule was made for 3 x CUs
   * near-equal pieces (rx_num_kv_splits_balanced, wg_target_mixed), all resident at once.  A uniform batch is ~0.5 %
   * faster at two, and a caller that cannot know (a captured graph replayed with new lengths) leaves it 0. */
  int32_t split_items_wgs_per_cu;
  /* ---- fused RoPE of the latent (MLA) decode, round 4: decode_attention_fwd_grouped_rope
   * (kernels/ops/attention/rocm_mla_decode_rope.py:45-439, called from forward_mla_fused_rope_rocm.py:177-215).
   * head_dim 576 = 512 latent + 64 rope, v_head_dim 512, one kv head, a 16-bit pool.  With rope_cos_sin set, q's rope
   * columns 512..575 arrive UNrotated and are rotated inside the kernel at rope_positions[b] (neox: partner c <-> c + 32,
   * GPT-J style: pairs (2 i, 2 i + 1); fp32 arithmetic, one rounding to the 16-bit operand), and so is the k_pe of each
   * request's NEWEST token (position seq_len - 1), every older row of the pool being rotated already.  Two forms:
   *   - the reference's: the newest row sits in the pool with its k_pe not rotated; the kernel uses the rotated values,
   *     returns them in rope_k_pe_out [bs, 64] and leaves the pool alone (the caller stores the row again);
   *   - k_new = [bs, 576] (k_new_stride_t elements apart; v_new unused): the step's rows have NOT been stored; the kernel
   *     reads them from k_new, rotates the rope part, attends, and writes the finished row to its slot -- RoPE, KV store
   *     and attention in one launch.
   * rope_cos_sin: [max_pos, >= 64] rows of cos (32) | sin (32), fp32 or the call's 16-bit dtype (_is_f32);
   * rope_positions int64 / int32 [bs]; rope_dim must be 64; rope_k_pe_out may be NULL. */
  const void* rope_cos_sin;
  int32_t rope_cos_sin_is_f32;
  int64_t rope_cos_sin_stride;
  const void* rope_positions;
  int32_t rope_positions_is_i64;
  int32_t rope_dim;
  int32_t rope_is_neox;
  void* rope_k_pe_out;
  int64_t rope_k_pe_out_stride;
  /* ---- relative-position score bias, round 5 (ABI 14): the reference's score_mod = relative_bias_score_mod with
   * aux_tensors = [rel_logits] (kernels/ops/attention/score_mod.py:44-56; call sites decode_attention.py:215-227,
   * 539-551; used by srt/models/inkling_common/attn.py:934-946).  score_bias [bs, Hq, score_bias_len] (fp32, or the
   * call's 16-bit dtype; last dim contiguous, element strides for request and head): the score of request b, head h
   * against list position n gets + score_bias[b, h, r], r = (len_b - 1) - n, when 0 <= r < score_bias_len -- after
   * scale, logit cap and the Grok temperature, before the softmax; len_b = the request's kv length.  D = 64 / 96 /
   * 128 / 256 kernel and the generic kernel; not the MLA kernels.  NULL: off. */
  const void* score_bias;
  int32_t score_bias_is_f32;
  int32_t score_bias_len;
  int64_t score_bias_stride_t, score_bias_stride_h;
  /* ---- per-unit descriptors, round 6 (ABI 15; no reference counterpart): tables built ONCE per forward by rx_decode_units
   * (all layers share them) that shorten every decode launch's prologue.  A workgroup otherwise walks a chain of dependent
   * round trips before its first K / V byte is requested -- kernel arguments -> (request, split) pair -> request row /
   * length / split count -> slot ids of its first tiles -> K / V rows; ~1.1 us each at kernel start, tools/decode_timeline.py
   * -- with the tables the chain is kernel arguments -> {descriptor, first-tile slot ids} -> K / V rows.
   * unit_desc int32 [units][8] = b, split, seq_len, num_kv_splits[b], row offset (low, high 32 bits, elements of
   * req_to_token), lo, hi (the split's token range); unit_first_slots int32 [units][128] = the slot of token
   * min(lo + j, hi - 1).  Unit u = the u-th (request, split) pair of split_items, or (max_kv_splits == 1) the u-th request in
   * launch order.  req_to_token mode of the D = 64 / 96 / 128 / 256 kernel only (ignored elsewhere); both or neither;
   * the tables must have been built from the SAME req_to_token / seq_lens / schedule the call passes. */
  const int32_t* unit_desc;
  const int32_t* unit_first_slots;
} rx_decode_params;

int rx_decode_attn(const rx_decode_params* p /* HOST */, void* stream);
/* Builds rx_decode_params.unit_desc / unit_first_slots (see there) on the device: one launch, stream ordered, allocation free.
 * split_items / split_items_count (with their cap = the number of table rows to fill) or, for an unsplit step, NULL and
 * cap = bs; request_order as in rx_decode_params (or NULL).  unit_desc holds cap x 8, unit_first_slots cap x 128 int32. */
int rx_decode_units(const int32_t* req_to_token, int64_t req_row_stride, const void* req_pool_indices,
                    int req_pool_indices_is_i64, const void* seq_lens, int seq_lens_is_i64, const int32_t* num_kv_splits,
                    int max_kv_splits, const int32_t* split_items, const int32_t* split_items_count, int cap,
                    const int32_t* request_order, int bs, int32_t* unit_desc, int32_t* unit_first_slots, void* stream);

/* ---- shared-prefix (cascade) decode plan ----------------------------------------------------------
 * SURVEY 8f-2.  The reference has only the building block (merge_state); with RadixAttention every request
 * of a batch that hit the same radix-tree path carries the SAME leading slots in its req_to_token row
 * (radix_cache.py:352-430 match_prefix -> allocation.py:55-101 write_cache_indices), and its decode
 * kernel re-reads those rows once per request.  This plan finds the batch-wide common prefix ON DEVICE
 * (no host sync; graph-replay safe) so that the caller can
 *   1. run rx_extend_attn once over the shared rows with all bs decode queries as the M dimension
 *      (num_chunks pseudo-requests, one per chunk of the prefix; skip_extend, not causal, LSE out),
 *   2. run rx_decode_attn over the per-request suffixes (kv_start) and let its stage 2 merge the
 *      chunk partials (extra_o / extra_lse).
 * Outputs (all int32, device):
 *   plan[0] = L = largest t <= min(min_b seq_len_b, max_shared) with
 *             req_to_token[rpi[b], 0:t] == req_to_token[rpi[0], 0:t] for every b; 0 if that is < min_shared
 *   plan[1]   scratch
 *   chunk_indptr[0..num_chunks] = min(i * per, L), per = ceil(ceil(L / num_chunks) / chunk_align) * chunk_align
 *   shared_indices[0:L] = req_to_token[rpi[0], 0:L]      (capacity max_shared)
 *   kv_start[b] = L,  suffix_lens[b] = seq_len_b - L. */
int rx_shared_prefix_plan(const int32_t* req_to_token, int64_t req_row_stride, const void* req_pool_indices,
                          int req_pool_indices_is_i64, const void* seq_lens, int seq_lens_is_i64, int bs,
                          int32_t max_shared, int32_t min_shared, int num_chunks, int chunk_align,
                          int32_t* plan, int32_t* chunk_indptr, int32_t* shared_indices, int32_t* kv_start,
                          int32_t* suffix_lens, void* stream);

/* ---- K7: extend attention ---------------------------------------------------------------
 * extend_attention_fwd (kernels/ops/attention/extend_attention.py:664-812) -> _fwd_kernel
 * (:241-661); caller TritonAttnBackend.forward_extend (triton_backend.py:1250-1437).
 * Request i owns queries qo_indptr[i]..qo_indptr[i+1] (new tokens; their K/V are the
 * contiguous k_extend/v_extend rows) and a cached prefix kv_indices[kv_indptr[i]..kv_indptr[i+1]]
 * read from the paged buffers.  lse (fp32 [T, Hq], natural log) optional. */
typedef struct rx_extend_params {
  const void* q;        /* [T, Hq, Dk] */
  const void* k_extend; /* [T, Hkv, Dk] */
  const void* v_extend; /* [T, Hkv, Dv] */
  void* o;              /* [T, Hq, Dv] */
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t,
      o_stride_h;
  rx_kv_layout kv;
  const void* qo_indptr; /* [bs+1] */
  int32_t qo_indptr_is_i64;
  const int32_t* kv_indptr; /* [bs+1] */
  const void* kv_indices;
  int32_t kv_indices_is_i64;
  float* lse; /* or NULL */
  int64_t lse_stride_t, lse_stride_h;
  int32_t bs, max_extend_len, num_q_heads, num_kv_heads, head_dim, v_head_dim;
  float sm_scale, k_scale, v_scale, logit_cap;
  int32_t is_causal, skip_prefix, skip_extend, sliding_window_size; /* window <= 0: off */
  const float* sinks;
  int32_t dtype;
  /* speculative-decoding tree mask (extend_attention.py:320-326,378-390,525-539; built by
   * TritonAttnBackend.init_forward_metadata for TARGET_VERIFY / DRAFT_EXTEND, triton_backend.py:845-866):
   * request i owns the bytes custom_mask[mask_indptr[i] ...], a row-major
   * [E_i, woff_i + P_i + E_i] 0/1 matrix (woff_i = window_kv_offsets[i], 0 when NULL).  In the
   * extend part the mask REPLACES the causal triangle; the prefix part is masked only when
   * skip_prefix_custom_mask == 0.  NULL = no mask. */
  const uint8_t* custom_mask;
  const int64_t* mask_indptr; /* int64[bs+1] */
  int32_t skip_prefix_custom_mask;
  const int32_t* window_kv_offsets; /* int32[bs] or NULL */
  /* Grok temperature (:336-343): L > 0 multiplies the scores of the query at absolute position
   * a = P_i + m by log2(a) / log2(L) when a > L (after scale and cap); <= 0 = off */
  int32_t xai_temperature_len;
  /* K8, the one-stage "unified" form used for deterministic inference (extend_attention_fwd_unified,
   * extend_attention.py:1160-1300; _fwd_kernel_unified :852-1158): non-NULL unified_prefix_lens (int32[bs])
   * means kv_indptr / kv_indices list prefix AND new tokens (already stored in the pool), k_extend /
   * v_extend are ignored, and query m of request i sees list position n iff n < prefix_i or
   * n - prefix_i <= m (causal); window: prefix_i + m <= n + W; a custom mask row is [kv_len_i] wide and
   * replaces the causal rule; xai factor = L / (prefix_i + m + 1) once prefix_i + m >= L (:940-946). */
  const int32_t* unified_prefix_lens;
  /* launch-shape hint, 0 = unknown: mean number of kv_indices entries per request (the host knows the
   * tensor's length; the per-request lengths live on the device).  Short work per (request, head, query
   * block) runs better as two 128-query workgroups per CU than as one 256-query workgroup. */
  int32_t avg_kv_len_hint;
  /* GQA-packed query rows (0 / 1 = off; q_pack = Hq / Hkv; D = 128 kernel only, not with unified_prefix_lens).
   * Same tensors and index arrays as without it.  The kernel then runs one workgroup "head" per KV head whose
   * query rows are (new token, q head of the group) pairs, row = token * G + g: a request with few new tokens
   * (speculative verify, short chunks) fills the 32-row query blocks and stages its K/V tiles once per kv head
   * instead of once per q head.  Bit-identical results. */
  int32_t q_pack;
  /* ---- relative-position score bias, round 5 (ABI 14): score_mod = relative_bias_score_mod, aux_tensors = [rel_logits]
   * (score_mod.py:44-56; call sites extend_attention.py:463-476 prefix stage, :594-607 extend stage, :1093-1104
   * unified).  score_bias [T, Hq, score_bias_len] (fp32 or the call's 16-bit dtype, last dim contiguous): the score of
   * query token t = qo_indptr[i] + m, head h against a key gets + score_bias[t, h, r], r = q_pos - kv_pos, when
   * 0 <= r < score_bias_len (after scale, cap and temperature; masked scores stay -inf).  q_pos = P_i + m; kv_pos = the
   * list position n for prefix keys and P_i + n for new tokens (unified form: q_pos = prefix_i + m, kv_pos = n).
   * D = 128 kernel (tiles out of the bias's reach keep the pipelined body) and the generic kernel (other head dims);
   * not with q_pack.  NULL: off. */
  const void* score_bias;
  int32_t score_bias_is_f32;
  int32_t score_bias_len;
  int64_t score_bias_stride_t, score_bias_stride_h;
} rx_extend_params;

int rx_extend_attn(const rx_extend_params* p /* HOST */, void* stream);

/* ---- fused rotary embedding + KV store ---------------------------------------------------------
 * q [n, Hq, D] and k [n, Hkv, D] are rotated IN PLACE (positions int64[n], cos_sin_cache fp32
 * [max_pos, rotary_dim] = [cos(rot/2) | sin(rot/2)], neox or gptj pairing, fp32 math, one rounding) and,
 * when lay != NULL, the rotated k rows and the v rows are written to the pool at loc in the same launch
 * (16-bit pool, or kv_fp8 = 1 with the quant-on-write of rx_store_kv_fp8).  Replaces
 * RotaryEmbedding.forward (srt/layers/rotary_embedding/base.py) + set_kv_buffer
 * (srt/mem_cache/memory_pool.py:2305-2381); the reference's fused forms are
 * kernels/ops/kvcache/rope_cache.py:101-… and kernels/jit/csrc/elementwise/rope.cuh.  Strides in elements. */
int rx_rope_store_kv(void* q, void* k, const void* v, int64_t q_stride_t, int64_t q_stride_h,
                     int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                     int64_t n, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim,
                     int rotary_dim, const int64_t* positions, const float* cos_sin_cache,
                     int64_t cos_sin_stride, int is_neox, const rx_kv_layout* lay /* HOST, or NULL */,
                     const void* loc, int loc_is_i64, int64_t size_limit, int64_t skip_index, float k_scale,
                     float v_scale, int dtype, int32_t* err_flag, void* stream);

/* ---- fused per-head RMSNorm of q and k + rotary embedding (+ KV store) (ABI v11) ---------------------
 * fused_qk_norm_rope (kernels/ops/attention/fused_qknorm_rope.py:37-100, :127-186; kernel
 * kernels/jit/csrc/elementwise/fused_qknorm_rope.cuh:78-246, frequencies :42-63; the reference's own check of it against
 * RMSNorm + RotaryEmbedding: kernels/aot/tests/test_fused_qk_norm_rope.py:31-128): what a QK-norm model runs on its
 * qkv projection in front of attention.  Per (token, head), IN PLACE on q [n, Hq, D] and k [n, Hkv, D] (views of one
 * qkv tensor are fine: strides in elements; v is never written):
 *   x <- x * rsqrt(mean(x^2) + eps) * w         w = q_weight / k_weight [D], the call's 16-bit dtype, fp32 math
 *   RoPE on the first rotary_dim columns (neox: pairs (p, p + rot/2); else (2p, 2p + 1)), times attention_factor,
 *   angle = position * freq_p with freq_p = base^(-2 p / rotary_dim) computed on the fly -- under YaRN (factor != 1)
 *   blended with freq_p / factor by the ramp clamp((p - low) / (high - low), 0, 1) as the reference kernel does -- or,
 *   when cos_sin_cache != NULL, read from the fp32 cache [max_pos, rotary_dim] = [cos | sin] (base / factor / low /
 *   high then unused);  ONE rounding to the 16-bit dtype at the end.
 * positions int32 (the reference's) or int64.  With lay != NULL the finished k rows and the v rows also go to the pool
 * at loc in the same launch, exactly as rx_rope_store_kv does (16-bit pool or fp8 quant-on-write). */
int rx_qknorm_rope_store_kv(void* q, void* k, const void* v, int64_t q_stride_t, int64_t q_stride_h,
                            int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                            int64_t n, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim,
                            int rotary_dim, const void* q_weight, const void* k_weight, float eps,
                            const void* positions, int positions_is_i64, float base, float factor, float low,
                            float high, float attention_factor, const float* cos_sin_cache /* or NULL */,
                            int64_t cos_sin_stride, int is_neox, const rx_kv_layout* lay /* HOST, or NULL */,
                            const void* loc, int loc_is_i64, int64_t size_limit, int64_t skip_index, float k_scale,
                            float v_scale, int dtype, int32_t* err_flag, void* stream);

/* ---- merge of two partial attention states ----------------------------------------------------
 * merge_state_triton (kernels/ops/attention/merge_state.py:8-96; CUDA twin
 * kernels/aot/csrc/attention/merge_attn_states.cu): the building block of prefix-cascade / chunked
 * prefix attention.  For every (token, head):
 *   m = max(lse_a, lse_b); w_a = exp(lse_a - m); w_b = exp(lse_b - m)
 *   out = (a * w_a + b * w_b) / (w_a + w_b);   out_lse = log(w_a + w_b) + m   (optional)
 * A +inf LSE is read as -inf (an empty partial), as the reference does.  a, b, out are dense
 * [num_tokens, num_heads, head_size] of dtype (rx_dtype), LSEs fp32 [num_tokens, num_heads];
 * head_size is a multiple of 8.  out may alias a or b. */
int rx_merge_state(const void* a, const float* lse_a, const void* b, const float* lse_b, void* out,
                   float* out_lse /* or NULL */, int64_t num_tokens, int num_heads, int head_size,
                   int dtype, void* stream);

/* n-way merge for a split pass (speculative verify over a long cached sequence, split into chunks so that a small
 * batch still fills the chip -- the job of kernels/ops/attention/verify_splitkv.py in the reference).  Group g
 * (one request) has num_chunks partials o_chunks [groups, num_chunks, rows_per_group, heads, head_size] with LSEs
 * [groups, num_chunks, rows_per_group, heads], plus optionally one more partial o_last [groups, rows_per_group,
 * heads, head_size] / lse_last (the new tokens' own block).  out [groups, rows_per_group, heads, head_size],
 * out_lse optional.  Same weights as rx_merge_state; empty partials (lse = -inf or +inf) are skipped. */
/* chunk boundaries for such a split pass, as an indptr into the SAME kv_indices: out[b * num_chunks + x] =
 * kv_indptr[b] + min(x * per_b, P_b) with P_b = kv_indptr[b+1] - kv_indptr[b] and per_b = ceil(P_b / num_chunks)
 * rounded up to chunk_align tokens; out[bs * num_chunks] = kv_indptr[bs].  int32[bs * num_chunks + 1]. */
int rx_chunk_indptr(const int32_t* kv_indptr, int bs, int num_chunks, int chunk_align, int32_t* out, void* stream);

int rx_merge_chunks(const void* o_chunks, const float* lse_chunks, int num_chunks, const void* o_last,
                    const float* lse_last, void* out, float* out_lse, int64_t groups, int rows_per_group,
                    int num_heads, int head_size, int dtype, void* stream);

/* ---- decode context parallel (DCP) ------------------------------------------------------------------
 * SURVEY 8e "alternative shardings".  The KV of one request is spread over the dcp_size ranks of a group by the owner
 * rule  position % dcp_size == dcp_rank;  a rank's pool holds virtual slot v at local slot v / dcp_size.  Each rank
 * attends its own tokens with the group's gathered q heads; the partial results are joined by their LSEs
 * (TritonAttnBackend.forward_decode, triton_backend.py:1797-1839; _forward_extend_dcp :1439-1569).  The exchanges are
 * the caller's (torch.distributed); these are the index math and the fp32 LSE arithmetic either side of them.
 *
 * rx_dcp_kv_indices = get_dcp_lens (srt/layers/dcp/layout.py:23-41) + the cumsum + create_triton_kv_indices_for_dcp_triton
 * (kernels/ops/attention/dcp_kernels.py:34-76), i.e. TritonAttnBackend._dcp_kv_indices (triton_backend.py:356-384):
 *   first_i = start_i + (dcp_rank - start_i) mod dcp_size;   n_i = max(0, ceil((start_i + lens_i - first_i) / dcp_size))
 *   kv_indptr[0] = 0, kv_indptr[i+1] = kv_indptr[i] + n_i;   dcp_lens_out[i] = n_i   (optional)
 *   kv_indices[kv_indptr[i] + j] = req_to_token[req_pool_indices[i]][first_i + j * dcp_size] / dcp_size
 * kv_start (int32[bs]) may be NULL (= 0). */
int rx_dcp_kv_indices(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices,
                      int pool_idx_is_i64, const void* lens, int lens_is_i64, const int32_t* kv_start, int dcp_size,
                      int dcp_rank, int32_t* kv_indptr_out, void* kv_indices_out, int out_is_i64, int32_t* dcp_lens_out,
                      int bs, void* stream);
/* Write locations of new tokens under DCP (TritonAttnBackend._set_kv_buffer, triton_backend.py:1227-1239 +
 * masked_set_kv_buffer_kernel, memory_pool.py:4609-4650): loc_out[i] = out_cache_loc[i] / dcp_size when
 * positions[i] % dcp_size == dcp_rank, else skip_index (the value rx_store_kv* is told to leave unwritten). */
int rx_dcp_store_loc(const void* out_cache_loc, int loc_is_i64, const void* positions, int pos_is_i64, int64_t n,
                     int dcp_size, int dcp_rank, int64_t skip_index, int64_t* loc_out, void* stream);
/* One rank's kv-split partials (rx_decode_attn stages = 1; attn_lse filled with -inf beforehand, as
 * triton_backend.py:1816 does) -> its normalised fp32 output [rows, head_size] and natural-log LSE [rows]
 * (rows = bs * heads): o_for_decode / local_lse of triton_backend.py:1806-1837.  A row with no live split gets 0 / -inf.
 * v_scale: the V descale stage 2 would have applied (stage-1 partials carry none). */
int rx_dcp_local_merge(const float* attn_logits, const float* attn_lse, int64_t rows, int num_splits, int head_size,
                       float v_scale, float* o32, float* lse_out, void* stream);
/* 16-bit partial (an extend kernel's output) -> fp32 for the same exchange; n elements, a multiple of 4 */
int rx_dcp_widen(const void* in, int64_t n, int dtype, float* out, void* stream);
/* cp_lse_ag_out_rs_mha (srt/layers/dcp/comm.py:82-108), the part before the all-reduce: with lses_all the all-gathered
 * fp32 [dcp_size, rows], o32[row] *= exp(lses_all[dcp_rank][row] - logsumexp_r lses_all[r][row]), NaN / inf in either
 * factor -> 0; global_lse [rows] (optional) receives the logsumexp. */
int rx_dcp_scale(float* o32, const float* lses_all, int64_t rows, int dcp_size, int dcp_rank, int head_size,
                 float* global_lse, void* stream);
/* ... and the part after it: this rank's heads [head_start, head_start + heads_local) of the summed fp32
 * [num_tokens, heads_all, head_size] -> out [num_tokens, heads_local, head_size] of dtype.  With cur_o / cur_lse (the
 * extend path's own-chunk partial, dtype [num_tokens, heads_local, head_size] + fp32 [num_tokens, heads_local]) the two
 * are first joined by their LSEs, global_lse [num_tokens, heads_all] being the prefix part's
 * (triton_backend.py:1560-1569). */
int rx_dcp_finish(const float* o32, const float* global_lse, const void* cur_o, const float* cur_lse, void* out,
                  int64_t num_tokens, int heads_all, int head_start, int heads_local, int head_size, int dtype,
                  void* stream);

/* ---- K9: paged slot allocation -----------------------------------------------------------
 * alloc_extend_kernel / alloc_decode_kernel (kernels/ops/memory/allocator.py:16-135), called
 * by PagedTokenToKVPoolAllocator.alloc_extend / alloc_decode (allocator/paged.py:172-259).
 * All index tensors int64.  out_indices has sum(seq_lens - prefix_lens) (extend) / bs (decode)
 * entries.  The caller advances its free-page list afterwards, as the reference does. */
int rx_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens, const int64_t* last_loc,
                    const int64_t* free_pages, int64_t* out_indices, int bs, int page_size,
                    void* stream);
int rx_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc, const int64_t* free_pages,
                    int64_t* out_indices, int bs, int page_size, void* stream);

/* ---- K11: req_to_token row write -----------------------------------------------------------
 * write_req_to_token_pool_triton (called srt/mem_cache/allocation.py:75-84): for request i
 *   req_to_token[req_pool_indices[i]][0:pre_lens[i]] = prefix_tensors[i][0:pre_lens[i]]
 *   req_to_token[req_pool_indices[i]][pre_lens[i]:seq_lens[i]] =
 *       out_cache_loc[extend_offset_i : extend_offset_i + seq_lens[i]-pre_lens[i]]
 * prefix_ptrs is a device array of bs device pointers to int64 prefix index lists
 * (entries may be NULL when pre_lens[i] == 0). */
int rx_write_req_to_token(int32_t* req_to_token, int64_t row_stride,
                          const int64_t* req_pool_indices, const int64_t* const* prefix_ptrs,
                          const int64_t* pre_lens, const int64_t* seq_lens,
                          const int64_t* extend_lens, const int64_t* out_cache_loc, int bs,
                          void* stream);

/* ---- a5 / 8f-1: device-resident free list of the slot / page allocators ------------------------------------
 * Replaces the torch-tensor free list of TokenToKVPoolAllocator (srt/mem_cache/allocator/token.py:27-84) and
 * PagedTokenToKVPoolAllocator (allocator/paged.py:105-345; merge_and_sort_free allocator/base.py:70-76) with a
 * ring in HBM that kernels update in place -- same list ORDER after every operation (it decides the KV page indices
 * a request gets, which must be bit-exact), no torch.cat / torch.unique, no host sync.
 * All pointers are device pointers owned by the caller:
 *   free_ring / release_ring  int64[capacity]   the two lists (release_ring may be NULL without need_sort)
 *   flags                     uint8[num_ids+1]  zero between calls; marks ids during sorted inserts
 *   tile_scratch              int64[rx_pool_tile_scratch_len(num_ids)]
 *   state                     int64[rx_pool_state_words()]: [0] free head, [1] free count, [2] release head,
 *                             [3] release count, [4] refused allocations (count check failed on the device)
 * capacity >= the largest number of ids a list can hold.  `which`: 0 = free list, 1 = release list. */
typedef struct rx_pool_desc {
  int64_t* free_ring;
  int64_t* release_ring;
  int64_t capacity;
  uint8_t* flags;
  int64_t num_ids;
  int64_t* tile_scratch;
  int64_t* state;
} rx_pool_desc;

int64_t rx_pool_tile_scratch_len(int64_t num_ids);
int rx_pool_state_words(void);
/* clear(): free = [first_id, first_id + n), release empty (token.py:42-49, paged.py:329-337) */
int rx_pool_reset(const rx_pool_desc* d, int64_t first_id, int64_t n, void* stream);
/* list := ids (restore / tests);  out[0..min(count, out_cap)) := list in order */
int rx_pool_load(const rx_pool_desc* d, int which, const int64_t* ids, int64_t n, void* stream);
int rx_pool_snapshot(const rx_pool_desc* d, int which, int64_t* out, int64_t out_cap, void* stream);
/* alloc (token.py:55-64, paged.py:149-170): num_pages ids off the head, expanded to page_size slots each */
int rx_pool_alloc(const rx_pool_desc* d, int64_t num_pages, int page_size, int64_t* out, void* stream);
/* alloc_extend / alloc_decode (paged.py:172-259; kernels/ops/memory/allocator.py:16-135) against the ring;
 * num_new_pages (known on the host from the CPU lens) ids leave the head afterwards */
int rx_pool_alloc_extend(const rx_pool_desc* d, const int64_t* prefix_lens, const int64_t* seq_lens,
                         const int64_t* last_loc, int64_t* out_indices, int bs, int page_size,
                         int64_t num_new_pages, void* stream);
int rx_pool_alloc_decode(const rx_pool_desc* d, const int64_t* seq_lens, const int64_t* last_loc,
                         int64_t* out_indices, int bs, int page_size, int64_t num_new_pages, void* stream);
/* alloc_for_decode as ONE launch (srt/mem_cache/allocation.py:539-593): for request i with seq_lens[i] tokens so far
 * (int64, BEFORE the new one) and row req_pool_indices[i] of req_to_token (int32, row_stride elements apart):
 * loc = row[seq-1] + 1 inside a page, or the first slot of the next free page when seq % page_size == 0 (page_size
 * 1: always) -- alloc_decode_kernel's rule (allocator.py:98-135); out_indices[i] = loc and row[seq] = loc.
 * num_new_pages = #{i : seq_lens[i] % page_size == 0}, known on the host. */
int rx_pool_alloc_decode_rows(const rx_pool_desc* d, int32_t* req_to_token, int64_t row_stride,
                              const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t* out_indices, int bs,
                              int page_size, int64_t num_new_pages, void* stream);
/* alloc_for_extend as ONE launch (srt/mem_cache/allocation.py:303-403: last_loc from the cached prefixes, alloc_extend
 * (kernels/ops/memory/allocator.py:16-95; page_size 1: alloc_token_slots, one fresh id per token) and
 * write_cache_indices :55-101).  table: DEVICE int64 [4, bs], rows = req_pool_idx | prefix_len | seq_len | device address of
 * the request's cached prefix slots (int64[prefix_len]; 0 when prefix_len == 0) -- one packed host table, one H2D copy.
 * out_indices int64[sum(seq - prefix)] (the batch's out_cache_loc); req_to_token row i gets the prefix slots at
 * [0, prefix_len) and the new slots at [prefix_len, seq_len).  num_new_pages as rx_pool_alloc_extend's. */
int rx_pool_alloc_extend_rows(const rx_pool_desc* d, int32_t* req_to_token, int64_t row_stride, const int64_t* table,
                              int64_t* out_indices, int bs, int page_size, int64_t num_new_pages, void* stream);
/* list := list + ids (token.py:66-76) */
int rx_pool_append(const rx_pool_desc* d, int which, const int64_t* ids, int64_t n, void* stream);
/* list := reps + list, reps = ([idx[0]] if has_first) + idx[start::stride], each / page_size  (free_segment's
 * stride-slice page representatives, paged.py:273-301) */
int rx_pool_prepend_strided(const rx_pool_desc* d, int which, const int64_t* idx, int64_t n_idx, int has_first,
                            int64_t start, int64_t stride, int page_size, void* stream);
/* free (paged.py:261-271): mark idx / page_size (several calls may accumulate: free_group), then
 * list := sorted(unique(marked)) + list -- what torch.unique + cat compute, without the host sync */
int rx_pool_mark(const rx_pool_desc* d, const int64_t* idx, int64_t n, int page_size, void* stream);
int rx_pool_flush_marks(const rx_pool_desc* d, int which, void* stream);
/* merge_and_sort_free (base.py:70-76): free := sort(free + release), release := empty */
int rx_pool_merge_sort(const rx_pool_desc* d, void* stream);

/* ---- K10: KV move (all layers) ---------------------------------------------------------------
 * copy_all_layer_kv_cache_tiled (kernels/ops/kvcache/cache_move.py:60-133) used by
 * MHATokenToKVPool.move_kv_cache (memory_pool.py:2775-2842):
 *   for every buffer b: buf_b[tgt_loc[i]] = buf_b[src_loc[i]]   (row_bytes[b] bytes each)
 * data_ptrs: device uint64[num_bufs] base addresses; row_bytes: device int64[num_bufs]
 * (the pool's data_ptrs / data_strides tables, memory_pool.py:2005-2028). */
int rx_move_kv(const uint64_t* data_ptrs, const int64_t* row_bytes, int num_bufs,
               const int64_t* tgt_loc, const int64_t* src_loc, int64_t n, void* stream);

/* The same move on a paged pool layout (the HND pools [pages, Hkv, page, D] of memory_pool.py:2032-2036, where a
 * slot's row is num_heads pieces): slot s lives at page s / page_size, offset s % page_size.
 * geom: device int64[num_bufs][4] = {page_stride, head_stride, tok_stride, piece_bytes} per buffer, in BYTES
 * (piece_bytes = head_dim * element size; a multiple of 4). */
int rx_move_kv_layout(const uint64_t* data_ptrs, const int64_t* geom, int num_bufs, int page_size, int num_heads,
                      const int64_t* tgt_loc, const int64_t* src_loc, int64_t n, void* stream);

/* ---- a16: native radix tree of cached KV prefixes (HOST side; no GPU work) -----------------------
 * RadixCache (srt/mem_cache/radix_cache.py:279-812): match_prefix :352-410, insert :412-432 /
 * :704-757, _split_node :674-694, inc/dec_lock_ref :592-626, evict :562-590, eviction policies
 * srt/mem_cache/evict_policy.py.  All pointers below are HOST pointers.  Token ids and KV slot
 * ids are int64; lengths are silently truncated to a multiple of page_size as the reference does.
 * Nodes are addressed by id; rx_radix_root() is the id of the root (never evicted).
 * eviction_policy: 0 lru, 1 lfu, 2 fifo, 3 mru, 4 filo, 5 priority, 6 slru. */
typedef struct rx_radix rx_radix;
rx_radix* rx_radix_create(int page_size, int eviction_policy);
void rx_radix_destroy(rx_radix* t);
void rx_radix_reset(rx_radix* t);
int64_t rx_radix_root(const rx_radix* t);
/* returns the matched length (writes that many slot ids to out_indices, which may be NULL);
 * -1 if cap is too small.  *last_node = deepest matched node (the root when nothing matched). */
int64_t rx_radix_match_prefix(rx_radix* t, const int64_t* token_ids, int64_t n, const char* extra_key,
                              int64_t* out_indices, int64_t cap, int64_t* last_node);
/* returns the length of the prefix that was already cached. */
int64_t rx_radix_insert(rx_radix* t, const int64_t* token_ids, const int64_t* values, int64_t n,
                        const char* extra_key, int64_t priority, int chunked, int64_t* last_node);
/* return the change of evictable size (INT64_MIN for an unknown node id). */
int64_t rx_radix_inc_lock_ref(rx_radix* t, int64_t node_id);
int64_t rx_radix_dec_lock_ref(rx_radix* t, int64_t node_id);
/* evicts leaves in policy order until >= num_tokens slots are freed; slot ids are written node by
 * node to out_slots and each node's count to out_seg_lens; returns the number of slots freed. */
int64_t rx_radix_evict(rx_radix* t, int64_t num_tokens, int64_t* out_slots, int64_t slot_cap,
                       int64_t* out_seg_lens, int64_t seg_cap, int64_t* num_segments);
int64_t rx_radix_evictable_size(const rx_radix* t);
int64_t rx_radix_protected_size(const rx_radix* t);
int64_t rx_radix_total_size(const rx_radix* t);
int64_t rx_radix_num_nodes(const rx_radix* t);
/* info6 = {parent id, key length, lock_ref, hit_count, #children, priority}; -1 if unknown id. */
int rx_radix_node_info(const rx_radix* t, int64_t node_id, int64_t* info6);

/* One request's cache bookkeeping in one call (RadixCache.cache_finished_req radix_cache.py:434-486 and
 * cache_unfinished_req :488-553): insert the request's page-aligned key with the slots of its req_to_token row,
 * move the locks, and report which row ranges go back to the allocator.
 * token_ids / slots: host int64[n] (slots = the row's first n entries).  flags: bit 0 finished, bit 1 insert
 * (finished only), bit 2 chunked (unfinished only).  last_node: node the request holds a lock on, < 0 for none.
 * out8 (host int64[8]): [0],[1] first range to free [begin, end) (free_segment with start_pos = begin);
 * [2],[3] second range (finished: the unaligned tail [key_len, n)); [4] page-aligned key length; [5] unfinished:
 * slots written to out_slots (the row's cached prefix as the tree now holds it); [6] unfinished: node now locked;
 * [7] prefix length the insert found cached.  Returns 0, -1 on bad arguments / out_slots too small. */
int rx_radix_cache_req(rx_radix* t, const int64_t* token_ids, const int64_t* slots, int64_t n, const char* extra_key,
                       int priority, int flags, int64_t protected_len, int64_t last_node, int64_t* out_slots,
                       int64_t out_cap, int64_t* out8);

/* ---- C1: peer-to-peer all-reduce over xGMI (one process per GPU) ---------------------------------
 * The sum all-reduce behind RowParallelLinear.forward (srt/layers/linear.py:1606-1627 ->
 * GroupCoordinator.all_reduce, srt/distributed/parallel_state.py:622-732, whose small-message path is a
 * custom IPC all-reduce).  Two-shot direct over the full mesh; see csrc/rx_allreduce.hip.
 *
 * Setup (host, once): every rank allocates one region of rx_ar_region_bytes(max_bytes) with
 * rx_ar_alloc_region (uncached device memory, zeroed), exports it with rx_ipc_get_handle, the 64-byte
 * handles are exchanged out of band (torch.distributed all_gather in sglang_amd/parallel.py), peers
 * are mapped with rx_ipc_open_handle, and rx_ar_init receives the `world` region pointers (own region =
 * the local pointer).  dev_err is a caller-owned, zeroed int32 device word (RX_DEVERR_AR_TIMEOUT).
 * rx_allreduce is stream ordered, allocation free and must be called in the same order with the same
 * count on every rank; in == out is allowed.  count is in elements (multiple of 8), dtype an rx_dtype. */
typedef struct rx_ar_ctx rx_ar_ctx;
int64_t rx_ar_region_bytes(int64_t max_bytes);
int rx_ar_alloc_region(int64_t bytes, void** dev_ptr_out);
int rx_ar_free_region(void* dev_ptr);
int rx_ipc_get_handle(void* dev_ptr, void* handle_out_64b);
int rx_ipc_open_handle(const void* handle_64b, void** dev_ptr_out);
int rx_ipc_close_handle(void* dev_ptr);
int rx_ar_init(rx_ar_ctx** ctx_out, int rank, int world, void* const* peer_regions, int64_t max_bytes,
               int32_t* dev_err);
int rx_allreduce(rx_ar_ctx* ctx, const void* in, void* out, int64_t count, int dtype, void* stream);
/* The deterministic form (round 6): the reference's AMD path under --enable-deterministic-inference forces its ONE-stage
 * kernel (device_communicators/custom_all_reduce.py:294-301; kernels/aot/csrc/allreduce/deterministic_all_reduce.hip:1-14:
 * every GPU reads all peers' data and reduces it locally in a fixed order).  Same arguments, context and stream rules as
 * rx_allreduce (the two may be mixed on one context); one flag exchange; every rank sums ALL elements itself in fp32 in rank
 * order 0 .. world-1 and rounds once, so an element's bits depend on the world's inputs for THAT element only -- not on the
 * message size, the batch it is embedded in, the rank or the run. */
int rx_allreduce_det(rx_ar_ctx* ctx, const void* in, void* out, int64_t count, int dtype, void* stream);

/* Fused all-reduce + residual add + RMSNorm (GroupCoordinator.fused_allreduce_rmsnorm, srt/distributed/
 * parallel_state.py:748-878; the split path it replaces: benchmark/kernels/all_reduce/benchmark_fused_ar_rms_amd.py
 * :171-182):
 *   residual_out = all_reduce_sum(in) + residual_in          (both rounded to the 16-bit dtype, as the split path)
 *   out          = residual_out * rsqrt(mean(residual_out^2) + eps) * weight     (fp32 math, one rounding)
 * in / residual_in / out / residual_out: [rows, hidden] contiguous, weight [hidden], all of `dtype`; hidden a multiple
 * of 8 and <= 16384; rows * hidden * 2 <= the context's max_bytes.  residual_out may alias residual_in (in-place
 * residual stream); out must be a buffer of its own.  Same context / stream rules as rx_allreduce. */
int rx_allreduce_rmsnorm(rx_ar_ctx* ctx, const void* in, const void* residual_in, const void* weight, void* out,
                         void* residual_out, int64_t rows, int64_t hidden, float eps, int dtype, void* stream);
int rx_ar_destroy(rx_ar_ctx* ctx);

/* ---- C3: the "quick" all-reduce -- large 16-bit messages over a block-scaled integer wire format (round 6) --------------
 * QuickAllReduce (srt/distributed/device_communicators/quick_all_reduce.py:41-267), which GroupCoordinator.all_reduce takes
 * behind the custom all-reduce and ahead of NCCL (parallel_state.py:886-948); kernel kernels/aot/csrc/allreduce/
 * quick_all_reduce.cuh:52-632 (ops init_custom_qr / qr_get_handle / qr_open_handles / qr_all_reduce / qr_destroy,
 * quick_all_reduce.cu:10-89).  Two-shot, pushed: every rank encodes its input in groups of 64 elements (two 32-element scale
 * blocks: even / odd elements) and writes segment r to rank r, which decodes the `world` versions, adds them in rank order in
 * the 16-bit type, encodes the sum and writes it to everybody.  quant_level: RX_QR_FP moves the 16-bit values (the result is
 * the rank-order sum in the tensor's type), RX_QR_INT8 / INT6 / INT4 move 8 / 6 / 4-bit codes + one 16-bit scale per block.
 * cast_bf16_to_fp16 != 0 (the reference's default, ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16): bf16 tensors are converted to fp16 on
 * the way in, reduced by the fp16 arithmetic and converted back.  The arithmetic is the reference's operation for operation
 * (csrc/rx_quick_allreduce.hip header; CPU restatement: quick_allreduce under oracle/); the slot / flag protocol is this
 * library's own: a FIXED region of rx_qr_region_bytes() whatever the message size, no per-call host state, HIP-graph safe.
 *
 * Setup as for rx_allreduce: a region of rx_qr_region_bytes() from rx_ar_alloc_region on every rank, handles exchanged and
 * mapped with rx_ipc_*, rx_qr_init with the `world` (2, 4 or 8) region pointers.  rx_quick_allreduce is stream ordered and
 * allocation free; same order, count, dtype and level on every rank; in == out allowed; count a multiple of 8 elements, any
 * size (the tail of the last 64-element group is taken as zeros, as the reference's bounded buffer loads do). */
typedef struct rx_qr_ctx rx_qr_ctx;
enum rx_qr_level { RX_QR_FP = 0, RX_QR_INT8 = 1, RX_QR_INT6 = 2, RX_QR_INT4 = 3 }; /* QuickReduceRegime, quick_all_reduce.py:41-46 */
int64_t rx_qr_region_bytes(void);
int rx_qr_init(rx_qr_ctx** ctx_out, int rank, int world, void* const* peer_regions, int32_t* dev_err);
int rx_quick_allreduce(rx_qr_ctx* ctx, const void* in, void* out, int64_t count, int dtype, int quant_level,
                       int cast_bf16_to_fp16, void* stream);
int rx_qr_destroy(rx_qr_ctx* ctx);
/* Test support: v_rcp_f16 of all 65536 fp16 bit patterns into dev_out_65536 (the fp16 codecs' encode scale is that
 * instruction's output, specified to 1 ulp; the oracle takes the table rather than assuming a rounding). */
int rx_rcp_f16_table(uint16_t* dev_out_65536, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RADIX_HIP_H_ */
