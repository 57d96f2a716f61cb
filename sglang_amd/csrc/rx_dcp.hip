// Decode context parallel (DCP), SURVEY 8e "alternative shardings": the KV of ONE request is spread over the ranks of
// a group by the owner rule  position % dcp_size == dcp_rank  (slot held locally = virtual slot / dcp_size); every rank
// attends its own tokens with the group's gathered q heads and the partial results are joined by their LSEs.
// Reference: srt/layers/dcp/layout.py:23-41 (get_dcp_lens), kernels/ops/attention/dcp_kernels.py:34-76
// (create_triton_kv_indices_for_dcp_triton), memory_pool.py:4609-4650 (masked_set_kv_buffer_kernel),
// srt/layers/dcp/comm.py:82-108 (cp_lse_ag_out_rs_mha), triton_backend.py:1797-1839 / 1439-1569 (the callers).
// The exchanges themselves (all-gather of q heads and LSEs, all-reduce of the scaled outputs) are torch.distributed
// calls in attention/dcp.py; these kernels are the index math and the fp32 LSE arithmetic either side of them.
#include "rx_common.h"

namespace rx {

// tokens of [start, start + len) owned by `rank`, and the first of them (layout.py:23-41)
__device__ __forceinline__ void dcp_range(int32_t start, int32_t len, int dcp, int rank, int32_t& first, int32_t& n) {
  int32_t m = (rank - start) % dcp;
  if (m < 0) m += dcp;
  first = start + m;
  const int32_t remaining = start + len - first;
  n = remaining > 0 ? (remaining + dcp - 1) / dcp : 0;
}

// one block: per-rank lengths, then their inclusive scan into kv_indptr (as indptr_scan_kernel, rx_misc.hip)
__global__ __launch_bounds__(1024) void dcp_scan_kernel(const void* __restrict__ lens, int lens64,
                                                        const int32_t* __restrict__ kv_start, int bs, int dcp, int rank,
                                                        int32_t* __restrict__ indptr, int32_t* __restrict__ dcp_lens) {
  __shared__ int32_t wave_sums[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
    carry_s = 0;
    indptr[0] = 0;
  }
  __syncthreads();
  for (int base = 0; base < bs; base += 1024) {
    const int i = base + tid;
    int32_t x = 0;
    if (i < bs) {
      int32_t first;
      dcp_range(kv_start ? kv_start[i] : 0, static_cast<int32_t>(load_idx(lens, i, lens64)), dcp, rank, first, x);
      if (dcp_lens) dcp_lens[i] = x;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wave_sums[wid] = x;
    __syncthreads();
    int32_t prefix = carry_s;
    for (int w = 0; w < wid; ++w) prefix += wave_sums[w];
    if (i < bs) indptr[i + 1] = prefix + x;
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void dcp_gather_kernel(const int32_t* __restrict__ req_to_token, int64_t row_stride,
                                                         const void* __restrict__ req_pool_indices, int pool64,
                                                         const int32_t* __restrict__ kv_start, int dcp, int rank,
                                                         const int32_t* __restrict__ kv_indptr, OutT* __restrict__ out) {
  const int b = blockIdx.y;
  const int32_t beg = kv_indptr[b];
  const int32_t len = kv_indptr[b + 1] - beg;
  const int32_t chunk0 = blockIdx.x * 1024;
  if (chunk0 >= len) return;
  const int32_t start = kv_start ? kv_start[b] : 0;
  int32_t m = (rank - start) % dcp;
  if (m < 0) m += dcp;
  const int64_t req = load_idx(req_pool_indices, b, pool64);
  const int32_t* src = req_to_token + req * row_stride + start + m;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int32_t t = chunk0 + j * 256 + threadIdx.x;
    if (t < len) out[static_cast<int64_t>(beg) + t] = static_cast<OutT>(src[static_cast<int64_t>(t) * dcp] / dcp);
  }
}

// write locations of the new tokens: local slot for the tokens this rank owns, `skip` for the others
__global__ __launch_bounds__(256) void dcp_store_loc_kernel(const void* __restrict__ loc, int loc64,
                                                            const void* __restrict__ positions, int pos64, int64_t n,
                                                            int dcp, int rank, int64_t skip, int64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t pos = load_idx(positions, i, pos64);
  out[i] = (pos % dcp == rank) ? load_idx(loc, i, loc64) / dcp : skip;
}

// kv-split partials of one rank -> its normalised fp32 output and natural-log LSE (decode: o_for_decode / local_lse,
// triton_backend.py:1806-1837).  Splits that did not run carry lse = -inf (the caller fills the buffer first).
__global__ __launch_bounds__(256) void dcp_local_merge_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ lse, int64_t rows, int S, int dv4,
                                                              float v_scale, float* __restrict__ o32,
                                                              float* __restrict__ lse_out) {
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t row = gid / dv4;
  if (row >= rows) return;
  const int d = static_cast<int>(gid % dv4) * 4;
  const float* l = lse + row * S;
  float e_max = -INFINITY;
  for (int s = 0; s < S; ++s) e_max = fmaxf(e_max, l[s]);
  float e_sum = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (e_max > -INFINITY) {
    const float* lp = logits + row * S * (dv4 * 4) + d;
    for (int s = 0; s < S; ++s) {
      if (!(l[s] > -INFINITY)) continue;  // its row is undefined
      const float w = __expf(l[s] - e_max);
      acc += w * *reinterpret_cast<const f32x4*>(lp + static_cast<int64_t>(s) * dv4 * 4);
      e_sum += w;
    }
    acc *= v_scale / e_sum;  // the partials carry no V scale (rx_decode_params.extra_o says the same of its inputs)
  }
  *reinterpret_cast<f32x4*>(o32 + row * (dv4 * 4) + d) = acc;
  if (d == 0) lse_out[row] = e_max > -INFINITY ? e_max + __logf(e_sum) : -INFINITY;
}

// 16-bit partial (the extend kernel's output) -> fp32, for the same exchange
template <typename T>
__global__ __launch_bounds__(256) void dcp_widen_kernel(const uint16_t* __restrict__ in, int64_t n4, float* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n4) return;
  const u32x2 raw = *reinterpret_cast<const u32x2*>(in + 4 * i);
  *reinterpret_cast<f32x4*>(out + 4 * i) =
      f32x4{T::to_f32(static_cast<uint16_t>(raw[0] & 0xffff)), T::to_f32(static_cast<uint16_t>(raw[0] >> 16)),
            T::to_f32(static_cast<uint16_t>(raw[1] & 0xffff)), T::to_f32(static_cast<uint16_t>(raw[1] >> 16))};
}

// cp_lse_ag_out_rs_mha, first half (comm.py:92-98): this rank's output times exp(lse_rank - logsumexp over ranks),
// NaN / inf -> 0 in both factors; lses_all is the all-gathered [dcp, rows].  global_lse [rows] optional.
__global__ __launch_bounds__(256) void dcp_scale_kernel(float* __restrict__ o32, const float* __restrict__ lses_all,
                                                        int64_t rows, int dcp, int rank, int dv4,
                                                        float* __restrict__ global_lse) {
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t row = gid / dv4;
  if (row >= rows) return;
  const int d = static_cast<int>(gid % dv4) * 4;
  float e_max = -INFINITY;
  for (int r = 0; r < dcp; ++r) e_max = fmaxf(e_max, lses_all[r * rows + row]);
  float g = -INFINITY;  // every rank empty
  if (e_max > -INFINITY) {
    float e_sum = 0.f;
    for (int r = 0; r < dcp; ++r) e_sum += __expf(lses_all[r * rows + row] - e_max);
    g = e_max + __logf(e_sum);
  }
  float scale = __expf(lses_all[rank * rows + row] - g);
  if (!(fabsf(scale) < INFINITY)) scale = 0.f;  // NaN (-inf - -inf) and inf -> 0
  f32x4 v = *reinterpret_cast<f32x4*>(o32 + row * (dv4 * 4) + d);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (fabsf(v[i]) < INFINITY) ? v[i] * scale : 0.f;
  *reinterpret_cast<f32x4*>(o32 + row * (dv4 * 4) + d) = v;
  if (global_lse && d == 0) global_lse[row] = g;
}

// second half: this rank's heads [h0, h0 + h_loc) of the all-reduced fp32 sum -> 16-bit output; with a second partial
// (the extend path's current-chunk result, 16-bit + LSE) the two are joined by their LSEs first
// (triton_backend.py:1560-1569).
template <typename T>
__global__ __launch_bounds__(256) void dcp_finish_kernel(const float* __restrict__ o32, const float* __restrict__ glse,
                                                         const uint16_t* __restrict__ cur, const float* __restrict__ cur_lse,
                                                         uint16_t* __restrict__ out, int64_t tokens, int h_all, int h0,
                                                         int h_loc, int dv4) {
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t lrow = gid / dv4;  // (token, local head)
  if (lrow >= tokens * h_loc) return;
  const int d = static_cast<int>(gid % dv4) * 4;
  const int64_t t = lrow / h_loc;
  const int hl = static_cast<int>(lrow % h_loc);
  const int64_t grow = t * h_all + h0 + hl;
  f32x4 v = *reinterpret_cast<const f32x4*>(o32 + grow * (dv4 * 4) + d);
  if (cur) {
    const float pl = glse[grow], cl = cur_lse[lrow];
    const float m = fmaxf(pl, cl);
    float wp = 0.f, wc = 0.f;
    if (m > -INFINITY) {
      const float ep = __expf(pl - m), ec = __expf(cl - m);
      const float inv = 1.0f / (ep + ec);
      wp = ep * inv;
      wc = ec * inv;
    }
    const u32x2 raw = *reinterpret_cast<const u32x2*>(cur + lrow * (dv4 * 4) + d);
    const f32x4 c = {T::to_f32(static_cast<uint16_t>(raw[0] & 0xffff)), T::to_f32(static_cast<uint16_t>(raw[0] >> 16)),
                     T::to_f32(static_cast<uint16_t>(raw[1] & 0xffff)), T::to_f32(static_cast<uint16_t>(raw[1] >> 16))};
    v = v * wp + c * wc;
  }
  u32x2 pk;
  pk[0] = pack2<T>(v[0], v[1]);
  pk[1] = pack2<T>(v[2], v[3]);
  *reinterpret_cast<u32x2*>(out + lrow * (dv4 * 4) + d) = pk;
}

}  // namespace rx

using namespace rx;

extern "C" {

int rx_dcp_kv_indices(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices,
                      int pool_idx_is_i64, const void* lens, int lens_is_i64, const int32_t* kv_start, int dcp_size,
                      int dcp_rank, int32_t* kv_indptr_out, void* kv_indices_out, int out_is_i64, int32_t* dcp_lens_out,
                      int bs, void* stream) {
  RX_REQUIRE(bs >= 0 && dcp_size >= 1 && dcp_rank >= 0 && dcp_rank < dcp_size, "rx_dcp_kv_indices: bs %d, rank %d of %d",
             bs, dcp_rank, dcp_size);
  RX_REQUIRE(kv_indptr_out, "rx_dcp_kv_indices: kv_indptr_out is null");
  auto s = static_cast<hipStream_t>(stream);
  if (bs == 0) {
    (void)hipMemsetAsync(kv_indptr_out, 0, sizeof(int32_t), s);
    return check_launch("rx_dcp_kv_indices(memset)");
  }
  RX_REQUIRE(req_to_token && req_pool_indices && lens, "rx_dcp_kv_indices: null pointer");
  hipLaunchKernelGGL(dcp_scan_kernel, dim3(1), dim3(1024), 0, s, lens, lens_is_i64, kv_start, bs, dcp_size, dcp_rank,
                     kv_indptr_out, dcp_lens_out);
  if (kv_indices_out) {
    const unsigned chunks = static_cast<unsigned>((row_stride / dcp_size + 1 + 1023) / 1024);
    const dim3 grid(chunks ? chunks : 1, bs);
    if (out_is_i64)
      hipLaunchKernelGGL(dcp_gather_kernel<int64_t>, grid, dim3(256), 0, s, req_to_token, row_stride, req_pool_indices,
                         pool_idx_is_i64, kv_start, dcp_size, dcp_rank, kv_indptr_out, (int64_t*)kv_indices_out);
    else
      hipLaunchKernelGGL(dcp_gather_kernel<int32_t>, grid, dim3(256), 0, s, req_to_token, row_stride, req_pool_indices,
                         pool_idx_is_i64, kv_start, dcp_size, dcp_rank, kv_indptr_out, (int32_t*)kv_indices_out);
  }
  return check_launch("rx_dcp_kv_indices");
}

int rx_dcp_store_loc(const void* out_cache_loc, int loc_is_i64, const void* positions, int pos_is_i64, int64_t n,
                     int dcp_size, int dcp_rank, int64_t skip_index, int64_t* loc_out, void* stream) {
  RX_REQUIRE(n >= 0 && dcp_size >= 1 && dcp_rank >= 0 && dcp_rank < dcp_size, "rx_dcp_store_loc: bad sizes");
  if (n == 0) return RX_OK;
  RX_REQUIRE(out_cache_loc && positions && loc_out, "rx_dcp_store_loc: null pointer");
  hipLaunchKernelGGL(dcp_store_loc_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), out_cache_loc, loc_is_i64, positions, pos_is_i64, n, dcp_size,
                     dcp_rank, skip_index, loc_out);
  return check_launch("rx_dcp_store_loc");
}

int rx_dcp_local_merge(const float* attn_logits, const float* attn_lse, int64_t rows, int num_splits, int head_size,
                       float v_scale, float* o32, float* lse_out, void* stream) {
  RX_REQUIRE(rows >= 0 && num_splits >= 1 && head_size > 0 && head_size % 4 == 0, "rx_dcp_local_merge: bad sizes");
  if (rows == 0) return RX_OK;
  RX_REQUIRE(attn_logits && attn_lse && o32 && lse_out, "rx_dcp_local_merge: null pointer");
  RX_REQUIRE((((uintptr_t)attn_logits | (uintptr_t)o32) & 15) == 0, "rx_dcp_local_merge: 16-byte alignment");
  const int dv4 = head_size / 4;
  hipLaunchKernelGGL(dcp_local_merge_kernel, dim3(static_cast<unsigned>((rows * dv4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), attn_logits, attn_lse, rows, num_splits, dv4, v_scale, o32, lse_out);
  return check_launch("rx_dcp_local_merge");
}

int rx_dcp_widen(const void* in, int64_t n, int dtype, float* out, void* stream) {
  RX_REQUIRE(n >= 0 && n % 4 == 0, "rx_dcp_widen: n %lld must be a multiple of 4", (long long)n);
  if (n == 0) return RX_OK;
  RX_REQUIRE(in && out && (((uintptr_t)in & 7) == 0) && (((uintptr_t)out & 15) == 0), "rx_dcp_widen: pointers");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_dcp_widen: dtype %d", dtype);
  const unsigned grid = static_cast<unsigned>((n / 4 + 255) / 256);
  auto s = static_cast<hipStream_t>(stream);
  if (dtype == RX_BF16) hipLaunchKernelGGL(dcp_widen_kernel<BF16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)in, n / 4, out);
  else hipLaunchKernelGGL(dcp_widen_kernel<F16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)in, n / 4, out);
  return check_launch("rx_dcp_widen");
}

int rx_dcp_scale(float* o32, const float* lses_all, int64_t rows, int dcp_size, int dcp_rank, int head_size,
                 float* global_lse, void* stream) {
  RX_REQUIRE(rows >= 0 && dcp_size >= 1 && dcp_rank >= 0 && dcp_rank < dcp_size && head_size > 0 && head_size % 4 == 0,
             "rx_dcp_scale: bad sizes");
  if (rows == 0) return RX_OK;
  RX_REQUIRE(o32 && lses_all && (((uintptr_t)o32 & 15) == 0), "rx_dcp_scale: pointers");
  const int dv4 = head_size / 4;
  hipLaunchKernelGGL(dcp_scale_kernel, dim3(static_cast<unsigned>((rows * dv4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), o32, lses_all, rows, dcp_size, dcp_rank, dv4, global_lse);
  return check_launch("rx_dcp_scale");
}

int rx_dcp_finish(const float* o32, const float* global_lse, const void* cur_o, const float* cur_lse, void* out,
                  int64_t num_tokens, int heads_all, int head_start, int heads_local, int head_size, int dtype,
                  void* stream) {
  RX_REQUIRE(num_tokens >= 0 && heads_all > 0 && heads_local > 0 && head_start >= 0 &&
                 head_start + heads_local <= heads_all && head_size > 0 && head_size % 4 == 0,
             "rx_dcp_finish: bad sizes");
  if (num_tokens == 0) return RX_OK;
  RX_REQUIRE(o32 && out && (!cur_o || (cur_lse && global_lse)), "rx_dcp_finish: null pointer");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_dcp_finish: dtype %d", dtype);
  RX_REQUIRE((((uintptr_t)o32 & 15) | ((uintptr_t)out & 7) | ((uintptr_t)cur_o & 7)) == 0, "rx_dcp_finish: alignment");
  const int dv4 = head_size / 4;
  const unsigned grid = static_cast<unsigned>((num_tokens * heads_local * dv4 + 255) / 256);
  auto s = static_cast<hipStream_t>(stream);
  if (dtype == RX_BF16)
    hipLaunchKernelGGL(dcp_finish_kernel<BF16>, dim3(grid), dim3(256), 0, s, o32, global_lse, (const uint16_t*)cur_o,
                       cur_lse, (uint16_t*)out, num_tokens, heads_all, head_start, heads_local, dv4);
  else
    hipLaunchKernelGGL(dcp_finish_kernel<F16>, dim3(grid), dim3(256), 0, s, o32, global_lse, (const uint16_t*)cur_o,
                       cur_lse, (uint16_t*)out, num_tokens, heads_all, head_start, heads_local, dv4);
  return check_launch("rx_dcp_finish");
}

}  // extern "C"
