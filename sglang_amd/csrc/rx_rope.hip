// Fused rotary embedding + KV-cache store: q and k are rotated in place and the rotated k plus v go
// straight into the paged pool in the same launch (16-bit pools or fp8 e4m3fn quant-on-write).
//
// Reference: the unfused pair RotaryEmbedding.forward (srt/layers/rotary_embedding/base.py: cos_sin_cache
// [max_pos, rot_dim] = [cos | sin], `_apply_rotary_emb` neox / gptj) followed by
// MHATokenToKVPool.set_kv_buffer (srt/mem_cache/memory_pool.py:2305-2381), and its fused forms
// kernels/ops/kvcache/rope_cache.py:101-… (fused_qk_rope_reshape_and_cache) and
// kernels/jit/csrc/elementwise/rope.cuh.  SURVEY.md §8f rank 3: this removes one read + one write of k
// (and a launch) from every layer of a decode step.
//
//   neox : pairs (i, i + rot/2):  o[i] = x[i] cos_i - x[i+rot/2] sin_i ;  o[i+rot/2] = x[i+rot/2] cos_i + x[i] sin_i
//   gptj : pairs (2i, 2i+1)    :  o[2i] = x[2i] cos_i - x[2i+1] sin_i ;   o[2i+1]   = x[2i+1] cos_i + x[2i] sin_i
// computed in fp32 from the fp32 cache, rounded once to the 16-bit dtype; columns >= rot_dim pass through.
// One wave per (token, head): q heads first, then the kv heads (which also carry the v row to the pool).
#include "rx_common.h"

namespace rx {

struct RopeArgs {
  uint16_t* q;
  uint16_t* k;
  const uint16_t* v;
  int64_t q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h;
  int64_t n;
  int32_t hq, hkv, d, dv, rot;
  const int64_t* positions;
  const float* cos_sin;
  int64_t cos_sin_stride;
  int32_t is_neox;
  // pool (optional)
  void* k_buf;
  void* v_buf;
  int32_t page_size, kv_fp8;
  int64_t kps, kts, khs, vps, vts, vhs;
  const void* loc;
  int32_t loc64;
  int64_t size_limit, skip_index;
  float k_scale, v_scale;
  int32_t* err_flag;
};

template <typename T>
__device__ __forceinline__ void put_elem(void* buf, bool fp8, int64_t off, float x, float scale) {
  if (fp8) {
    if (scale != 1.0f) x = T::to_f32(T::from_f32(x / scale));
    const uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0u, false);
    static_cast<uint8_t*>(buf)[off] = static_cast<uint8_t>(r & 0xffu);
  } else {
    static_cast<uint16_t*>(buf)[off] = T::from_f32(x);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rope_store_kernel(const RopeArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int heads = a.hq + a.hkv;
  if (unit >= a.n * heads) return;
  const int64_t t = unit / heads;
  const int hh = static_cast<int>(unit % heads);
  const bool is_k = hh >= a.hq;
  const int h = is_k ? hh - a.hq : hh;
  uint16_t* x = is_k ? a.k + t * a.k_stride_t + h * a.k_stride_h : a.q + t * a.q_stride_t + h * a.q_stride_h;
  const float* cs = a.cos_sin + a.positions[t] * a.cos_sin_stride;
  const int half = a.rot >> 1;

  // pool slot of this token (kv heads only)
  bool store = false;
  int64_t koff = 0, voff = 0;
  if (is_k && a.k_buf) {
    const int64_t idx = load_idx(a.loc, t, a.loc64);
    if (idx != a.skip_index) {
      if (idx < 0 || idx >= a.size_limit) {
        if (lane == 0 && a.err_flag) atomicOr(a.err_flag, RX_DEVERR_SLOT_OOB);
      } else {
        store = true;
        const int64_t pg = idx / a.page_size, off = idx % a.page_size;
        koff = pg * a.kps + off * a.kts + h * a.khs;
        voff = pg * a.vps + off * a.vts + h * a.vhs;
      }
    }
  }
  const bool fp8 = a.kv_fp8 != 0;
  // rotated part: pair index p -> elements (i0, i1)
  for (int p = lane; p < half; p += 64) {
    const int i0 = a.is_neox ? p : 2 * p, i1 = a.is_neox ? p + half : 2 * p + 1;
    const float c = cs[p], s = cs[half + p];
    const float x0 = T::to_f32(x[i0]), x1 = T::to_f32(x[i1]);
    const uint16_t o0 = T::from_f32(x0 * c - x1 * s), o1 = T::from_f32(x1 * c + x0 * s);
    x[i0] = o0;
    x[i1] = o1;
    if (store) {  // the pool holds exactly what attention would read back from k (rounded once)
      put_elem<T>(a.k_buf, fp8, koff + i0, T::to_f32(o0), a.k_scale);
      put_elem<T>(a.k_buf, fp8, koff + i1, T::to_f32(o1), a.k_scale);
    }
  }
  if (store) {
    for (int i = a.rot + lane; i < a.d; i += 64) put_elem<T>(a.k_buf, fp8, koff + i, T::to_f32(x[i]), a.k_scale);
    const uint16_t* vr = a.v + t * a.v_stride_t + h * a.v_stride_h;
    for (int i = lane; i < a.dv; i += 64) put_elem<T>(a.v_buf, fp8, voff + i, T::to_f32(vr[i]), a.v_scale);
  }
}

}  // namespace rx

using namespace rx;

extern "C" int rx_rope_store_kv(void* q, void* k, const void* v, int64_t q_stride_t, int64_t q_stride_h,
                                int64_t k_stride_t, int64_t k_stride_h, int64_t v_stride_t, int64_t v_stride_h,
                                int64_t n, int num_q_heads, int num_kv_heads, int head_dim, int v_head_dim,
                                int rotary_dim, const int64_t* positions, const float* cos_sin_cache,
                                int64_t cos_sin_stride, int is_neox, const rx_kv_layout* lay, const void* loc,
                                int loc_is_i64, int64_t size_limit, int64_t skip_index, float k_scale,
                                float v_scale, int dtype, int32_t* err_flag, void* stream) {
  RX_REQUIRE(n >= 0, "rx_rope_store_kv: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(q && k && positions && cos_sin_cache, "rx_rope_store_kv: null pointer");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_rope_store_kv: dtype %d", dtype);
  RX_REQUIRE(num_q_heads > 0 && num_kv_heads > 0 && head_dim > 0, "rx_rope_store_kv: bad head geometry");
  RX_REQUIRE(rotary_dim > 0 && rotary_dim % 2 == 0 && rotary_dim <= head_dim,
             "rx_rope_store_kv: rotary_dim %d must be even and <= head_dim %d", rotary_dim, head_dim);
  RopeArgs a{};
  a.q = static_cast<uint16_t*>(q);
  a.k = static_cast<uint16_t*>(k);
  a.v = static_cast<const uint16_t*>(v);
  a.q_stride_t = q_stride_t; a.q_stride_h = q_stride_h;
  a.k_stride_t = k_stride_t; a.k_stride_h = k_stride_h;
  a.v_stride_t = v_stride_t; a.v_stride_h = v_stride_h;
  a.n = n; a.hq = num_q_heads; a.hkv = num_kv_heads; a.d = head_dim; a.dv = v_head_dim; a.rot = rotary_dim;
  a.positions = positions; a.cos_sin = cos_sin_cache; a.cos_sin_stride = cos_sin_stride; a.is_neox = is_neox;
  if (lay) {
    RX_REQUIRE(v && loc && lay->k_buf && lay->v_buf && v_head_dim > 0, "rx_rope_store_kv: pool store needs v, loc and the layout's buffers");
    RX_REQUIRE(lay->page_size >= 1 && size_limit > 0, "rx_rope_store_kv: bad page_size / size_limit");
    RX_REQUIRE(k_scale > 0.f && v_scale > 0.f, "rx_rope_store_kv: scales must be > 0");
    a.k_buf = const_cast<void*>(lay->k_buf); a.v_buf = const_cast<void*>(lay->v_buf);
    a.page_size = lay->page_size; a.kv_fp8 = lay->kv_fp8;
    a.kps = lay->k_page_stride; a.kts = lay->k_tok_stride; a.khs = lay->k_head_stride;
    a.vps = lay->v_page_stride; a.vts = lay->v_tok_stride; a.vhs = lay->v_head_stride;
    a.loc = loc; a.loc64 = loc_is_i64; a.size_limit = size_limit; a.skip_index = skip_index;
    a.k_scale = k_scale; a.v_scale = v_scale; a.err_flag = err_flag;
  }
  const int64_t units = n * (num_q_heads + num_kv_heads);
  const dim3 grid(static_cast<unsigned>((units + 3) / 4)), block(256);
  auto s = static_cast<hipStream_t>(stream);
  if (dtype == RX_BF16) hipLaunchKernelGGL(rope_store_kernel<BF16>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(rope_store_kernel<F16>, grid, block, 0, s, a);
  return check_launch("rx_rope_store_kv");
}
