// Integer / byte kernels of the RadixAttention path: KV store (K1), kv-index build (K2),
// kv-split scheduler (K3), paged slot allocation (K9), KV move (K10), req_to_token write (K11).
// All HBM-bound byte movers: 16-byte per-lane accesses, one wave (or a fraction) per row.
#include <dlfcn.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "rx_common.h"

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace rx {

static thread_local char g_err[512];
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static thread_local char g_dispatch[256];
void note_dispatch(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_dispatch, sizeof(g_dispatch), fmt, ap);
  va_end(ap);
}

// ---- tooling hooks -------------------------------------------------------------------------------------------------
namespace {
using roctx_push_t = int (*)(const char*);
using roctx_pop_t = int (*)();
struct Roctx {
  roctx_push_t push = nullptr;
  roctx_pop_t pop = nullptr;
  Roctx() {
    for (const char* name : {"libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
      }
    }
  }
};
Roctx& roctx() {
  static Roctx r;
  return r;
}
const char* g_dump_dir = nullptr;  // RX_DUMP_DIR, read with the options (load_options below): once, off every launch path
const char* dump_dir() {
  (void)options();
  return g_dump_dir;
}
}  // namespace

int device_cu_count() {
  static std::atomic<int> cache[64];  // 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

RangeGuard::RangeGuard(const char* name) : on(false) {
  if (options().roctx && roctx().push) {
    roctx().push(name);
    on = true;
  }
}
RangeGuard::~RangeGuard() {
  if (on) roctx().pop();
}

int dump_on_error(const char* entry, int status, const void* params, size_t bytes) {
  const char* dir = dump_dir();
  if (status == RX_OK || !dir) return status;
  static std::atomic<int> seq{0};
  char path[768];
  const int n = seq.fetch_add(1);
  snprintf(path, sizeof(path), "%s/rx_%s_%d_%d.txt", dir, entry, static_cast<int>(getpid()), n);
  if (FILE* f = fopen(path, "w")) {
    fprintf(f, "entry: %s\nstatus: %d\nabi: %d\nerror: %s\nparams_bytes: %zu\nlast_dispatch: %s\n", entry, status, RX_ABI_VERSION,
            err_buf(), bytes, g_dispatch);
    fclose(f);
  }
  if (params && bytes) {
    snprintf(path, sizeof(path), "%s/rx_%s_%d_%d.bin", dir, entry, static_cast<int>(getpid()), n);
    if (FILE* f = fopen(path, "wb")) {
      fwrite(params, 1, bytes, f);
      fclose(f);
    }
  }
  return status;
}

struct OptionEntry {
  const char* name;
  int Options::*field;
};
static const OptionEntry kOptionTable[] = {
    {"ext32_autopack", &Options::ext32_autopack},         {"ext32_small_wg", &Options::ext32_small_wg},
    {"ext32_small_wg_tiles", &Options::ext32_small_wg_tiles}, {"ext32_pack_min_len", &Options::ext32_pack_min_len},
    {"ext32_pack_min_tiles", &Options::ext32_pack_min_tiles}, {"ext32_pack4_tiles", &Options::ext32_pack4_tiles}, {"ext32_pack_min_wgs", &Options::ext32_pack_min_wgs},
    {"roctx", &Options::roctx},                           {"qr_max_blocks", &Options::qr_max_blocks},
    {"qr_fenced", &Options::qr_fenced},                   {"ar_fenced", &Options::ar_fenced},
    {"ar_spin_log2", &Options::ar_spin_log2},             {"ar_blocks", &Options::ar_blocks},
    {"ext64", &Options::ext64},                           {"ext32_count_redo", &Options::ext32_count_redo},
    {"ext32_plain", &Options::ext32_plain},               {"ext32_uni", &Options::ext32_uni},               
                 {"extend_16x16_d128", &Options::extend_16x16_d128},
    {"extend_d256", &Options::extend_d256},               {"extend_d256_min_rows", &Options::extend_d256_min_rows},
                   {"extend_d256_at128", &Options::extend_d256_at128},
    {"extend_d256_at64", &Options::extend_d256_at64},     {"extend_d256_at96", &Options::extend_d256_at96},
    {"extend_nd", &Options::extend_nd},                   {"extend_nd_big", &Options::extend_nd_big},
    {"extend_mla", &Options::extend_mla},                 {"extend_mla_shared_v", &Options::extend_mla_shared_v},
    {"decode_mla8_dma", &Options::decode_mla8_dma},       {"decode_mla8_t64", &Options::decode_mla8_t64},
    {"merge_in_kernel_max_mb", &Options::merge_in_kernel_max_mb}, {"merge_in_kernel_max_mb_mla", &Options::merge_in_kernel_max_mb_mla},
};
static Options load_options() {  // once, at first use: RX_OPT_<NAME> (upper case) overrides a default
  Options o;
  for (const OptionEntry& e : kOptionTable) {
    char env[64] = "RX_OPT_";
    size_t n = strlen(env);
    for (const char* c = e.name; *c && n + 1 < sizeof(env); ++c) env[n++] = static_cast<char>(toupper(*c));
    env[n] = 0;
    if (const char* v = getenv(env)) o.*(e.field) = atoi(v);
  }
  if (const char* d = getenv("RX_DUMP_DIR"))  // (load_options: the argument-dump directory, see dump_on_error)
    if (*d) g_dump_dir = strdup(d);
  return o;
}
Options& options() {
  static Options o = load_options();
  return o;
}

// ---------------------------------------------------------------------------------------
// K1  store_kv.  Reference: store_kvcache, kernels/jit/csrc/elementwise/kvcache.cuh:189-219
// (one warp or 1/2/4 warps per row).  Here: 256-thread blocks, each 64-lane wave copies the
// K row and the V row of one token with 16-byte vectors (a 2 KiB Llama row = 2 sweeps of a
// wave); rows whose byte count is not a multiple of 16 fall back to 4-byte units.
// ---------------------------------------------------------------------------------------
template <int VEC>  // bytes per lane access: 16 or 4
__global__ __launch_bounds__(256) void store_kv_kernel(
    const char* __restrict__ k, const char* __restrict__ v, char* __restrict__ kc,
    char* __restrict__ vc, const void* __restrict__ loc, int64_t n, int64_t k_row, int64_t v_row,
    int64_t ks, int64_t vs, int64_t kcs, int64_t vcs, int loc64, int64_t size_limit,
    int64_t skip_index, int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t idx = load_idx(loc, row, loc64);
  if (idx == skip_index) return;
  if (idx < 0 || idx >= size_limit) {
    if (lane == 0 && err_flag) atomicOr(err_flag, RX_DEVERR_SLOT_OOB);
    return;
  }
  using V = typename std::conditional<VEC == 16, u32x4, uint32_t>::type;
  const char* ksrc = k + row * ks;
  const char* vsrc = v + row * vs;
  char* kdst = kc + idx * kcs;
  char* vdst = vc + idx * vcs;
  for (int64_t off = lane * VEC; off < k_row; off += 64 * VEC)
    *reinterpret_cast<V*>(kdst + off) = *reinterpret_cast<const V*>(ksrc + off);
  for (int64_t off = lane * VEC; off < v_row; off += 64 * VEC)
    *reinterpret_cast<V*>(vdst + off) = *reinterpret_cast<const V*>(vsrc + off);
}


// K1 for paged (HND) pools: dst(slot, head) = (slot/page)*page_stride + (slot%page)*tok_stride +
// head*head_stride.  The reference scatters with torch index_put for this layout
// (memory_pool.py:2372-2379); here one wave per token walks the heads with 16-byte vectors.
__global__ __launch_bounds__(256) void store_kv_layout_kernel(
    const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, uint16_t* __restrict__ kc,
    uint16_t* __restrict__ vc, const void* __restrict__ loc, int64_t n, int hkv, int dk, int dv,
    int64_t k_stride_t, int64_t v_stride_t, int page_size, int64_t kps, int64_t kts, int64_t khs,
    int64_t vps, int64_t vts, int64_t vhs, int loc64, int64_t size_limit, int64_t skip_index,
    int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t idx = load_idx(loc, row, loc64);
  if (idx == skip_index) return;
  if (idx < 0 || idx >= size_limit) {
    if (lane == 0 && err_flag) atomicOr(err_flag, RX_DEVERR_SLOT_OOB);
    return;
  }
  const int64_t pg = idx / page_size, off = idx % page_size;
  for (int e = lane * 8; e < hkv * dk; e += 64 * 8) {  // 8 elements = 16 B per lane
    const int h = e / dk, d = e % dk;
    *reinterpret_cast<u32x4*>(kc + pg * kps + off * kts + h * khs + d) =
        *reinterpret_cast<const u32x4*>(k + row * k_stride_t + e);
  }
  for (int e = lane * 8; e < hkv * dv; e += 64 * 8) {
    const int h = e / dv, d = e % dv;
    *reinterpret_cast<u32x4*>(vc + pg * vps + off * vts + h * vhs + d) =
        *reinterpret_cast<const u32x4*>(v + row * v_stride_t + e);
  }
}

// Quantising store: one wave per token, 8 source elements (16 B) -> 8 fp8 bytes per lane.
template <typename T>
__global__ __launch_bounds__(256) void store_kv_fp8_kernel(
    const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, uint8_t* __restrict__ kc,
    uint8_t* __restrict__ vc, const void* __restrict__ loc, int64_t n, int hkv, int dk, int dv,
    int64_t k_stride_t, int64_t v_stride_t, int page_size, int64_t kps, int64_t kts, int64_t khs,
    int64_t vps, int64_t vts, int64_t vhs, float k_scale, float v_scale, int loc64,
    int64_t size_limit, int64_t skip_index, int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t idx = load_idx(loc, row, loc64);
  if (idx == skip_index) return;
  if (idx < 0 || idx >= size_limit) {
    if (lane == 0 && err_flag) atomicOr(err_flag, RX_DEVERR_SLOT_OOB);
    return;
  }
  const int64_t pg = idx / page_size, off = idx % page_size;
  auto quant8 = [](u32x4 raw, float scale) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = T::to_f32(static_cast<uint16_t>(raw[i] & 0xffffu));
      f[2 * i + 1] = T::to_f32(static_cast<uint16_t>(raw[i] >> 16));
    }
    if (scale != 1.0f) {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = T::to_f32(T::from_f32(f[i] / scale));  // div_ rounds to the source dtype
    }
    u32x2 out;
    out[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0u, false);
    out[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], out[0], true);
    out[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0u, false);
    out[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], out[1], true);
    return out;
  };
  for (int e = lane * 8; e < hkv * dk; e += 64 * 8) {
    const int h = e / dk, d = e % dk;
    *reinterpret_cast<u32x2*>(kc + pg * kps + off * kts + h * khs + d) =
        quant8(*reinterpret_cast<const u32x4*>(k + row * k_stride_t + e), k_scale);
  }
  for (int e = lane * 8; e < hkv * dv; e += 64 * 8) {
    const int h = e / dv, d = e % dv;
    *reinterpret_cast<u32x2*>(vc + pg * vps + off * vts + h * vhs + d) =
        quant8(*reinterpret_cast<const u32x4*>(v + row * v_stride_t + e), v_scale);
  }
}

// fused_fp8_qkv_kv_cache (kernels/jit/csrc/attention/fused_fp8_qkv_kv_cache.cuh:56-91): y = float(x) * (1.0f / *scale)
// saturated to +-448 and rounded to e4m3fn (CUDA's static_cast<fp8_e4m3>: satfinite, RNE); q with scale 1.  One wave per
// token, 8 source elements (16 B) -> 8 fp8 bytes per lane.  The scales are DEVICE scalars, as the reference's.
template <typename T, bool QUANT_Q>
__global__ __launch_bounds__(256) void fused_fp8_qkv_store_kernel(
    const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
    uint8_t* __restrict__ q_out, uint8_t* __restrict__ kc, uint8_t* __restrict__ vc, const void* __restrict__ loc,
    const float* __restrict__ k_scale, const float* __restrict__ v_scale, int64_t n, int q_dim, int hkv, int dk, int dv,
    int64_t q_stride_t, int64_t k_stride_t, int64_t v_stride_t, int page_size, int64_t kps, int64_t kts, int64_t khs,
    int64_t vps, int64_t vts, int64_t vhs, int loc64, int64_t size_limit, int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  auto quant8 = [](u32x4 raw, float inv) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = T::to_f32(static_cast<uint16_t>(raw[i] & 0xffffu)) * inv;
      f[2 * i + 1] = T::to_f32(static_cast<uint16_t>(raw[i] >> 16)) * inv;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (f[i] != f[i]) ? f[i] : fminf(fmaxf(f[i], -448.f), 448.f);  // satfinite; NaN stays NaN
    u32x2 out;
    out[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0u, false);
    out[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], out[0], true);
    out[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0u, false);
    out[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], out[1], true);
    return out;
  };
  if constexpr (QUANT_Q) {
    for (int e = lane * 8; e < q_dim; e += 64 * 8)
      *reinterpret_cast<u32x2*>(q_out + row * q_dim + e) = quant8(*reinterpret_cast<const u32x4*>(q + row * q_stride_t + e), 1.0f);
  }
  const int64_t idx = load_idx(loc, row, loc64);
  if (idx < 0 || idx >= size_limit) {
    if (lane == 0 && err_flag) atomicOr(err_flag, RX_DEVERR_SLOT_OOB);
    return;
  }
  const float inv_k = 1.0f / (k_scale ? *k_scale : 1.0f);
  const float inv_v = 1.0f / (v_scale ? *v_scale : 1.0f);
  const int64_t pg = idx / page_size, off = idx % page_size;
  for (int e = lane * 8; e < hkv * dk; e += 64 * 8) {
    const int h = e / dk, d = e % dk;
    *reinterpret_cast<u32x2*>(kc + pg * kps + off * kts + h * khs + d) =
        quant8(*reinterpret_cast<const u32x4*>(k + row * k_stride_t + e), inv_k);
  }
  for (int e = lane * 8; e < hkv * dv; e += 64 * 8) {
    const int h = e / dv, d = e % dv;
    *reinterpret_cast<u32x2*>(vc + pg * vps + off * vts + h * vhs + d) =
        quant8(*reinterpret_cast<const u32x4*>(v + row * v_stride_t + e), inv_v);
  }
}

// K12 read side: one wave per gathered row, 8 elements per lane per step.
template <typename T, bool KV8>
__global__ __launch_bounds__(256) void get_mla_kv_kernel(const void* __restrict__ buf, int64_t row_stride,
                                                         const void* __restrict__ loc, int loc64, int64_t n,
                                                         int nope_cols, int rope_cols,
                                                         uint16_t* __restrict__ nope_out,
                                                         uint16_t* __restrict__ rope_out,
                                                         int64_t size_limit, int32_t* err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t idx = load_idx(loc, row, loc64);
  if (idx < 0 || idx >= size_limit) {
    if (lane == 0 && err_flag) atomicOr(err_flag, RX_DEVERR_SLOT_OOB);
    return;
  }
  for (int e = lane * 8; e < nope_cols + rope_cols; e += 64 * 8) {
    u32x4 v16;
    if constexpr (KV8) {
      v16 = fp8x8_to_16<T>(*reinterpret_cast<const u32x2*>(static_cast<const uint8_t*>(buf) + idx * row_stride + e));
    } else {
      v16 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(buf) + idx * row_stride + e);
    }
    if (e < nope_cols) *reinterpret_cast<u32x4*>(nope_out + row * nope_cols + e) = v16;
    else *reinterpret_cast<u32x4*>(rope_out + row * rope_cols + (e - nope_cols)) = v16;
  }
}

// merge_state: one thread = 8 consecutive columns of one (token, head)
template <typename T>
__global__ __launch_bounds__(256) void merge_state_kernel(const uint16_t* __restrict__ a,
                                                          const float* __restrict__ lse_a,
                                                          const uint16_t* __restrict__ b,
                                                          const float* __restrict__ lse_b,
                                                          uint16_t* __restrict__ out, float* __restrict__ out_lse,
                                                          int64_t rows, int head_size) {
  const int per_row = head_size >> 3;
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t row = gid / per_row;
  if (row >= rows) return;
  const int c = static_cast<int>(gid % per_row) * 8;
  float la = lse_a[row], lb = lse_b[row];
  la = (la == INFINITY) ? -INFINITY : la;  // merge_state.py:28-29
  lb = (lb == INFINITY) ? -INFINITY : lb;
  const float m = fmaxf(la, lb);
  const float wa = __expf(la - m), wb = __expf(lb - m);
  const float se = wa + wb;
  const float sa = wa / se, sb = wb / se;
  const u32x4 va = *reinterpret_cast<const u32x4*>(a + row * head_size + c);
  const u32x4 vb = *reinterpret_cast<const u32x4*>(b + row * head_size + c);
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a0 = T::to_f32(static_cast<uint16_t>(va[j] & 0xffffu)), a1 = T::to_f32(static_cast<uint16_t>(va[j] >> 16));
    const float b0 = T::to_f32(static_cast<uint16_t>(vb[j] & 0xffffu)), b1 = T::to_f32(static_cast<uint16_t>(vb[j] >> 16));
    o[j] = pack2<T>(a0 * sa + b0 * sb, a1 * sa + b1 * sb);
  }
  *reinterpret_cast<u32x4*>(out + row * head_size + c) = o;
  if (out_lse && c == 0) out_lse[row] = __logf(se) + m;
}

// merge_chunks: n-way form of merge_state for a split pass.  Group g (= one request) has S chunk partials
// [S][R rows] (chunk-major inside the group) and optionally one more partial [R rows] from another tensor;
// one thread = 8 consecutive columns of one (row, head).  Empty partials (lse -inf / +inf) are skipped.
template <typename T>
__global__ __launch_bounds__(256) void merge_chunks_kernel(const uint16_t* __restrict__ oc, const float* __restrict__ lc,
                                                           int S, const uint16_t* __restrict__ ol,
                                                           const float* __restrict__ ll, uint16_t* __restrict__ out,
                                                           float* __restrict__ out_lse, int64_t groups, int R, int heads,
                                                           int head_size) {
  const int per_row = head_size >> 3;
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t rh = gid / per_row;  // (group, row, head) flattened
  if (rh >= groups * R * heads) return;
  const int c = static_cast<int>(gid % per_row) * 8;
  const int64_t g = rh / (static_cast<int64_t>(R) * heads);
  const int64_t in_g = rh % (static_cast<int64_t>(R) * heads);  // row * heads + head
  const int64_t base = g * S * R * heads + in_g;               // chunk 0 of this (row, head); + x * R * heads
  // loads in independent batches of 8 (one memory latency per batch, not per chunk); an empty partial gets weight 0
  const int64_t cstride = static_cast<int64_t>(R) * heads;
  auto clean = [](float l) { return (l < INFINITY) ? l : -INFINITY; };  // +inf / NaN -> empty
  float m = -INFINITY;
  for (int x0 = 0; x0 < S; x0 += 8) {
    float l8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l8[j] = lc[base + static_cast<int64_t>(min(x0 + j, S - 1)) * cstride];
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, clean(l8[j]));
  }
  float l_last = -INFINITY;
  if (ol) {
    l_last = clean(ll[rh]);
    m = fmaxf(m, l_last);
  }
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float se = 0.f;
  auto add = [&](u32x4 v, float w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += w * T::to_f32(static_cast<uint16_t>(v[j] & 0xffffu));
      acc[2 * j + 1] += w * T::to_f32(static_cast<uint16_t>(v[j] >> 16));
    }
    se += w;
  };
  for (int x0 = 0; x0 < S; x0 += 8) {
    float l8[8];
    u32x4 v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = base + static_cast<int64_t>(min(x0 + j, S - 1)) * cstride;
      l8[j] = lc[r];
      v8[j] = *reinterpret_cast<const u32x4*>(oc + r * head_size + c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float l = clean(l8[j]);
      if (x0 + j < S && l > -INFINITY) add(v8[j], __expf(l - m));  // (an empty partial's row may hold NaN)
    }
  }
  if (ol && l_last > -INFINITY) add(*reinterpret_cast<const u32x4*>(ol + rh * head_size + c), __expf(l_last - m));
  const float inv = 1.0f / se;
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack2<T>(acc[2 * j] * inv, acc[2 * j + 1] * inv);
  *reinterpret_cast<u32x4*>(out + rh * head_size + c) = o;
  if (out_lse && c == 0) out_lse[rh] = __logf(se) + m;
}

// ---------------------------------------------------------------------------------------
// K2  kv_indptr scan + ragged gather of req_to_token rows.
// Reference: create_flashinfer_kv_indices_triton (kv_indices.py:8-46), grid (bs,), 512-wide.
// Here: one 1024-thread block scans lens -> kv_indptr (bs is at most a few thousand), then a
// (chunk, bs) grid copies 1024-token chunks, 4 B per lane coalesced.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void indptr_scan_kernel(const void* __restrict__ lens,
                                                           int lens64, int bs,
                                                           int32_t* __restrict__ out) {
  __shared__ int32_t wave_sums[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
    carry_s = 0;
    out[0] = 0;
  }
  __syncthreads();
  for (int base = 0; base < bs; base += 1024) {
    const int i = base + tid;
    int32_t x = (i < bs) ? static_cast<int32_t>(load_idx(lens, i, lens64)) : 0;
    // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wave_sums[wid] = x;
    __syncthreads();
    int32_t prefix = carry_s;
    for (int w = 0; w < wid; ++w) prefix += wave_sums[w];
    if (i < bs) out[i + 1] = prefix + x;
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// K8 metadata: build_unified_kv_indices (kernels/ops/attention/extend_attention.py:193-238; copy kernel :135-190)
// One scan block (unified lens = prefix len + extend len -> indptr, prefix_lens), then a (chunk, request) grid that
// copies the request's prefix slots and then its new tokens' slots, coalesced.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void unified_indptr_kernel(const int32_t* __restrict__ prefix_indptr, const void* __restrict__ ext_lens,
                                                              int lens64, int bs, int32_t* __restrict__ out,
                                                              int32_t* __restrict__ prefix_lens) {
  __shared__ int32_t wave_sums[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) {
    carry_s = 0;
    out[0] = 0;
  }
  __syncthreads();
  for (int base = 0; base < bs; base += 1024) {
    const int i = base + tid;
    int32_t x = 0;
    if (i < bs) {
      const int32_t pl = prefix_indptr[i + 1] - prefix_indptr[i];
      if (prefix_lens) prefix_lens[i] = pl;
      x = pl + static_cast<int32_t>(load_idx(ext_lens, i, lens64));
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
    if (i < bs) out[i + 1] = prefix + x;
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
  }
}

template <typename PreT, typename ExtT>
__global__ __launch_bounds__(256) void unified_indices_copy_kernel(const int32_t* __restrict__ prefix_indptr, const PreT* __restrict__ prefix_idx,
                                                                   const void* __restrict__ ext_start, int start64,
                                                                   const ExtT* __restrict__ ext_idx,
                                                                   const int32_t* __restrict__ unified_indptr,
                                                                   int64_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int32_t p0 = prefix_indptr[b], pl = prefix_indptr[b + 1] - p0;
  const int32_t u0 = unified_indptr[b], total = unified_indptr[b + 1] - u0;
  const int64_t e0 = load_idx(ext_start, b, start64);
  for (int32_t j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256)
    out[u0 + j] = j < pl ? static_cast<int64_t>(prefix_idx[p0 + j]) : static_cast<int64_t>(ext_idx[e0 + (j - pl)]);
}

template <typename OutT>
__global__ __launch_bounds__(256) void kv_indices_gather_kernel(
    const int32_t* __restrict__ req_to_token, int64_t row_stride,
    const void* __restrict__ req_pool_indices, int pool64, const int32_t* __restrict__ kv_start,
    const int32_t* __restrict__ kv_indptr, OutT* __restrict__ out) {
  const int b = blockIdx.y;
  const int32_t beg = kv_indptr[b];
  const int32_t len = kv_indptr[b + 1] - beg;
  const int32_t chunk0 = blockIdx.x * 1024;
  if (chunk0 >= len) return;
  const int64_t req = load_idx(req_pool_indices, b, pool64);
  const int32_t* src = req_to_token + req * row_stride + (kv_start ? kv_start[b] : 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int32_t t = chunk0 + j * 256 + threadIdx.x;
    if (t < len) out[static_cast<int64_t>(beg) + t] = static_cast<OutT>(src[t]);
  }
}

// ---------------------------------------------------------------------------------------
// K3  num_kv_splits.  Reference: get_num_kv_splits_triton (metadata.py:11-60), grid (1,).
// Integer-exact restatement; the only float op is log2(max_seq/64) in fp32 as in Triton.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t cdiv32(int32_t a, int32_t b) { return (a + b - 1) / b; }

__global__ __launch_bounds__(1024) void num_kv_splits_kernel(
    const void* __restrict__ seq_lens, int is64, int num_seq, int num_group, int num_head,
    int num_kv_head, int max_kv_splits, int device_core_count, int32_t* __restrict__ out) {
  __shared__ int32_t smax[16], smin[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int32_t mx = 0, mn = INT32_MAX;
  for (int i = tid; i < num_seq; i += 1024) {
    const int32_t s = static_cast<int32_t>(load_idx(seq_lens, i, is64));
    mx = max(mx, s);
    mn = min(mn, s);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mx = max(mx, __shfl_xor(mx, d));
    mn = min(mn, __shfl_xor(mn, d));
  }
  if (lane == 0) {
    smax[wid] = mx;
    smin[wid] = mn;
  }
  __syncthreads();
  mx = smax[0];
  mn = smin[0];
  for (int w = 1; w < 16; ++w) {
    mx = max(mx, smax[w]);
    mn = min(mn, smin[w]);
  }
  const int32_t max_seq_len = mx;
  int32_t min_seq_len = mn;
  if (max_seq_len * 8 < min_seq_len * 10) min_seq_len = max_seq_len;
  if (min_seq_len < 1) min_seq_len = 1;  // empty request: the reference divides by zero here
  const int32_t max_kv_splits_1 = min(cdiv32(max_seq_len, min_seq_len), max_kv_splits);
  const int32_t kv_chunk_size_1 = cdiv32(max_seq_len, max_kv_splits_1);
  const float ext_seq_len = static_cast<float>(max_seq_len) / 64.0f;
  const int32_t ext_cores = static_cast<int32_t>(
      static_cast<float>(device_core_count) * fmaxf(log2f(ext_seq_len), 1.0f));
  int32_t block_h = 16;
  const int32_t num_kv_group = num_head / num_kv_head;
  int32_t token_grid;
  if (num_kv_group == 1) {
    token_grid = num_seq * num_group * num_head;
  } else {
    block_h = min(block_h, num_kv_group);
    token_grid = num_seq * num_group * cdiv32(num_head, block_h);
  }
  const int32_t max_kv_splits_2 = min(cdiv32(ext_cores, token_grid), max_kv_splits);
  const int32_t kv_chunk_size_2 = cdiv32(max_seq_len, max_kv_splits_2);
  for (int i = tid; i < num_seq; i += 1024) {
    const int32_t s = static_cast<int32_t>(load_idx(seq_lens, i, is64));
    const int32_t n = max(cdiv32(s, kv_chunk_size_1), cdiv32(s, kv_chunk_size_2));
    for (int g = 0; g < num_group; ++g) out[i * num_group + g] = n;
  }
}

// ---------------------------------------------------------------------------------------
// K9  paged allocation.  Reference: alloc_extend_kernel / alloc_decode_kernel
// (kernels/ops/memory/allocator.py:16-135), grid (bs,), each program re-sums the lens of the
// requests before it.  Same decomposition here: one 256-thread block per request.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t block_sum_256(int64_t x, int64_t* sh) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  const int64_t r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void alloc_extend_kernel(
    const int64_t* __restrict__ pre_lens, const int64_t* __restrict__ seq_lens,
    const int64_t* __restrict__ last_loc, const int64_t* __restrict__ free_pages,
    int64_t* __restrict__ out, int page_size) {
  __shared__ int64_t sh[4];
  const int pid = blockIdx.x, tid = threadIdx.x;
  const int64_t ps = page_size;
  int64_t ext_before = 0, pages_before = 0;
  for (int i = tid; i < pid; i += 256) {
    const int64_t s = seq_lens[i], p = pre_lens[i];
    ext_before += s - p;
    pages_before += (s + ps - 1) / ps - (p + ps - 1) / ps;
  }
  const int64_t out_start = block_sum_256(ext_before, sh);
  const int64_t page_start = block_sum_256(pages_before, sh);
  const int64_t seq = seq_lens[pid], pre = pre_lens[pid];
  const int64_t new_pages = (seq + ps - 1) / ps - (pre + ps - 1) / ps;
  // part 1: finish the old partial page
  const int64_t pre_up = (pre + ps - 1) / ps * ps;
  const int64_t n1 = min(seq, pre_up) - pre;
  const int64_t ll = last_loc[pid];
  for (int64_t j = tid; j < n1; j += 256) out[out_start + j] = ll + 1 + j;
  if (pre + n1 == seq) return;
  // part 2: whole new pages
  const int64_t n2 = seq / ps * ps - pre_up;
  for (int64_t j = tid; j < n2; j += 256)
    out[out_start + n1 + j] = free_pages[page_start + j / ps] * ps + j % ps;
  if (pre + n1 + n2 == seq) return;
  // part 3: the new partial page
  const int64_t n3 = seq - seq / ps * ps;
  const int64_t start = free_pages[page_start + new_pages - 1];
  for (int64_t j = tid; j < n3; j += 256) out[out_start + n1 + n2 + j] = start * ps + j;
}

__global__ __launch_bounds__(256) void alloc_decode_kernel(
    const int64_t* __restrict__ seq_lens, const int64_t* __restrict__ last_loc,
    const int64_t* __restrict__ free_pages, int64_t* __restrict__ out, int page_size) {
  __shared__ int64_t sh[4];
  const int pid = blockIdx.x, tid = threadIdx.x;
  const int64_t ps = page_size;
  int64_t pages_before = 0;
  for (int i = tid; i < pid; i += 256) {
    const int64_t s = seq_lens[i];
    pages_before += (s + ps - 1) / ps - (s - 1 + ps - 1) / ps;
  }
  const int64_t page_start = block_sum_256(pages_before, sh);
  if (tid != 0) return;
  const int64_t s = seq_lens[pid];
  const int64_t mine = (s + ps - 1) / ps - (s - 1 + ps - 1) / ps;
  out[pid] = (mine == 0) ? last_loc[pid] + 1 : free_pages[page_start] * ps;
}

// ---------------------------------------------------------------------------------------
// K11  write_req_to_token.  Reference: write_req_to_token_pool_triton
// (called srt/mem_cache/allocation.py:75-84), grid (bs,).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void write_req_to_token_kernel(
    int32_t* __restrict__ req_to_token, int64_t row_stride,
    const int64_t* __restrict__ req_pool_indices, const int64_t* const* __restrict__ prefix_ptrs,
    const int64_t* __restrict__ pre_lens, const int64_t* __restrict__ seq_lens,
    const int64_t* __restrict__ extend_lens, const int64_t* __restrict__ out_cache_loc) {
  __shared__ int64_t sh[4];
  const int pid = blockIdx.x, tid = threadIdx.x;
  int64_t before = 0;
  for (int i = tid; i < pid; i += 256) before += extend_lens[i];
  const int64_t ext_off = block_sum_256(before, sh);
  int32_t* row = req_to_token + req_pool_indices[pid] * row_stride;
  const int64_t pre = pre_lens[pid], seq = seq_lens[pid];
  const int64_t* pfx = prefix_ptrs ? prefix_ptrs[pid] : nullptr;
  if (pfx)
    for (int64_t j = tid; j < pre; j += 256) row[j] = static_cast<int32_t>(pfx[j]);
  for (int64_t j = tid; j < seq - pre; j += 256)
    row[pre + j] = static_cast<int32_t>(out_cache_loc[ext_off + j]);
}

// ---------------------------------------------------------------------------------------
// K10  move_kv over every layer buffer.  Reference: copy_all_layer_kv_cache_tiled
// (kernels/ops/kvcache/cache_move.py:60-133): grid (buffer, token-tile); one wave per row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void move_kv_kernel(const uint64_t* __restrict__ data_ptrs,
                                                      const int64_t* __restrict__ row_bytes,
                                                      const int64_t* __restrict__ tgt,
                                                      const int64_t* __restrict__ src,
                                                      int64_t n) {
  const int buf = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  char* base = reinterpret_cast<char*>(data_ptrs[buf]);
  const int64_t rb = row_bytes[buf];
  const char* s = base + src[i] * rb;
  char* d = base + tgt[i] * rb;
  if ((rb & 15) == 0) {
    for (int64_t off = lane * 16; off < rb; off += 64 * 16)
      *reinterpret_cast<u32x4*>(d + off) = *reinterpret_cast<const u32x4*>(s + off);
  } else {
    for (int64_t off = lane * 4; off < rb; off += 64 * 4)
      *reinterpret_cast<uint32_t*>(d + off) = *reinterpret_cast<const uint32_t*>(s + off);
  }
}

// K10 on a paged layout (HND pools: a slot's row is Hkv pieces, one per head, page_stride / head_stride /
// tok_stride apart): same contract, the row address comes from rx_kv_layout-style strides.  geom[b] =
// {page_stride, head_stride, tok_stride, piece_bytes} of buffer b, all in BYTES.  One wave per (slot, buffer).
__global__ __launch_bounds__(256) void move_kv_layout_kernel(const uint64_t* __restrict__ data_ptrs,
                                                             const int64_t* __restrict__ geom, int32_t page_size,
                                                             int32_t num_heads, const int64_t* __restrict__ tgt,
                                                             const int64_t* __restrict__ src, int64_t n) {
  const int buf = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  char* base = reinterpret_cast<char*>(data_ptrs[buf]);
  const int64_t ps = geom[4 * buf], hs = geom[4 * buf + 1], ts = geom[4 * buf + 2], piece = geom[4 * buf + 3];
  const int64_t s_slot = src[i], t_slot = tgt[i];
  const char* s = base + (s_slot / page_size) * ps + (s_slot % page_size) * ts;
  char* d = base + (t_slot / page_size) * ps + (t_slot % page_size) * ts;
  const int64_t total = piece * num_heads;
  if ((piece & 15) == 0) {
    for (int64_t off = lane * 16; off < total; off += 64 * 16) {
      const int64_t h = off / piece, w = off - h * piece;
      *reinterpret_cast<u32x4*>(d + h * hs + w) = *reinterpret_cast<const u32x4*>(s + h * hs + w);
    }
  } else {
    for (int64_t off = lane * 4; off < total; off += 64 * 4) {
      const int64_t h = off / piece, w = off - h * piece;
      *reinterpret_cast<uint32_t*>(d + h * hs + w) = *reinterpret_cast<const uint32_t*>(s + h * hs + w);
    }
  }
}

}  // namespace rx

using namespace rx;

// ---- per-step page tables of EAGLE's multi-step draft decode (generate_draft_decode_kv_indices, cache_locs.py:56-141) ----
// One launch for every (step, request, branch): block (z = b * topk + k, chunk, step).  A branch's keys at step i are the
// request's cached tokens plus the i + 1 draft tokens the branch has written.  The prefix sums the reference recomputes per
// program (tl.sum over the lengths / positions before this one) are block reductions here; rows are copied 1024 tokens per
// block, int32 page-table words read coalesced, index words written coalesced.
namespace rx {
__device__ __forceinline__ int64_t block_sum_before(const void* v, int is64, int n, int64_t* sh) {
  int64_t acc = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += load_idx(v, i, is64);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  int64_t tot = 0;
  for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) tot += sh[w];
  __syncthreads();
  return tot;
}

template <typename OutT>
__global__ __launch_bounds__(256) void draft_decode_kv_indices_kernel(
    const int32_t* __restrict__ req_to_token, int64_t row_stride, const void* __restrict__ req_pool_indices, int rpi64,
    const void* __restrict__ seq_lens, int sl64, const void* __restrict__ positions, int pos64, int num_seqs, int topk,
    int num_steps, int page_size, OutT* __restrict__ kv_indices, int64_t kv_indices_stride, int32_t* __restrict__ kv_indptr,
    int64_t kv_indptr_stride) {
  __shared__ int64_t sh[4];
  const int z = blockIdx.x, b = z / topk, k = z - b * topk;
  const int chunk = blockIdx.y, iters = blockIdx.z + 1;
  const int64_t n = load_idx(seq_lens, b, sl64);
  const int64_t lo = static_cast<int64_t>(chunk) * 1024;
  if (chunk > 0 && lo >= n) return;                     // (block-uniform)
  const int64_t cum = block_sum_before(seq_lens, sl64, b, sh);
  const int32_t* row = req_to_token + load_idx(req_pool_indices, b, rpi64) * row_stride;
  OutT* dst = kv_indices + kv_indices_stride * blockIdx.z + cum * topk + static_cast<int64_t>(b) * iters * topk +
              static_cast<int64_t>(k) * (n + iters);
  for (int64_t j = lo + threadIdx.x; j < min(lo + 1024, n); j += 256) dst[j] = static_cast<OutT>(row[j]);
  if (chunk != 0) return;
  int64_t start;
  if (page_size == 1 || topk == 1) {
    start = n + static_cast<int64_t>(k) * num_steps;
  } else {  // every branch on pages of its own behind the request's last (partial) page
    const int64_t last = n % page_size, new_pages = (last + num_steps + page_size - 1) / page_size;
    start = n / page_size * page_size + static_cast<int64_t>(k) * new_pages * page_size + last;
  }
  if (static_cast<int>(threadIdx.x) < iters) dst[n + threadIdx.x] = static_cast<OutT>(row[start + threadIdx.x]);
  const int zz = z == 0 ? num_seqs * topk : z;          // (entry 0 stays as the caller left it: 0)
  const int64_t base = block_sum_before(positions, pos64, zz, sh);
  if (threadIdx.x == 0) kv_indptr[kv_indptr_stride * blockIdx.z + zz] = static_cast<int32_t>(base + static_cast<int64_t>(zz) * iters);
}
}  // namespace rx


namespace rx {
// one wave: both clocks before and after spin_ticks of the constant-rate one, asleep in between (no issue slots, no power)
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long spin_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64(), c0 = clock64();
  unsigned long long t1 = t0;
  while (t1 - t0 < spin_ticks) {
    __builtin_amdgcn_s_sleep(64);
    t1 = wall_clock64();
  }
  const unsigned long long c1 = clock64();
  out[0] = c1 - c0;
  out[1] = t1 - t0;
}
}  // namespace rx

extern "C" {

int rx_clock_probe(uint64_t* out2_dev, int32_t spin_us, void* stream) {
  RX_REQUIRE(out2_dev, "rx_clock_probe: null output");
  RX_REQUIRE(spin_us > 0 && spin_us <= 1000000, "rx_clock_probe: spin_us must be in (0, 1e6]");
  hipLaunchKernelGGL(rx::clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<unsigned long long*>(out2_dev), static_cast<unsigned long long>(spin_us) * 100ull);
  if (hipGetLastError() != hipSuccess) return rx::fail(RX_ERR_LAUNCH, "rx_clock_probe: launch failed");
  return RX_OK;
}

int rx_version(void) { return RX_ABI_VERSION; }
int64_t rx_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(rx_kv_layout);
    case 1: return sizeof(rx_decode_params);
    case 2: return sizeof(rx_extend_params);
    default: return -1;
  }
}
const char* rx_last_error(void) { return rx::err_buf(); }
const char* rx_last_dispatch(void) { return rx::g_dispatch; }
int rx_set_option(const char* name, int value) {
  RX_REQUIRE(name, "rx_set_option: null name");
  for (const rx::OptionEntry& e : rx::kOptionTable)
    if (!strcmp(e.name, name)) {
      rx::options().*(e.field) = value;
      return RX_OK;
    }
  return rx::fail(RX_ERR_INVALID_ARG, "rx_set_option: unknown option '%s'", name);
}
int rx_get_option(const char* name, int* value) {
  RX_REQUIRE(name && value, "rx_get_option: null argument");
  for (const rx::OptionEntry& e : rx::kOptionTable)
    if (!strcmp(e.name, name)) {
      *value = rx::options().*(e.field);
      return RX_OK;
    }
  return rx::fail(RX_ERR_INVALID_ARG, "rx_get_option: unknown option '%s'", name);
}

int rx_store_kv(const void* k, const void* v, void* k_cache, void* v_cache, const void* loc,
                int64_t n, int64_t k_row_bytes, int64_t v_row_bytes, int64_t k_stride_bytes,
                int64_t v_stride_bytes, int64_t kc_stride_bytes, int64_t vc_stride_bytes,
                int loc_is_i64, int64_t size_limit, int64_t skip_index, int32_t* err_flag,
                void* stream) {
  RX_RANGE("rx_store_kv");
  RX_REQUIRE(n >= 0, "rx_store_kv: n=%lld < 0", (long long)n);
  if (n == 0) return RX_OK;
  RX_REQUIRE(k && v && k_cache && v_cache && loc, "rx_store_kv: null pointer");
  RX_REQUIRE(k_row_bytes > 0 && v_row_bytes > 0 && k_row_bytes % 4 == 0 && v_row_bytes % 4 == 0,
             "rx_store_kv: row bytes (%lld,%lld) must be positive multiples of 4",
             (long long)k_row_bytes, (long long)v_row_bytes);
  RX_REQUIRE(size_limit > 0, "rx_store_kv: size_limit must be > 0");
  const bool vec16 = ((k_row_bytes | v_row_bytes | k_stride_bytes | v_stride_bytes |
                       kc_stride_bytes | vc_stride_bytes) % 16 == 0) &&
                     (((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0);
  if (!vec16)
    RX_REQUIRE(((k_stride_bytes | v_stride_bytes | kc_stride_bytes | vc_stride_bytes) % 4 == 0) &&
                   (((uintptr_t)k | (uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) % 4 == 0),
               "rx_store_kv: strides / base pointers must be 4-byte aligned");
  const dim3 grid(static_cast<unsigned>((n + 3) / 4)), block(256);
  auto s = static_cast<hipStream_t>(stream);
  if (vec16)
    hipLaunchKernelGGL(store_kv_kernel<16>, grid, block, 0, s, (const char*)k, (const char*)v,
                       (char*)k_cache, (char*)v_cache, loc, n, k_row_bytes, v_row_bytes,
                       k_stride_bytes, v_stride_bytes, kc_stride_bytes, vc_stride_bytes,
                       loc_is_i64, size_limit, skip_index, err_flag);
  else
    hipLaunchKernelGGL(store_kv_kernel<4>, grid, block, 0, s, (const char*)k, (const char*)v,
                       (char*)k_cache, (char*)v_cache, loc, n, k_row_bytes, v_row_bytes,
                       k_stride_bytes, v_stride_bytes, kc_stride_bytes, vc_stride_bytes,
                       loc_is_i64, size_limit, skip_index, err_flag);
  return check_launch("rx_store_kv");
}


int rx_store_kv_layout(const void* k, const void* v, const rx_kv_layout* lay, const void* loc,
                       int64_t n, int num_kv_heads, int head_dim, int v_head_dim,
                       int64_t k_stride_t, int64_t v_stride_t, int loc_is_i64, int64_t size_limit,
                       int64_t skip_index, int32_t* err_flag, void* stream) {
  RX_RANGE("rx_store_kv_layout");
  RX_REQUIRE(n >= 0, "rx_store_kv_layout: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(k && v && lay && lay->k_buf && lay->v_buf && loc, "rx_store_kv_layout: null pointer");
  RX_REQUIRE(lay->kv_fp8 == 0, "rx_store_kv_layout: fp8 pool -- use rx_store_kv_fp8");
  RX_REQUIRE(num_kv_heads > 0 && head_dim > 0 && v_head_dim > 0 && head_dim % 8 == 0 &&
                 v_head_dim % 8 == 0,
             "rx_store_kv_layout: head dims must be positive multiples of 8 (16-bit elements)");
  const int64_t all = k_stride_t | v_stride_t | lay->k_page_stride | lay->k_tok_stride |
                      lay->k_head_stride | lay->v_page_stride | lay->v_tok_stride | lay->v_head_stride;
  RX_REQUIRE(all % 8 == 0 && lay->page_size >= 1 && size_limit > 0,
             "rx_store_kv_layout: strides must be multiples of 8 elements");
  hipLaunchKernelGGL(store_kv_layout_kernel, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const uint16_t*)k, (const uint16_t*)v,
                     (uint16_t*)lay->k_buf, (uint16_t*)lay->v_buf, loc, n, num_kv_heads, head_dim,
                     v_head_dim, k_stride_t, v_stride_t, lay->page_size, lay->k_page_stride,
                     lay->k_tok_stride, lay->k_head_stride, lay->v_page_stride, lay->v_tok_stride,
                     lay->v_head_stride, loc_is_i64, size_limit, skip_index, err_flag);
  return check_launch("rx_store_kv_layout");
}

int rx_store_kv_fp8(const void* k, const void* v, const rx_kv_layout* lay, const void* loc, int64_t n,
                    int num_kv_heads, int head_dim, int v_head_dim, int64_t k_stride_t,
                    int64_t v_stride_t, int src_dtype, float k_scale, float v_scale, int loc_is_i64,
                    int64_t size_limit, int64_t skip_index, int32_t* err_flag, void* stream) {
  RX_RANGE("rx_store_kv_fp8");
  RX_REQUIRE(n >= 0, "rx_store_kv_fp8: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(k && v && lay && lay->k_buf && lay->v_buf && loc, "rx_store_kv_fp8: null pointer");
  RX_REQUIRE(lay->kv_fp8 == 1, "rx_store_kv_fp8: the layout is not an fp8 pool (kv_fp8 = %d)", lay->kv_fp8);
  RX_REQUIRE(src_dtype == RX_BF16 || src_dtype == RX_F16, "rx_store_kv_fp8: src_dtype %d", src_dtype);
  RX_REQUIRE(num_kv_heads > 0 && head_dim > 0 && v_head_dim > 0 && head_dim % 8 == 0 &&
                 v_head_dim % 8 == 0,
             "rx_store_kv_fp8: head dims must be positive multiples of 8");
  RX_REQUIRE(k_scale > 0.f && v_scale > 0.f, "rx_store_kv_fp8: scales must be > 0");
  const int64_t all = k_stride_t | v_stride_t | lay->k_page_stride | lay->k_tok_stride |
                      lay->k_head_stride | lay->v_page_stride | lay->v_tok_stride | lay->v_head_stride;
  RX_REQUIRE(all % 8 == 0 && lay->page_size >= 1 && size_limit > 0 &&
                 (((uintptr_t)lay->k_buf | (uintptr_t)lay->v_buf) % 8 == 0) &&
                 (((uintptr_t)k | (uintptr_t)v) % 16 == 0),
             "rx_store_kv_fp8: strides must be multiples of 8 elements, sources 16-byte and pool 8-byte aligned");
#define RX_SQ(TT)                                                                                      \
  hipLaunchKernelGGL(store_kv_fp8_kernel<TT>, dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0, \
                     static_cast<hipStream_t>(stream), (const uint16_t*)k, (const uint16_t*)v,         \
                     (uint8_t*)lay->k_buf, (uint8_t*)lay->v_buf, loc, n, num_kv_heads, head_dim,       \
                     v_head_dim, k_stride_t, v_stride_t, lay->page_size, lay->k_page_stride,           \
                     lay->k_tok_stride, lay->k_head_stride, lay->v_page_stride, lay->v_tok_stride,     \
                     lay->v_head_stride, k_scale, v_scale, loc_is_i64, size_limit, skip_index, err_flag)
  if (src_dtype == RX_BF16) RX_SQ(BF16);
  else RX_SQ(F16);
#undef RX_SQ
  return check_launch("rx_store_kv_fp8");
}

int rx_fused_fp8_qkv_kv_cache(const void* q, const void* k, const void* v, void* q_out, const rx_kv_layout* lay,
                              const void* cache_loc, int loc_is_i64, const float* k_scale, const float* v_scale, int64_t n,
                              int q_dim, int num_kv_heads, int head_dim, int v_head_dim, int64_t q_stride_t,
                              int64_t k_stride_t, int64_t v_stride_t, int src_dtype, int64_t size_limit, int32_t* err_flag,
                              void* stream) {
  RX_RANGE("rx_fused_fp8_qkv_kv_cache");
  RX_REQUIRE(n >= 0, "rx_fused_fp8_qkv_kv_cache: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(k && v && lay && lay->k_buf && lay->v_buf && cache_loc, "rx_fused_fp8_qkv_kv_cache: null pointer");
  RX_REQUIRE((q == nullptr) == (q_out == nullptr), "rx_fused_fp8_qkv_kv_cache: q and q_out must both be given or both omitted");
  RX_REQUIRE(lay->kv_fp8 == 1, "rx_fused_fp8_qkv_kv_cache: the layout is not an fp8 pool (kv_fp8 = %d)", lay->kv_fp8);
  RX_REQUIRE(src_dtype == RX_BF16 || src_dtype == RX_F16, "rx_fused_fp8_qkv_kv_cache: src_dtype %d", src_dtype);
  RX_REQUIRE(num_kv_heads > 0 && head_dim > 0 && v_head_dim > 0 && head_dim % 8 == 0 && v_head_dim % 8 == 0,
             "rx_fused_fp8_qkv_kv_cache: head dims must be positive multiples of 8");
  const int64_t all = k_stride_t | v_stride_t | lay->k_page_stride | lay->k_tok_stride | lay->k_head_stride |
                      lay->v_page_stride | lay->v_tok_stride | lay->v_head_stride;
  RX_REQUIRE(all % 8 == 0 && lay->page_size >= 1 && size_limit > 0 &&
                 (((uintptr_t)lay->k_buf | (uintptr_t)lay->v_buf) % 8 == 0) && (((uintptr_t)k | (uintptr_t)v) % 16 == 0),
             "rx_fused_fp8_qkv_kv_cache: strides must be multiples of 8 elements, sources 16-byte and pool 8-byte aligned");
  if (q)
    RX_REQUIRE(q_dim > 0 && q_dim % 8 == 0 && q_stride_t % 8 == 0 && (uintptr_t)q % 16 == 0 && (uintptr_t)q_out % 8 == 0,
               "rx_fused_fp8_qkv_kv_cache: q rows must be multiples of 8 elements, q 16-byte and q_out 8-byte aligned");
#define RX_FQ(TT, QQ)                                                                                               \
  hipLaunchKernelGGL((fused_fp8_qkv_store_kernel<TT, QQ>), dim3(static_cast<unsigned>((n + 3) / 4)), dim3(256), 0,  \
                     static_cast<hipStream_t>(stream), (const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,  \
                     (uint8_t*)q_out, (uint8_t*)lay->k_buf, (uint8_t*)lay->v_buf, cache_loc, k_scale, v_scale, n,   \
                     q_dim, num_kv_heads, head_dim, v_head_dim, q_stride_t, k_stride_t, v_stride_t, lay->page_size, \
                     lay->k_page_stride, lay->k_tok_stride, lay->k_head_stride, lay->v_page_stride,                 \
                     lay->v_tok_stride, lay->v_head_stride, loc_is_i64, size_limit, err_flag)
  if (src_dtype == RX_BF16) { if (q) RX_FQ(BF16, true); else RX_FQ(BF16, false); }
  else { if (q) RX_FQ(F16, true); else RX_FQ(F16, false); }
#undef RX_FQ
  return check_launch("rx_fused_fp8_qkv_kv_cache");
}

int rx_get_mla_kv(const void* kv_buf, int64_t row_stride, int kv_fp8, const void* loc, int loc_is_i64,
                  int64_t n, int nope_cols, int rope_cols, void* nope_out, void* rope_out, int dst_dtype,
                  int64_t size_limit, int32_t* err_flag, void* stream) {
  RX_REQUIRE(n >= 0, "rx_get_mla_kv: n < 0");
  if (n == 0) return RX_OK;
  RX_REQUIRE(kv_buf && loc && nope_out && rope_out, "rx_get_mla_kv: null pointer");
  RX_REQUIRE(dst_dtype == RX_BF16 || dst_dtype == RX_F16, "rx_get_mla_kv: dst_dtype %d", dst_dtype);
  RX_REQUIRE(nope_cols > 0 && rope_cols > 0 && nope_cols % 8 == 0 && rope_cols % 8 == 0 && row_stride % 8 == 0 &&
                 size_limit > 0,
             "rx_get_mla_kv: column counts and the row stride must be positive multiples of 8");
  RX_REQUIRE(((uintptr_t)kv_buf % (kv_fp8 ? 8 : 16)) == 0 && (((uintptr_t)nope_out | (uintptr_t)rope_out) % 16) == 0,
             "rx_get_mla_kv: misaligned buffer");
  const dim3 grid(static_cast<unsigned>((n + 3) / 4)), block(256);
  auto s = static_cast<hipStream_t>(stream);
#define RX_GM(TT, K8)                                                                                   \
  hipLaunchKernelGGL((get_mla_kv_kernel<TT, K8>), grid, block, 0, s, kv_buf, row_stride, loc, loc_is_i64, n, \
                     nope_cols, rope_cols, (uint16_t*)nope_out, (uint16_t*)rope_out, size_limit, err_flag)
  if (dst_dtype == RX_BF16) {
    if (kv_fp8) RX_GM(BF16, true);
    else RX_GM(BF16, false);
  } else {
    if (kv_fp8) RX_GM(F16, true);
    else RX_GM(F16, false);
  }
#undef RX_GM
  return check_launch("rx_get_mla_kv");
}

int rx_merge_state(const void* a, const float* lse_a, const void* b, const float* lse_b, void* out,
                   float* out_lse, int64_t num_tokens, int num_heads, int head_size, int dtype, void* stream) {
  RX_RANGE("rx_merge_state");
  RX_REQUIRE(num_tokens >= 0 && num_heads > 0, "rx_merge_state: bad sizes");
  if (num_tokens == 0) return RX_OK;
  RX_REQUIRE(a && b && lse_a && lse_b && out, "rx_merge_state: null pointer");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_merge_state: dtype %d", dtype);
  RX_REQUIRE(head_size > 0 && head_size % 8 == 0, "rx_merge_state: head_size %d must be a multiple of 8", head_size);
  RX_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "rx_merge_state: 16-byte alignment");
  const int64_t rows = num_tokens * num_heads;
  const unsigned grid = static_cast<unsigned>((rows * (head_size >> 3) + 255) / 256);
  auto s = static_cast<hipStream_t>(stream);
  if (dtype == RX_BF16)
    hipLaunchKernelGGL(merge_state_kernel<BF16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)a, lse_a,
                       (const uint16_t*)b, lse_b, (uint16_t*)out, out_lse, rows, head_size);
  else
    hipLaunchKernelGGL(merge_state_kernel<F16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)a, lse_a,
                       (const uint16_t*)b, lse_b, (uint16_t*)out, out_lse, rows, head_size);
  return check_launch("rx_merge_state");
}

int rx_merge_chunks(const void* o_chunks, const float* lse_chunks, int num_chunks, const void* o_last,
                    const float* lse_last, void* out, float* out_lse, int64_t groups, int rows_per_group,
                    int num_heads, int head_size, int dtype, void* stream) {
  RX_REQUIRE(groups >= 0 && rows_per_group > 0 && num_heads > 0 && num_chunks >= 1, "rx_merge_chunks: bad sizes");
  if (groups == 0) return RX_OK;
  RX_REQUIRE(o_chunks && lse_chunks && out && (!o_last || lse_last), "rx_merge_chunks: null pointer");
  RX_REQUIRE(dtype == RX_BF16 || dtype == RX_F16, "rx_merge_chunks: dtype %d", dtype);
  RX_REQUIRE(head_size > 0 && head_size % 8 == 0, "rx_merge_chunks: head_size %d must be a multiple of 8", head_size);
  RX_REQUIRE((((uintptr_t)o_chunks | (uintptr_t)o_last | (uintptr_t)out) & 15) == 0, "rx_merge_chunks: 16-byte alignment");
  const int64_t n = groups * rows_per_group * num_heads * (head_size >> 3);
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  auto s = static_cast<hipStream_t>(stream);
  if (dtype == RX_BF16)
    hipLaunchKernelGGL(merge_chunks_kernel<BF16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)o_chunks, lse_chunks,
                       num_chunks, (const uint16_t*)o_last, lse_last, (uint16_t*)out, out_lse, groups, rows_per_group,
                       num_heads, head_size);
  else
    hipLaunchKernelGGL(merge_chunks_kernel<F16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)o_chunks, lse_chunks,
                       num_chunks, (const uint16_t*)o_last, lse_last, (uint16_t*)out, out_lse, groups, rows_per_group,
                       num_heads, head_size);
  return check_launch("rx_merge_chunks");
}

int rx_build_unified_kv_indices(const int32_t* prefix_kv_indptr, const void* prefix_kv_indices, int prefix_is_i64,
                                const void* extend_start_loc, int start_is_i64, const void* extend_seq_lens,
                                int lens_is_i64, const void* extend_kv_indices, int extend_is_i64, int bs,
                                int64_t max_tokens_per_request, int32_t* unified_kv_indptr, int64_t* unified_kv_indices,
                                int32_t* prefix_lens, void* stream) {
  RX_RANGE("rx_build_unified_kv_indices");
  RX_REQUIRE(bs >= 0, "rx_build_unified_kv_indices: bs < 0");
  RX_REQUIRE(unified_kv_indptr, "rx_build_unified_kv_indices: unified_kv_indptr is null");
  auto s = static_cast<hipStream_t>(stream);
  if (bs == 0) {
    (void)hipMemsetAsync(unified_kv_indptr, 0, sizeof(int32_t), s);
    return check_launch("rx_build_unified_kv_indices(memset)");
  }
  RX_REQUIRE(prefix_kv_indptr && extend_start_loc && extend_seq_lens && unified_kv_indices,
             "rx_build_unified_kv_indices: null pointer");
  hipLaunchKernelGGL(unified_indptr_kernel, dim3(1), dim3(1024), 0, s, prefix_kv_indptr, extend_seq_lens, lens_is_i64, bs,
                     unified_kv_indptr, prefix_lens);
  // grid.x: 256-slot chunks of the longest request the caller allows for (grid-stride beyond it)
  const int64_t per = max_tokens_per_request > 0 ? max_tokens_per_request : 4096;
  const dim3 grid(static_cast<unsigned>(std::min<int64_t>((per + 255) / 256, 1024)), bs);
#define RX_UC(PT, ET)                                                                                                 \
  hipLaunchKernelGGL((unified_indices_copy_kernel<PT, ET>), grid, dim3(256), 0, s, prefix_kv_indptr, (const PT*)prefix_kv_indices, \
                     extend_start_loc, start_is_i64, (const ET*)extend_kv_indices, unified_kv_indptr, unified_kv_indices)
  if (prefix_is_i64) {
    if (extend_is_i64) RX_UC(int64_t, int64_t);
    else RX_UC(int64_t, int32_t);
  } else {
    if (extend_is_i64) RX_UC(int32_t, int64_t);
    else RX_UC(int32_t, int32_t);
  }
#undef RX_UC
  return check_launch("rx_build_unified_kv_indices");
}

int rx_build_kv_indices(const int32_t* req_to_token, int64_t row_stride,
                        const void* req_pool_indices, int pool_idx_is_i64, const void* lens,
                        int lens_is_i64, const int32_t* kv_start, int32_t* kv_indptr_out,
                        void* kv_indices_out, int out_is_i64, int bs, void* stream) {
  RX_RANGE("rx_build_kv_indices");
  RX_REQUIRE(bs >= 0, "rx_build_kv_indices: bs < 0");
  RX_REQUIRE(kv_indptr_out, "rx_build_kv_indices: kv_indptr_out is null");
  auto s = static_cast<hipStream_t>(stream);
  if (bs == 0) {
    (void)hipMemsetAsync(kv_indptr_out, 0, sizeof(int32_t), s);
    return check_launch("rx_build_kv_indices(memset)");
  }
  RX_REQUIRE(req_to_token && req_pool_indices && lens, "rx_build_kv_indices: null pointer");
  hipLaunchKernelGGL(indptr_scan_kernel, dim3(1), dim3(1024), 0, s, lens, lens_is_i64, bs,
                     kv_indptr_out);
  if (kv_indices_out) {
    // chunks of 1024 tokens; grid.x covers the longest possible row
    const unsigned chunks = static_cast<unsigned>((row_stride + 1023) / 1024);
    const dim3 grid(chunks ? chunks : 1, bs);
    if (out_is_i64)
      hipLaunchKernelGGL(kv_indices_gather_kernel<int64_t>, grid, dim3(256), 0, s, req_to_token,
                         row_stride, req_pool_indices, pool_idx_is_i64, kv_start, kv_indptr_out,
                         (int64_t*)kv_indices_out);
    else
      hipLaunchKernelGGL(kv_indices_gather_kernel<int32_t>, grid, dim3(256), 0, s, req_to_token,
                         row_stride, req_pool_indices, pool_idx_is_i64, kv_start, kv_indptr_out,
                         (int32_t*)kv_indices_out);
  }
  return check_launch("rx_build_kv_indices");
}

int rx_draft_decode_kv_indices(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int pool_idx_is_i64,
                               const void* seq_lens, int seq_lens_is_i64, const void* positions, int positions_is_i64,
                               int num_seqs, int topk, int num_steps, int page_size, void* kv_indices, int kv_indices_is_i64,
                               int64_t kv_indices_stride, int32_t* kv_indptr, int64_t kv_indptr_stride, void* stream) {
  RX_RANGE("rx_draft_decode_kv_indices");
  RX_REQUIRE(num_seqs >= 0 && topk >= 1 && num_steps >= 1 && num_steps <= 256 && page_size >= 1,
             "rx_draft_decode_kv_indices: bad sizes (topk >= 1, 1 <= num_steps <= 256, page_size >= 1)");
  if (num_seqs == 0) return RX_OK;
  RX_REQUIRE(req_to_token && req_pool_indices && seq_lens && positions && kv_indices && kv_indptr,
             "rx_draft_decode_kv_indices: null pointer");
  RX_REQUIRE(kv_indptr_stride >= static_cast<int64_t>(num_seqs) * topk + 1, "rx_draft_decode_kv_indices: kv_indptr rows too short");
  const unsigned chunks = static_cast<unsigned>((row_stride + 1023) / 1024);
  const dim3 grid(static_cast<unsigned>(num_seqs) * topk, chunks ? chunks : 1, num_steps);
  auto s = static_cast<hipStream_t>(stream);
  if (kv_indices_is_i64)
    hipLaunchKernelGGL(rx::draft_decode_kv_indices_kernel<int64_t>, grid, dim3(256), 0, s, req_to_token, row_stride, req_pool_indices,
                       pool_idx_is_i64, seq_lens, seq_lens_is_i64, positions, positions_is_i64, num_seqs, topk, num_steps,
                       page_size, static_cast<int64_t*>(kv_indices), kv_indices_stride, kv_indptr, kv_indptr_stride);
  else
    hipLaunchKernelGGL(rx::draft_decode_kv_indices_kernel<int32_t>, grid, dim3(256), 0, s, req_to_token, row_stride, req_pool_indices,
                       pool_idx_is_i64, seq_lens, seq_lens_is_i64, positions, positions_is_i64, num_seqs, topk, num_steps,
                       page_size, static_cast<int32_t*>(kv_indices), kv_indices_stride, kv_indptr, kv_indptr_stride);
  return check_launch("rx_draft_decode_kv_indices");
}

int rx_num_kv_splits(const void* seq_lens, int seq_lens_is_i64, int num_seq, int num_group,
                     int num_head, int num_kv_head, int max_kv_splits, int device_core_count,
                     int32_t* out, void* stream) {
  RX_RANGE("rx_num_kv_splits");
  RX_REQUIRE(seq_lens && out, "rx_num_kv_splits: null pointer");
  RX_REQUIRE(num_seq > 0 && num_group > 0 && num_head > 0 && num_kv_head > 0 &&
                 max_kv_splits > 0 && num_head % num_kv_head == 0,
             "rx_num_kv_splits: bad sizes");
  hipLaunchKernelGGL(num_kv_splits_kernel, dim3(1), dim3(1024), 0,
                     static_cast<hipStream_t>(stream), seq_lens, seq_lens_is_i64, num_seq,
                     num_group, num_head, num_kv_head, max_kv_splits, device_core_count, out);
  return check_launch("rx_num_kv_splits");
}

namespace rx {
__global__ __launch_bounds__(256) void num_kv_splits_native_kernel(const void* __restrict__ seq_lens, int is64,
                                                                   int bs, int uniform, int min_tokens,
                                                                   int32_t* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= bs) return;
  const int32_t len = static_cast<int32_t>(load_idx(seq_lens, b, is64));
  out[b] = max(1, min(uniform, cdiv32(len, min_tokens)));
}
}  // namespace rx

int rx_num_kv_splits_native(const void* seq_lens, int seq_lens_is_i64, int bs, int wg_per_request, int cu_count,
                            int max_kv_splits, int min_tokens_per_split, int32_t* out, void* stream) {
  RX_RANGE("rx_num_kv_splits_native");
  RX_REQUIRE(bs >= 0, "rx_num_kv_splits_native: bs < 0");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(seq_lens && out, "rx_num_kv_splits_native: null pointer");
  RX_REQUIRE(wg_per_request > 0 && cu_count > 0 && max_kv_splits > 0 && min_tokens_per_split > 0,
             "rx_num_kv_splits_native: bad sizes");
  const int64_t wgs = static_cast<int64_t>(bs) * wg_per_request;
  int uniform = static_cast<int>((cu_count + wgs - 1) / wgs);
  uniform = uniform < 1 ? 1 : (uniform > max_kv_splits ? max_kv_splits : uniform);
  hipLaunchKernelGGL(rx::num_kv_splits_native_kernel, dim3((bs + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seq_lens, seq_lens_is_i64, bs, uniform, min_tokens_per_split,
                     out);
  return check_launch("rx_num_kv_splits_native");
}

namespace rx {
// one block: the batch's token total, then every request's count (and, for a mixed batch, the same again on the
// larger workgroup budget)
__device__ __forceinline__ int64_t balanced_count(int64_t len, int64_t tstar, int cap) {
  int64_t n = 1;
  if (2 * len > 3 * tstar) n = min<int64_t>(cap, (len + tstar - 1) / tstar);  // only what is well above an even share
  return max<int64_t>(n, 1);
}

__global__ __launch_bounds__(1024) void num_kv_splits_balanced_kernel(const void* __restrict__ seq_lens, int is64, int bs,
                                                                      int wg_per_request, int wg_target, int cap,
                                                                      int min_tokens, int wg_mixed, int32_t* __restrict__ out) {
  __shared__ unsigned long long part[16];
  __shared__ unsigned long long total_s;
  __shared__ int flags_s[2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  auto block_sum = [&](unsigned long long acc) {  // every thread gets the block's total
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    __syncthreads();  // part / total_s of the previous use are read by now
    if (lane == 0) part[wid] = acc;
    __syncthreads();
    if (tid == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < 16; ++w) t += part[w];
      total_s = t;
    }
    __syncthreads();
    return total_s;
  };
  auto block_max = [&](unsigned long long acc) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long o = __shfl_xor(acc, d);
      acc = o > acc ? o : acc;
    }
    __syncthreads();
    if (lane == 0) part[wid] = acc;
    __syncthreads();
    if (tid == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < 16; ++w) t = part[w] > t ? part[w] : t;
      total_s = t;
    }
    __syncthreads();
    return total_s;
  };
  if (tid < 2) flags_s[tid] = 0;
  unsigned long long acc = 0, amax = 0, alive = 0;
  for (int i = tid; i < bs; i += 1024) {
    const unsigned long long len = static_cast<unsigned long long>(max<int64_t>(load_idx(seq_lens, i, is64), 0));
    acc += len;
    amax = len > amax ? len : amax;
    alive += len > 0;
  }
  const unsigned long long total = block_sum(acc);
  // tokens one workgroup should carry so that wg_target workgroups share the batch evenly
  const unsigned long long work = total * static_cast<unsigned long long>(wg_per_request);
  if (wg_mixed != 0) {
    // The FILL rule (round 4; a near-uniform batch of about one to three whole-request workgroups per CU).  The decode
    // kernel moves ~27 GB/s per CU however many workgroups the CU holds, so the launch lasts as long as the CU with the
    // most bytes: 1.25 requests per CU unsplit is two requests' time on a quarter of the chip with the rest idle
    // (TP = 8 shard, 320 x 4 k: 153 us; every request cut in 3 -> 3.75 pieces per CU -> 125 us), while a batch of
    // 0.7 - 1 requests per CU is bound by HBM as a whole and every split only adds its merge (256 x 4 k: 100 us
    // whole, 105 cut in two; 192 x 4 k: 76 vs 93 cut in three).  So: nobody is cut at 0.7 - 1 blocks per CU, and
    // between 1 and 3 every request takes the smallest count S <= 6 whose pieces fill >= 85 % of a whole number of
    // rounds of CUs.  tools/decode_sweep.py has the sweep.
    const unsigned long long mx = block_max(amax), live = block_sum(alive);
    const long long cus = max(1, wg_target / 2), blocks = static_cast<long long>(live) * wg_per_request;
    // (0 < wg_mixed <= wg_target: a kernel without the live-pairs grid whose workgroups are latency-bound one by one --
    // the MLA kernels: several of them on a CU do not slow each other, so from 0.8 blocks per CU up the launch is bound
    // by HBM as a whole and nobody is cut, whatever the count: 384 x 4 k fp8 rows 184 us whole, 197 cut in two)
    const bool whole_only = wg_mixed > 0 && wg_mixed <= wg_target;
    if (whole_only) {
      if (live > 0 && 2 * mx * live <= 3 * total && 10 * blocks >= 8 * cus) {
        for (int i = tid; i < bs; i += 1024) out[i] = 1;
        return;
      }
    } else if (live > 0 && 2 * mx * live <= 3 * total && 10 * blocks < 7 * cus) {
      // Fewer than 0.7 blocks per CU: everybody is cut, and the count decides how evenly the pieces cover the chip.  The
      // even share's n = ceil(mean / t*) lands anywhere: 176 requests -> 3 pieces = 528 workgroups, 2.06 per CU: 92 us
      // (4 -> 704 = 2.75 per CU: 77 us); 104 blocks of 8 k tokens 5 -> 108 us (7 -> 92).  When n's workgroups fill < 85 %
      // of whole rounds of CUs, the nearest count (smaller first) that does, with >= 1.5 workgroups per CU in total and
      // pieces of at least min_tokens, replaces it; if there is none, or n fills well, the even share below stands.
      const long long mean = static_cast<long long>(total / live);
      const long long ts = max<long long>(min_tokens, static_cast<long long>((work + wg_target - 1) / wg_target));
      const long long n0 = max<long long>(1, min<long long>(cap, (mean + ts - 1) / ts));
      const long long smax = min<long long>(cap, max<long long>(1, mean / min_tokens));
      auto fills = [&](long long sp) {
        const long long w = blocks * sp, rounds = (w + cus - 1) / cus;
        return 2 * w >= 3 * cus && 100 * w >= 85 * rounds * cus;
      };
      long long S = 0;
      if (!fills(n0)) {
        for (long long dlt = 1; dlt <= 8 && S == 0; ++dlt) {
          if (n0 - dlt >= 2 && fills(n0 - dlt)) S = n0 - dlt;
          else if (n0 + dlt <= smax && fills(n0 + dlt)) S = n0 + dlt;
        }
      }
      if (S > 0) {
        for (int i = tid; i < bs; i += 1024) {
          const int64_t len = max<int64_t>(load_idx(seq_lens, i, is64), 0);
          out[i] = static_cast<int32_t>(max<int64_t>(1, min<int64_t>(S, len / min_tokens)));
        }
        return;
      }
    } else if (live > 0 && 2 * mx * live <= 3 * total && blocks < 3 * cus) {
      long long S = 1;
      if (blocks > cus) {
        long long best = 1, bn = 0, bd = 1;  // best fill so far as the fraction bn / bd
        S = 0;
        for (long long sp = 1; sp <= min(cap, 6); ++sp) {
          const long long rounds = (blocks * sp + cus - 1) / cus;
          if (100 * blocks * sp >= 85 * rounds * cus) {
            S = sp;
            break;
          }
          if (blocks * sp * bd > bn * rounds * cus) {
            best = sp;
            bn = blocks * sp;
            bd = rounds * cus;
          }
        }
        if (S == 0) S = best;
      }
      for (int i = tid; i < bs; i += 1024) {
        const int64_t len = max<int64_t>(load_idx(seq_lens, i, is64), 0);
        out[i] = static_cast<int32_t>(max<int64_t>(1, min<int64_t>(S, len / 256)));  // pieces of at least 256 tokens
      }
      return;
    }
  }
  int64_t tstar = max<int64_t>(min_tokens, static_cast<int64_t>((work + wg_target - 1) / wg_target));
  bool any_split = false, any_whole = false;
  for (int i = tid; i < bs; i += 1024) {
    const int64_t n = balanced_count(max<int64_t>(load_idx(seq_lens, i, is64), 0), tstar, cap);
    out[i] = static_cast<int32_t>(n);
    any_split |= n > 1;
    any_whole |= n == 1;
  }
  if (wg_mixed >= 0 && wg_mixed <= wg_target) return;
  if (any_split) flags_s[0] = 1;
  if (any_whole) flags_s[1] = 1;
  __syncthreads();
  if (!(flags_s[0] && flags_s[1])) return;
  if (wg_mixed < 0) {
    // The ROUNDS rule (for the kernel's usual two workgroups per CU, i.e. also for a graph-replayed step): wg_target
    // slots take the workgroups in launch order, longest pieces first, and a workgroup's rate hardly depends on how many
    // others run -- so when the batch needs R >= 2 rounds of workgroups, the unsplit requests (mean length a) go through
    // in R rounds and a long request's pieces should last exactly as long: p = R a.  (Measured, one 32 k request among
    // 63 of 1 k: 22 pieces of 1.5 k from the even share 94 us, 16 pieces of 2 k = 2 a 80 us.)  With R = 1 everything is
    // resident at once and the even share above is what fills the slots.
    unsigned long long p1 = 0, su = 0, cu = 0;
    for (int i = tid; i < bs; i += 1024) {
      const int64_t len = max<int64_t>(load_idx(seq_lens, i, is64), 0);
      const int64_t n = balanced_count(len, tstar, cap);
      p1 += static_cast<unsigned long long>(n);
      if (n == 1) {
        su += static_cast<unsigned long long>(len);
        cu += 1;
      }
    }
    const unsigned long long tot1 = block_sum(p1) * static_cast<unsigned long long>(wg_per_request);
    su = block_sum(su);
    cu = block_sum(cu);
    if ((tot1 + wg_target - 1) / wg_target < 2) return;
    const int64_t a = max<int64_t>(1, static_cast<int64_t>((su + cu - 1) / cu));
    int64_t p = a;
    for (int R = 1; R <= 4; ++R) {
      p = max<int64_t>(R * a, tstar);  // never finer than the even share (a batch whose unsplit requests are tiny)
      unsigned long long pc = 0;
      for (int i = tid; i < bs; i += 1024) {
        const int64_t len = max<int64_t>(load_idx(seq_lens, i, is64), 0);
        if (balanced_count(len, tstar, cap) > 1) pc += static_cast<unsigned long long>(min<int64_t>(cap, (len + p - 1) / p));
      }
      const unsigned long long tot = (cu + block_sum(pc)) * static_cast<unsigned long long>(wg_per_request);
      if ((tot + wg_target - 1) / wg_target <= static_cast<unsigned long long>(R)) break;
    }
    for (int i = tid; i < bs; i += 1024) {
      const int64_t len = max<int64_t>(load_idx(seq_lens, i, is64), 0);
      if (balanced_count(len, tstar, cap) > 1)
        out[i] = static_cast<int32_t>(max<int64_t>(1, min<int64_t>(cap, (len + p - 1) / p)));
    }
    return;
  }
  // a MIXED batch (some requests cut, some not) is where workgroups differ in size: re-derive the schedule for the
  // budget of the live-pairs grid (three workgroups per CU, all resident: rx_decode_params.split_items), and scale t*
  // up once if rounding up overshoots that budget (a second round of workgroups costs more than coarser pieces)
  tstar = max<int64_t>(min_tokens, static_cast<int64_t>((work + wg_mixed - 1) / wg_mixed));
  unsigned long long pairs = 0;
  for (int i = tid; i < bs; i += 1024)
    pairs += static_cast<unsigned long long>(balanced_count(max<int64_t>(load_idx(seq_lens, i, is64), 0), tstar, cap));
  const unsigned long long wgs = block_sum(pairs) * static_cast<unsigned long long>(wg_per_request);
  if (wgs > static_cast<unsigned long long>(wg_mixed))
    tstar = static_cast<int64_t>((static_cast<unsigned long long>(tstar) * wgs + wg_mixed - 1) / wg_mixed);
  for (int i = tid; i < bs; i += 1024)
    out[i] = static_cast<int32_t>(balanced_count(max<int64_t>(load_idx(seq_lens, i, is64), 0), tstar, cap));
}
}  // namespace rx

namespace rx {
// one block: exclusive scan of the split counts in launch order, then the pairs (rx_decode_params.split_items)
// fix != NULL (rx_split_items_guarded; == splits, writable): a schedule with more live pairs than cap is REPLACED by one
// whole pass per request (every count 1, bs pairs; cap >= bs) and overflow[0] is set -- never a table with pairs missing
__global__ __launch_bounds__(1024) void split_items_kernel(const int32_t* splits, const int32_t* __restrict__ order,
                                                           int bs, int32_t* __restrict__ items, int32_t* __restrict__ count, int cap,
                                                           int32_t* fix, int32_t* __restrict__ overflow) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < bs; c0 += 1024) {
    const int i = c0 + tid;
    const int b = i < bs ? (order ? order[i] : i) : 0;
    const int32_t n = i < bs ? max(splits[b], 1) : 0;
    int32_t x = n;  // inclusive scan inside the wave, then across the 16 waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int32_t off = carry_s;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    const int32_t first = off + x - n;
    for (int32_t s = 0; s < n; ++s) {
      if (first + s < cap) {
        items[2 * (first + s)] = b;
        items[2 * (first + s) + 1] = s;
      }
    }
    __syncthreads();
    if (tid == 1023) carry_s = off + x;
    __syncthreads();
  }
  const int32_t total = carry_s;
  if (fix && total > cap) {  // wave-uniform: carry_s is one LDS word behind the loop's last barrier
    for (int i = tid; i < bs; i += 1024) {
      const int b = order ? order[i] : i;
      fix[b] = 1;
      items[2 * i] = b;
      items[2 * i + 1] = 0;
    }
    if (tid == 0) {
      count[0] = bs;
      if (overflow) overflow[0] = 1;
    }
    return;
  }
  if (tid == 0) count[0] = total;
}
}  // namespace rx

int rx_split_items(const int32_t* num_kv_splits, const int32_t* order, int bs, int32_t* items, int32_t* count, int cap,
                   void* stream) {
  RX_RANGE("rx_split_items");
  RX_REQUIRE(bs >= 0 && cap >= 0, "rx_split_items: negative sizes");
  RX_REQUIRE(count, "rx_split_items: null count");
  if (bs == 0) {
    hipMemsetAsync(count, 0, sizeof(int32_t), static_cast<hipStream_t>(stream));
    return check_launch("rx_split_items");
  }
  RX_REQUIRE(num_kv_splits && (items || cap == 0), "rx_split_items: null pointer");
  hipLaunchKernelGGL(rx::split_items_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), num_kv_splits, order, bs,
                     items, count, cap, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr));
  return check_launch("rx_split_items");
}

int rx_split_items_guarded(int32_t* num_kv_splits, const int32_t* order, int bs, int32_t* items, int32_t* count, int cap,
                           int32_t* overflow, void* stream) {
  RX_RANGE("rx_split_items_guarded");
  RX_REQUIRE(bs >= 0 && cap >= bs, "rx_split_items_guarded: cap must hold one pair per request (cap >= bs)");
  RX_REQUIRE(count, "rx_split_items_guarded: null count");
  if (bs == 0) {
    hipMemsetAsync(count, 0, sizeof(int32_t), static_cast<hipStream_t>(stream));
    return check_launch("rx_split_items_guarded");
  }
  RX_REQUIRE(num_kv_splits && items, "rx_split_items_guarded: null pointer");
  hipLaunchKernelGGL(rx::split_items_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), num_kv_splits, order, bs,
                     items, count, cap, num_kv_splits, overflow);
  return check_launch("rx_split_items_guarded");
}

int rx_num_kv_splits_balanced(const void* seq_lens, int seq_lens_is_i64, int bs, int wg_per_request, int wg_target,
                              int max_kv_splits, int min_tokens_per_split, int wg_target_mixed, int32_t* out,
                              void* stream) {
  RX_RANGE("rx_num_kv_splits_balanced");
  RX_REQUIRE(bs >= 0, "rx_num_kv_splits_balanced: bs < 0");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(seq_lens && out, "rx_num_kv_splits_balanced: null pointer");
  RX_REQUIRE(wg_per_request > 0 && wg_target > 0 && max_kv_splits > 0 && min_tokens_per_split > 0 && wg_target_mixed >= -1,
             "rx_num_kv_splits_balanced: bad sizes");
  hipLaunchKernelGGL(rx::num_kv_splits_balanced_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), seq_lens,
                     seq_lens_is_i64, bs, wg_per_request, wg_target, max_kv_splits, min_tokens_per_split, wg_target_mixed, out);
  return check_launch("rx_num_kv_splits_balanced");
}

int rx_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens, const int64_t* last_loc,
                    const int64_t* free_pages, int64_t* out_indices, int bs, int page_size,
                    void* stream) {
  RX_REQUIRE(bs >= 0 && page_size > 0, "rx_alloc_extend: bad sizes");
  if (bs == 0) return RX_OK;
  // free_pages may be NULL when the free list is empty and no request needs a new page (the caller
  // checks the page budget first, allocator/paged.py:190-196): the kernels read it only for new pages
  RX_REQUIRE(prefix_lens && seq_lens && last_loc && out_indices, "rx_alloc_extend: null pointer");
  hipLaunchKernelGGL(alloc_extend_kernel, dim3(bs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), prefix_lens, seq_lens, last_loc,
                     free_pages, out_indices, page_size);
  return check_launch("rx_alloc_extend");
}

int rx_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc, const int64_t* free_pages,
                    int64_t* out_indices, int bs, int page_size, void* stream) {
  RX_REQUIRE(bs >= 0 && page_size > 0, "rx_alloc_decode: bad sizes");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(seq_lens && last_loc && out_indices, "rx_alloc_decode: null pointer");  // free_pages: see above
  hipLaunchKernelGGL(alloc_decode_kernel, dim3(bs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), seq_lens, last_loc, free_pages,
                     out_indices, page_size);
  return check_launch("rx_alloc_decode");
}

int rx_write_req_to_token(int32_t* req_to_token, int64_t row_stride,
                          const int64_t* req_pool_indices, const int64_t* const* prefix_ptrs,
                          const int64_t* pre_lens, const int64_t* seq_lens,
                          const int64_t* extend_lens, const int64_t* out_cache_loc, int bs,
                          void* stream) {
  RX_REQUIRE(bs >= 0, "rx_write_req_to_token: bs < 0");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(req_to_token && req_pool_indices && pre_lens && seq_lens && extend_lens &&
                 out_cache_loc,
             "rx_write_req_to_token: null pointer");
  hipLaunchKernelGGL(write_req_to_token_kernel, dim3(bs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), req_to_token, row_stride, req_pool_indices,
                     prefix_ptrs, pre_lens, seq_lens, extend_lens, out_cache_loc);
  return check_launch("rx_write_req_to_token");
}

int rx_move_kv(const uint64_t* data_ptrs, const int64_t* row_bytes, int num_bufs,
               const int64_t* tgt_loc, const int64_t* src_loc, int64_t n, void* stream) {
  RX_RANGE("rx_move_kv");
  RX_REQUIRE(num_bufs >= 0 && n >= 0, "rx_move_kv: bad sizes");
  if (num_bufs == 0 || n == 0) return RX_OK;
  RX_REQUIRE(data_ptrs && row_bytes && tgt_loc && src_loc, "rx_move_kv: null pointer");
  hipLaunchKernelGGL(move_kv_kernel, dim3(static_cast<unsigned>((n + 3) / 4), num_bufs),
                     dim3(256), 0, static_cast<hipStream_t>(stream), data_ptrs, row_bytes,
                     tgt_loc, src_loc, n);
  return check_launch("rx_move_kv");
}

int rx_move_kv_layout(const uint64_t* data_ptrs, const int64_t* geom, int num_bufs, int page_size, int num_heads,
                      const int64_t* tgt_loc, const int64_t* src_loc, int64_t n, void* stream) {
  RX_REQUIRE(num_bufs >= 0 && n >= 0 && page_size >= 1 && num_heads >= 1, "rx_move_kv_layout: bad sizes");
  if (num_bufs == 0 || n == 0) return RX_OK;
  RX_REQUIRE(data_ptrs && geom && tgt_loc && src_loc, "rx_move_kv_layout: null pointer");
  hipLaunchKernelGGL(move_kv_layout_kernel, dim3(static_cast<unsigned>((n + 3) / 4), num_bufs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), data_ptrs, geom, page_size, num_heads, tgt_loc, src_loc, n);
  return check_launch("rx_move_kv_layout");
}

namespace rx {
// ---- shared-prefix (cascade) decode plan --------------------------------------------------------------
// Three stream-ordered launches, all integer work on the 4-byte req_to_token table:
//   init    : plan[1] = min(min_b seq_len_b, max_shared)
//   compare : column t of every row against row rpi[0]; a mismatch lowers plan[1] to t (atomicMin)
//   emit    : threshold, chunk boundaries, the shared slot list, kv_start / suffix_lens
__global__ __launch_bounds__(256) void shared_prefix_init_kernel(const void* __restrict__ seq_lens, int sl64, int bs,
                                                                 int32_t max_shared, int32_t* __restrict__ plan) {
  __shared__ int32_t red[4];
  int32_t m = max_shared;
  for (int b = threadIdx.x; b < bs; b += 256) m = min(m, static_cast<int32_t>(load_idx(seq_lens, b, sl64)));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = min(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) plan[1] = max(0, min(min(red[0], red[1]), min(red[2], red[3])));
}

constexpr int kPlanRows = 16;  // rows of the table one workgroup compares
__global__ __launch_bounds__(256) void shared_prefix_compare_kernel(const int32_t* __restrict__ req_to_token,
                                                                    int64_t row_stride,
                                                                    const void* __restrict__ rpi, int rpi64, int bs,
                                                                    int32_t* __restrict__ plan) {
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  // plan[1] only ever decreases while this kernel runs; a stale (larger) bound costs extra compares only
  const int32_t bound = __atomic_load_n(plan + 1, __ATOMIC_RELAXED);
  if (blockIdx.x * 256 >= bound) return;
  const int b0 = 1 + blockIdx.y * kPlanRows;
  const int32_t want = t < bound ? req_to_token[load_idx(rpi, 0, rpi64) * row_stride + t] : 0;
  bool diff = false;
#pragma unroll 4
  for (int i = 0; i < kPlanRows; ++i) {
    const int b = b0 + i;
    if (b < bs && t < bound) diff |= req_to_token[load_idx(rpi, b, rpi64) * row_stride + t] != want;
  }
  // first differing column of this wave, one atomic per wave
  const uint64_t mask = __ballot(diff);
  if (mask != 0 && (threadIdx.x & 63) == 0)
    atomicMin(plan + 1, (t & ~63) + static_cast<int32_t>(__builtin_ctzll(mask)));
}

__global__ __launch_bounds__(256) void shared_prefix_emit_kernel(
    const int32_t* __restrict__ req_to_token, int64_t row_stride, const void* __restrict__ rpi, int rpi64,
    const void* __restrict__ seq_lens, int sl64, int bs, int32_t min_shared, int num_chunks, int chunk_align,
    int32_t* __restrict__ plan, int32_t* __restrict__ chunk_indptr, int32_t* __restrict__ shared_indices,
    int32_t* __restrict__ kv_start, int32_t* __restrict__ suffix_lens) {
  int32_t L = plan[1];
  if (L < min_shared) L = 0;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid == 0) plan[0] = L;
  if (gid <= num_chunks) {
    const int32_t per = cdiv32(cdiv32(L, num_chunks), chunk_align) * chunk_align;
    chunk_indptr[gid] = static_cast<int32_t>(min(static_cast<int64_t>(gid) * per, static_cast<int64_t>(L)));
  }
  if (gid < bs) {
    kv_start[gid] = L;
    suffix_lens[gid] = static_cast<int32_t>(load_idx(seq_lens, gid, sl64)) - L;
  }
  if (gid < L) shared_indices[gid] = req_to_token[load_idx(rpi, 0, rpi64) * row_stride + gid];
}
}  // namespace rx

int rx_shared_prefix_plan(const int32_t* req_to_token, int64_t req_row_stride, const void* req_pool_indices,
                          int req_pool_indices_is_i64, const void* seq_lens, int seq_lens_is_i64, int bs,
                          int32_t max_shared, int32_t min_shared, int num_chunks, int chunk_align,
                          int32_t* plan, int32_t* chunk_indptr, int32_t* shared_indices, int32_t* kv_start,
                          int32_t* suffix_lens, void* stream) {
  RX_REQUIRE(bs >= 0, "rx_shared_prefix_plan: bs < 0");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(req_to_token && req_pool_indices && seq_lens && plan && chunk_indptr && shared_indices && kv_start &&
                 suffix_lens,
             "rx_shared_prefix_plan: null pointer");
  RX_REQUIRE(max_shared >= 0 && min_shared >= 0 && num_chunks >= 1 && chunk_align >= 1 &&
                 max_shared <= req_row_stride,
             "rx_shared_prefix_plan: bad sizes (max_shared=%d min_shared=%d num_chunks=%d chunk_align=%d)",
             max_shared, min_shared, num_chunks, chunk_align);
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rx::shared_prefix_init_kernel, dim3(1), dim3(256), 0, s, seq_lens, seq_lens_is_i64, bs,
                     max_shared, plan);
  if (bs > 1 && max_shared > 0)
    hipLaunchKernelGGL(rx::shared_prefix_compare_kernel,
                       dim3((max_shared + 255) / 256, (bs - 1 + rx::kPlanRows - 1) / rx::kPlanRows), dim3(256), 0, s,
                       req_to_token, req_row_stride, req_pool_indices, req_pool_indices_is_i64, bs, plan);
  const int n = max_shared > bs ? (max_shared > num_chunks + 1 ? max_shared : num_chunks + 1)
                                : (bs > num_chunks + 1 ? bs : num_chunks + 1);
  hipLaunchKernelGGL(rx::shared_prefix_emit_kernel, dim3((n + 255) / 256), dim3(256), 0, s, req_to_token,
                     req_row_stride, req_pool_indices, req_pool_indices_is_i64, seq_lens, seq_lens_is_i64, bs,
                     min_shared, num_chunks, chunk_align, plan, chunk_indptr, shared_indices, kv_start, suffix_lens);
  return check_launch("rx_shared_prefix_plan");
}

namespace rx {
// chunk boundaries of every request's kv list for a split pass: out[b * S + x] = indptr[b] + min(x * per_b, P_b),
// per_b = ceil(ceil(P_b / S) / align) * align, out[bs * S] = indptr[bs]
__global__ __launch_bounds__(256) void chunk_indptr_kernel(const int32_t* __restrict__ indptr, int bs, int S, int align,
                                                           int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > bs * S) return;
  if (i == bs * S) {
    out[i] = indptr[bs];
    return;
  }
  const int b = i / S, x = i % S;
  const int32_t beg = indptr[b], P = indptr[b + 1] - beg;
  const int32_t per = cdiv32(cdiv32(P, S), align) * align;
  out[i] = beg + static_cast<int32_t>(min(static_cast<int64_t>(x) * per, static_cast<int64_t>(P)));
}
}  // namespace rx

int rx_chunk_indptr(const int32_t* kv_indptr, int bs, int num_chunks, int chunk_align, int32_t* out, void* stream) {
  RX_REQUIRE(bs >= 0 && num_chunks >= 1 && chunk_align >= 1, "rx_chunk_indptr: bad sizes");
  RX_REQUIRE(kv_indptr && out, "rx_chunk_indptr: null pointer");
  const int n = bs * num_chunks + 1;
  hipLaunchKernelGGL(rx::chunk_indptr_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     kv_indptr, bs, num_chunks, chunk_align, out);
  return check_launch("rx_chunk_indptr");
}

}  // extern "C"
