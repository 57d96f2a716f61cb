// a5 / §8f-1: the KV slot / page allocator's free list as a DEVICE-RESIDENT structure.
//
// Reference contract (what must stay bit-exact -- it decides which KV page indices a request gets):
//   TokenToKVPoolAllocator        srt/mem_cache/allocator/token.py:27-84
//   PagedTokenToKVPoolAllocator   srt/mem_cache/allocator/paged.py:105-345
//   merge_and_sort_free           srt/mem_cache/allocator/base.py:70-76
// i.e. a FIFO list `free_pages` (alloc takes its head), paged frees go to its FRONT as the sorted set of the freed
// pages (paged.py:261-271: torch.unique -- a host sync -- then cat((ids, free_pages))), token frees to its BACK
// (token.py:66-76), fixed-shape segment frees to the front in piece order (paged.py:273-301), and with need_sort
// frees collect in `release_pages` until free = sort(free + release).
//
// There the list is a torch tensor re-created by slicing / torch.cat on every call and `free` synchronises the
// host through torch.unique.  Here it is a ring of int64 ids in HBM with head / count words next to it; every
// operation is one to three stream-ordered kernels that update the ring in place: no allocation, no host sync, no
// host-visible data dependence (the host keeps a mirror of the counts, exact except after a data-dependent free,
// and only then reads two words back).  The sorted-unique front insert is a flag array indexed by page id plus a
// tiled prefix-scan compaction: O(pages) but sync-free, and the same three kernels implement merge-and-sort.
#include "rx_common.h"

#include <algorithm>

namespace rx {

// state words (int64) of a pool
enum { kFreeHead = 0, kFreeCount = 1, kRelHead = 2, kRelCount = 3, kOom = 4, kBase = 5, kTotal = 6, kOverflow = 7, kStateWords = 8 };
constexpr int kTile = 2048;  // ids per compaction tile: 256 threads x 8 flags

struct PoolArgs {
  int64_t* ring[2];  // 0 = free list, 1 = release list
  int64_t cap;
  uint8_t* flags;    // [num_ids + 1], zero between operations
  int64_t num_ids;   // largest valid id
  int64_t* tiles;    // [n_tiles + 1] compaction scratch
  int64_t* state;
};

__device__ __forceinline__ int64_t wrap(int64_t i, int64_t cap) {
  i %= cap;
  return i < 0 ? i + cap : i;
}

__global__ __launch_bounds__(256) void pool_reset_kernel(PoolArgs p, int64_t first_id, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) p.ring[0][i] = first_id + i;
  if (i == 0) {
    p.state[kFreeHead] = 0;
    p.state[kFreeCount] = n;
    p.state[kRelHead] = 0;
    p.state[kRelCount] = 0;
    p.state[kOom] = 0;
    p.state[kOverflow] = 0;
  }
}

// ring `which` := ids[0..n)
__global__ __launch_bounds__(256) void pool_load_kernel(PoolArgs p, int which, const int64_t* __restrict__ ids,
                                                        int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) p.ring[which][i] = ids[i];
  if (i == 0) {
    p.state[2 * which] = 0;
    p.state[2 * which + 1] = n;
  }
}

// out[0..count) := ring `which` in list order; out_count[0] = count (device word, optional)
__global__ __launch_bounds__(256) void pool_snapshot_kernel(PoolArgs p, int which, int64_t* __restrict__ out,
                                                            int64_t out_cap) {
  const int64_t head = p.state[2 * which], cnt = p.state[2 * which + 1];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < cnt && i < out_cap) out[i] = p.ring[which][wrap(head + i, p.cap)];
}

// take `need` ids off the head of the free list; every block reads the same head, one thread moves it afterwards
__global__ void pool_advance_kernel(PoolArgs p, int64_t need) {
  if (p.state[kFreeCount] >= need) {
    p.state[kFreeHead] = wrap(p.state[kFreeHead] + need, p.cap);
    p.state[kFreeCount] -= need;
  } else {
    p.state[kOom] += 1;  // the host mirror should have refused this call
  }
}

// alloc (token.py:55-64, paged.py:149-170): out[i * page_size + j] = page_i * page_size + j
__global__ __launch_bounds__(256) void pool_alloc_kernel(PoolArgs p, int64_t num_pages, int page_size,
                                                         int64_t* __restrict__ out) {
  if (p.state[kFreeCount] < num_pages) return;
  const int64_t head = p.state[kFreeHead];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= num_pages * page_size) return;
  const int64_t pg = p.ring[0][wrap(head + i / page_size, p.cap)];
  out[i] = pg * page_size + i % page_size;
}

__device__ __forceinline__ int64_t pool_block_sum(int64_t x, int64_t* sh) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  const int64_t r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

// alloc_extend_kernel (kernels/ops/memory/allocator.py:16-95) reading the pages through the ring
__global__ __launch_bounds__(256) void pool_alloc_extend_kernel(PoolArgs p, const int64_t* __restrict__ pre_lens,
                                                                const int64_t* __restrict__ seq_lens,
                                                                const int64_t* __restrict__ last_loc,
                                                                int64_t* __restrict__ out, int page_size,
                                                                int64_t need) {
  __shared__ int64_t sh[4];
  if (p.state[kFreeCount] < need) return;
  const int64_t head = p.state[kFreeHead];
  const int pid = blockIdx.x, tid = threadIdx.x;
  const int64_t ps = page_size;
  int64_t ext_before = 0, pages_before = 0;
  for (int i = tid; i < pid; i += 256) {
    const int64_t s = seq_lens[i], q = pre_lens[i];
    ext_before += s - q;
    pages_before += (s + ps - 1) / ps - (q + ps - 1) / ps;
  }
  const int64_t out_start = pool_block_sum(ext_before, sh);
  const int64_t page_start = pool_block_sum(pages_before, sh);
  const int64_t seq = seq_lens[pid], pre = pre_lens[pid];
  const int64_t new_pages = (seq + ps - 1) / ps - (pre + ps - 1) / ps;
  const int64_t pre_up = (pre + ps - 1) / ps * ps;
  const int64_t n1 = min(seq, pre_up) - pre;        // the rest of the last cached page
  const int64_t ll = last_loc[pid];
  for (int64_t j = tid; j < n1; j += 256) out[out_start + j] = ll + 1 + j;
  if (pre + n1 == seq) return;
  const int64_t n2 = seq / ps * ps - pre_up;        // whole new pages
  for (int64_t j = tid; j < n2; j += 256)
    out[out_start + n1 + j] = p.ring[0][wrap(head + page_start + j / ps, p.cap)] * ps + j % ps;
  if (pre + n1 + n2 == seq) return;
  const int64_t n3 = seq - seq / ps * ps;           // the new partial page
  const int64_t start = p.ring[0][wrap(head + page_start + new_pages - 1, p.cap)];
  for (int64_t j = tid; j < n3; j += 256) out[out_start + n1 + n2 + j] = start * ps + j;
}

// alloc_decode_kernel (allocator.py:98-135)
__global__ __launch_bounds__(256) void pool_alloc_decode_kernel(PoolArgs p, const int64_t* __restrict__ seq_lens,
                                                                const int64_t* __restrict__ last_loc,
                                                                int64_t* __restrict__ out, int page_size,
                                                                int64_t need) {
  __shared__ int64_t sh[4];
  if (p.state[kFreeCount] < need) return;
  const int64_t head = p.state[kFreeHead];
  const int pid = blockIdx.x, tid = threadIdx.x;
  const int64_t ps = page_size;
  int64_t pages_before = 0;
  for (int i = tid; i < pid; i += 256) {
    const int64_t s = seq_lens[i];
    pages_before += (s + ps - 1) / ps - (s - 1 + ps - 1) / ps;
  }
  const int64_t page_start = pool_block_sum(pages_before, sh);
  if (tid != 0) return;
  const int64_t s = seq_lens[pid];
  const int64_t mine = (s + ps - 1) / ps - (s - 1 + ps - 1) / ps;
  out[pid] = (mine == 0) ? last_loc[pid] + 1 : p.ring[0][wrap(head + page_start, p.cap)] * ps;
}

// alloc_for_decode in one launch (srt/mem_cache/allocation.py:539-593): per request read the last slot from its
// req_to_token row, assign the new token's slot (the next slot of that page, or the first slot of a fresh page:
// alloc_decode_kernel's rule; page_size 1: always a fresh id), write it to out AND to the row at position seq_len.
// seq_lens = lengths BEFORE the new token.
__global__ __launch_bounds__(256) void pool_alloc_decode_rows_kernel(PoolArgs p, int32_t* __restrict__ req_to_token,
                                                                     int64_t row_stride,
                                                                     const int64_t* __restrict__ req_pool_indices,
                                                                     const int64_t* __restrict__ seq_lens,
                                                                     int64_t* __restrict__ out, int page_size,
                                                                     int64_t need) {
  __shared__ int64_t sh[4];
  if (p.state[kFreeCount] < need) return;
  const int64_t head = p.state[kFreeHead];
  const int pid = blockIdx.x, tid = threadIdx.x;
  const int64_t ps = page_size;
  int64_t pages_before = 0;
  for (int i = tid; i < pid; i += 256) pages_before += (seq_lens[i] % ps == 0) ? 1 : 0;  // length before % ps == 0: new page
  const int64_t page_start = pool_block_sum(pages_before, sh);
  if (tid != 0) return;
  const int64_t s = seq_lens[pid];
  int32_t* row = req_to_token + req_pool_indices[pid] * row_stride;
  const int64_t loc = (s % ps == 0) ? p.ring[0][wrap(head + page_start, p.cap)] * ps
                                    : static_cast<int64_t>(row[s - 1]) + 1;
  out[pid] = loc;
  row[s] = static_cast<int32_t>(loc);
}

// alloc_for_extend in one launch (srt/mem_cache/allocation.py:303-403: build last_loc from the cached prefixes,
// alloc_extend / alloc_token_slots, write_cache_indices): per request read the last cached slot through the request's
// prefix pointer, assign the new tokens' slots (alloc_extend_kernel's three parts; page_size 1: one fresh id per token),
// write them to out AND -- with the cached prefix slots in front -- to the request's req_to_token row.
// table: DEVICE int64 [4, bs] = req_pool_idx | prefix_len | seq_len | device address of the request's prefix slots
// (int64[prefix_len], 0 when prefix_len is 0): ONE host-to-device copy of a packed host table feeds the call.
__global__ __launch_bounds__(256) void pool_alloc_extend_rows_kernel(PoolArgs p, int32_t* __restrict__ req_to_token,
                                                                     int64_t row_stride, const int64_t* __restrict__ table,
                                                                     int bs, int64_t* __restrict__ out, int page_size,
                                                                     int64_t need) {
  __shared__ int64_t sh[4];
  if (p.state[kFreeCount] < need) return;
  const int64_t head = p.state[kFreeHead];
  const int pid = blockIdx.x, tid = threadIdx.x;
  const int64_t ps = page_size;
  const int64_t* pre_lens = table + bs;
  const int64_t* seq_lens = table + 2 * bs;
  int64_t ext_before = 0, pages_before = 0;
  for (int i = tid; i < pid; i += 256) {
    const int64_t s = seq_lens[i], q = pre_lens[i];
    ext_before += s - q;
    pages_before += (s + ps - 1) / ps - (q + ps - 1) / ps;
  }
  const int64_t out_start = pool_block_sum(ext_before, sh);
  const int64_t page_start = pool_block_sum(pages_before, sh);
  const int64_t seq = seq_lens[pid], pre = pre_lens[pid];
  const int64_t* prefix = reinterpret_cast<const int64_t*>(static_cast<uintptr_t>(table[3 * bs + pid]));
  int32_t* row = req_to_token + table[pid] * row_stride;
  for (int64_t j = tid; j < pre; j += 256) row[j] = static_cast<int32_t>(prefix[j]);  // write_cache_indices, prefix part
  const int64_t new_pages = (seq + ps - 1) / ps - (pre + ps - 1) / ps;
  const int64_t pre_up = (pre + ps - 1) / ps * ps;
  const int64_t n1 = min(seq, pre_up) - pre;        // the rest of the last cached page
  const int64_t ll = (pre > 0) ? prefix[pre - 1] : -1;
  for (int64_t j = tid; j < n1; j += 256) {
    out[out_start + j] = ll + 1 + j;
    row[pre + j] = static_cast<int32_t>(ll + 1 + j);
  }
  if (pre + n1 == seq) return;
  const int64_t n2 = seq / ps * ps - pre_up;        // whole new pages
  for (int64_t j = tid; j < n2; j += 256) {
    const int64_t loc = p.ring[0][wrap(head + page_start + j / ps, p.cap)] * ps + j % ps;
    out[out_start + n1 + j] = loc;
    row[pre + n1 + j] = static_cast<int32_t>(loc);
  }
  if (pre + n1 + n2 == seq) return;
  const int64_t n3 = seq - seq / ps * ps;           // the new partial page
  const int64_t start = p.ring[0][wrap(head + page_start + new_pages - 1, p.cap)];
  for (int64_t j = tid; j < n3; j += 256) {
    out[out_start + n1 + n2 + j] = start * ps + j;
    row[pre + n1 + n2 + j] = static_cast<int32_t>(start * ps + j);
  }
}

// list `which` := list + ids   (token.py:66-76: free_pages = cat((free_pages, free_index)))
__global__ __launch_bounds__(256) void pool_append_kernel(PoolArgs p, int which, const int64_t* __restrict__ ids,
                                                          int64_t n) {
  const int64_t tail = p.state[2 * which] + p.state[2 * which + 1];
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) p.ring[which][wrap(tail + i, p.cap)] = ids[i];
}
// kOverflow: the two lists together never hold more than the pool's num_ids ids (ring capacity num_ids + 1).  A double free, or a
// free_segment whose page was also passed to free(), pushes count + n past it: the ring wraps over live entries and
// the same page would be handed out twice.  The writes have happened by the time the count moves, so this is a
// detector, not a guard: the word is surfaced by the host's next read-back (allocator._sync_counts raises).
__global__ void pool_grow_kernel(PoolArgs p, int which, int64_t n, int front) {
  if (p.state[kFreeCount] + p.state[kRelCount] + n > p.num_ids) p.state[kOverflow] += 1;
  if (front) p.state[2 * which] = wrap(p.state[2 * which] - n, p.cap);
  p.state[2 * which + 1] += n;
}

// list `which` := reps + list, reps = [idx[0]] (if has_first) + idx[start::stride], each // page_size
// (free_segment, paged.py:273-301: page representatives are stride slices of the freed run)
__global__ __launch_bounds__(256) void pool_prepend_strided_kernel(PoolArgs p, int which,
                                                                   const int64_t* __restrict__ idx, int64_t n_idx,
                                                                   int has_first, int64_t start, int64_t stride,
                                                                   int page_size, int64_t k) {
  const int64_t new_head = p.state[2 * which] - k;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= k) return;
  int64_t src;
  if (has_first) src = (i == 0) ? 0 : start + (i - 1) * stride;
  else src = start + i * stride;
  if (src < n_idx) p.ring[which][wrap(new_head + i, p.cap)] = idx[src] / page_size;
}

// ---- sorted-unique insert and merge-sort: flags by id + tiled compaction -------------------------------------
__global__ __launch_bounds__(256) void pool_mark_kernel(PoolArgs p, const int64_t* __restrict__ idx, int64_t n,
                                                        int page_size) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t id = idx[i] / page_size;
  if (id >= 0 && id <= p.num_ids) p.flags[id] = 1;
}
__global__ __launch_bounds__(256) void pool_mark_ring_kernel(PoolArgs p, int which) {
  const int64_t head = p.state[2 * which], cnt = p.state[2 * which + 1];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < cnt;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t id = p.ring[which][wrap(head + i, p.cap)];
    if (id >= 0 && id <= p.num_ids) p.flags[id] = 1;
  }
}

__device__ __forceinline__ int tile_flags8(const PoolArgs& p, int64_t base, uint8_t (&f)[8]) {
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f[j] = (base + j <= p.num_ids) ? p.flags[base + j] : 0;
    c += f[j] != 0;
  }
  return c;
}

__global__ __launch_bounds__(256) void pool_tile_count_kernel(PoolArgs p) {
  __shared__ int64_t sh[4];
  uint8_t f[8];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + threadIdx.x * 8;
  const int64_t c = pool_block_sum(tile_flags8(p, base, f), sh);
  if (threadIdx.x == 0) p.tiles[blockIdx.x] = c;
}

// exclusive scan of the tile counts (one block), then the list bookkeeping:
//   mode 0: the marked ids go in FRONT of list `which`      (paged.py:261-271 / 308-312)
//   mode 1: free := the marked ids (all of free + release, ascending), release := empty   (base.py:70-76)
__global__ __launch_bounds__(256) void pool_tile_scan_kernel(PoolArgs p, int64_t n_tiles, int mode, int which) {
  __shared__ int64_t sh[4];
  __shared__ int64_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < n_tiles; c0 += 256) {
    const int64_t i = c0 + threadIdx.x;
    const int64_t v = i < n_tiles ? p.tiles[i] : 0;
    // inclusive scan inside the wave, then across the four waves
    int64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t y = __shfl_up(x, d);
      if ((threadIdx.x & 63) >= d) x += y;
    }
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    int64_t wave_off = 0;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) wave_off += sh[w];
    const int64_t carry = carry_s;
    if (i < n_tiles) p.tiles[i] = carry + wave_off + x - v;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = carry + wave_off + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int64_t total = carry_s;
    p.state[kTotal] = total;
    if (mode == 0) {
      if (p.state[kFreeCount] + p.state[kRelCount] + total > p.num_ids) p.state[kOverflow] += 1;
      const int64_t nh = wrap(p.state[2 * which] - total, p.cap);
      p.state[2 * which] = nh;
      p.state[2 * which + 1] += total;
      p.state[kBase] = nh;
    } else {
      p.state[kFreeHead] = 0;
      p.state[kFreeCount] = total;
      p.state[kRelHead] = 0;
      p.state[kRelCount] = 0;
      p.state[kBase] = 0;
    }
  }
}

__global__ __launch_bounds__(256) void pool_tile_scatter_kernel(PoolArgs p, int which) {
  __shared__ int64_t sh[4];
  uint8_t f[8];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTile + threadIdx.x * 8;
  const int c = tile_flags8(p, base, f);
  // exclusive scan of the per-thread counts over the block
  int64_t x = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t y = __shfl_up(x, d);
    if ((threadIdx.x & 63) >= d) x += y;
  }
  if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = x;
  __syncthreads();
  int64_t off = x - c;
  for (int w = 0; w < (threadIdx.x >> 6); ++w) off += sh[w];
  int64_t dst = p.state[kBase] + p.tiles[blockIdx.x] + off;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (f[j]) {
      p.ring[which][wrap(dst++, p.cap)] = base + j;
      p.flags[base + j] = 0;
    }
  }
}

static PoolArgs pool_args(const rx_pool_desc* d) {
  PoolArgs p;
  p.ring[0] = d->free_ring;
  p.ring[1] = d->release_ring;
  p.cap = d->capacity;
  p.flags = d->flags;
  p.num_ids = d->num_ids;
  p.tiles = d->tile_scratch;
  p.state = d->state;
  return p;
}
static inline unsigned blocks_for(int64_t n) { return static_cast<unsigned>(n > 0 ? (n + 255) / 256 : 1); }
static inline int64_t pool_tiles(const rx_pool_desc* d) { return (d->num_ids + 1 + kTile - 1) / kTile; }

}  // namespace rx

using namespace rx;

#define RX_POOL_CHECK(d)                                                                                        \
  RX_REQUIRE((d) && (d)->free_ring && (d)->state && (d)->capacity > 0 && (d)->num_ids >= 0, "rx_pool: bad descriptor")

extern "C" {

int64_t rx_pool_tile_scratch_len(int64_t num_ids) { return num_ids >= 0 ? (num_ids + 1 + kTile - 1) / kTile + 1 : -1; }
int rx_pool_state_words(void) { return kStateWords; }

int rx_pool_reset(const rx_pool_desc* d, int64_t first_id, int64_t n, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(n >= 0 && n <= d->capacity && first_id >= 0 && first_id + n - 1 <= d->num_ids,
             "rx_pool_reset: ids [%lld, %lld) do not fit capacity %lld / num_ids %lld", (long long)first_id,
             (long long)(first_id + n), (long long)d->capacity, (long long)d->num_ids);
  hipLaunchKernelGGL(pool_reset_kernel, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pool_args(d), first_id, n);
  return check_launch("rx_pool_reset");
}

int rx_pool_load(const rx_pool_desc* d, int which, const int64_t* ids, int64_t n, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE((which == 0 || (which == 1 && d->release_ring)) && n >= 0 && n <= d->capacity && (ids || n == 0),
             "rx_pool_load: bad arguments");
  hipLaunchKernelGGL(pool_load_kernel, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pool_args(d), which, ids, n);
  return check_launch("rx_pool_load");
}

int rx_pool_snapshot(const rx_pool_desc* d, int which, int64_t* out, int64_t out_cap, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE((which == 0 || (which == 1 && d->release_ring)) && out_cap >= 0 && (out || out_cap == 0),
             "rx_pool_snapshot: bad arguments");
  if (out_cap == 0) return RX_OK;
  hipLaunchKernelGGL(pool_snapshot_kernel, dim3(blocks_for(out_cap)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pool_args(d), which, out, out_cap);
  return check_launch("rx_pool_snapshot");
}

int rx_pool_alloc(const rx_pool_desc* d, int64_t num_pages, int page_size, int64_t* out, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(num_pages >= 0 && page_size >= 1 && (out || num_pages == 0), "rx_pool_alloc: bad arguments");
  if (num_pages == 0) return RX_OK;
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_alloc_kernel, dim3(blocks_for(num_pages * page_size)), dim3(256), 0, s, pool_args(d),
                     num_pages, page_size, out);
  hipLaunchKernelGGL(pool_advance_kernel, dim3(1), dim3(1), 0, s, pool_args(d), num_pages);
  return check_launch("rx_pool_alloc");
}

int rx_pool_alloc_extend(const rx_pool_desc* d, const int64_t* prefix_lens, const int64_t* seq_lens,
                         const int64_t* last_loc, int64_t* out_indices, int bs, int page_size,
                         int64_t num_new_pages, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(bs >= 0 && page_size >= 1 && num_new_pages >= 0, "rx_pool_alloc_extend: bad sizes");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(prefix_lens && seq_lens && last_loc && out_indices, "rx_pool_alloc_extend: null pointer");
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_alloc_extend_kernel, dim3(bs), dim3(256), 0, s, pool_args(d), prefix_lens, seq_lens,
                     last_loc, out_indices, page_size, num_new_pages);
  if (num_new_pages > 0) hipLaunchKernelGGL(pool_advance_kernel, dim3(1), dim3(1), 0, s, pool_args(d), num_new_pages);
  return check_launch("rx_pool_alloc_extend");
}

int rx_pool_alloc_decode(const rx_pool_desc* d, const int64_t* seq_lens, const int64_t* last_loc,
                         int64_t* out_indices, int bs, int page_size, int64_t num_new_pages, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(bs >= 0 && page_size >= 1 && num_new_pages >= 0, "rx_pool_alloc_decode: bad sizes");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(seq_lens && last_loc && out_indices, "rx_pool_alloc_decode: null pointer");
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_alloc_decode_kernel, dim3(bs), dim3(256), 0, s, pool_args(d), seq_lens, last_loc,
                     out_indices, page_size, num_new_pages);
  if (num_new_pages > 0) hipLaunchKernelGGL(pool_advance_kernel, dim3(1), dim3(1), 0, s, pool_args(d), num_new_pages);
  return check_launch("rx_pool_alloc_decode");
}

int rx_pool_alloc_decode_rows(const rx_pool_desc* d, int32_t* req_to_token, int64_t row_stride,
                              const int64_t* req_pool_indices, const int64_t* seq_lens, int64_t* out_indices, int bs,
                              int page_size, int64_t num_new_pages, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(bs >= 0 && page_size >= 1 && num_new_pages >= 0 && row_stride > 0, "rx_pool_alloc_decode_rows: bad sizes");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(req_to_token && req_pool_indices && seq_lens && out_indices, "rx_pool_alloc_decode_rows: null pointer");
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_alloc_decode_rows_kernel, dim3(bs), dim3(256), 0, s, pool_args(d), req_to_token, row_stride,
                     req_pool_indices, seq_lens, out_indices, page_size, num_new_pages);
  if (num_new_pages > 0) hipLaunchKernelGGL(pool_advance_kernel, dim3(1), dim3(1), 0, s, pool_args(d), num_new_pages);
  return check_launch("rx_pool_alloc_decode_rows");
}

int rx_pool_alloc_extend_rows(const rx_pool_desc* d, int32_t* req_to_token, int64_t row_stride, const int64_t* table,
                              int64_t* out_indices, int bs, int page_size, int64_t num_new_pages, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(bs >= 0 && page_size >= 1 && num_new_pages >= 0 && row_stride > 0, "rx_pool_alloc_extend_rows: bad sizes");
  if (bs == 0) return RX_OK;
  RX_REQUIRE(req_to_token && table && out_indices, "rx_pool_alloc_extend_rows: null pointer");
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_alloc_extend_rows_kernel, dim3(bs), dim3(256), 0, s, pool_args(d), req_to_token, row_stride,
                     table, bs, out_indices, page_size, num_new_pages);
  if (num_new_pages > 0) hipLaunchKernelGGL(pool_advance_kernel, dim3(1), dim3(1), 0, s, pool_args(d), num_new_pages);
  return check_launch("rx_pool_alloc_extend_rows");
}

int rx_pool_append(const rx_pool_desc* d, int which, const int64_t* ids, int64_t n, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE((which == 0 || (which == 1 && d->release_ring)) && n >= 0 && (ids || n == 0), "rx_pool_append: bad arguments");
  if (n == 0) return RX_OK;
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_append_kernel, dim3(blocks_for(n)), dim3(256), 0, s, pool_args(d), which, ids, n);
  hipLaunchKernelGGL(pool_grow_kernel, dim3(1), dim3(1), 0, s, pool_args(d), which, n, 0);
  return check_launch("rx_pool_append");
}

int rx_pool_prepend_strided(const rx_pool_desc* d, int which, const int64_t* idx, int64_t n_idx, int has_first,
                            int64_t start, int64_t stride, int page_size, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE((which == 0 || (which == 1 && d->release_ring)) && n_idx >= 0 && start >= 0 && stride >= 1 &&
                 page_size >= 1 && (idx || n_idx == 0),
             "rx_pool_prepend_strided: bad arguments");
  const int64_t k = (has_first && n_idx > 0 ? 1 : 0) + (start < n_idx ? (n_idx - start + stride - 1) / stride : 0);
  if (k == 0) return RX_OK;
  auto s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pool_prepend_strided_kernel, dim3(blocks_for(k)), dim3(256), 0, s, pool_args(d), which, idx,
                     n_idx, has_first && n_idx > 0 ? 1 : 0, start, stride, page_size, k);
  hipLaunchKernelGGL(pool_grow_kernel, dim3(1), dim3(1), 0, s, pool_args(d), which, k, 1);
  return check_launch("rx_pool_prepend_strided");
}

int rx_pool_mark(const rx_pool_desc* d, const int64_t* idx, int64_t n, int page_size, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(d->flags && n >= 0 && page_size >= 1 && (idx || n == 0), "rx_pool_mark: bad arguments");
  if (n == 0) return RX_OK;
  hipLaunchKernelGGL(pool_mark_kernel, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pool_args(d), idx, n, page_size);
  return check_launch("rx_pool_mark");
}

int rx_pool_flush_marks(const rx_pool_desc* d, int which, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(d->flags && d->tile_scratch && (which == 0 || (which == 1 && d->release_ring)),
             "rx_pool_flush_marks: bad arguments");
  auto s = static_cast<hipStream_t>(stream);
  const int64_t nt = pool_tiles(d);
  hipLaunchKernelGGL(pool_tile_count_kernel, dim3(static_cast<unsigned>(nt)), dim3(256), 0, s, pool_args(d));
  hipLaunchKernelGGL(pool_tile_scan_kernel, dim3(1), dim3(256), 0, s, pool_args(d), nt, 0, which);
  hipLaunchKernelGGL(pool_tile_scatter_kernel, dim3(static_cast<unsigned>(nt)), dim3(256), 0, s, pool_args(d), which);
  return check_launch("rx_pool_flush_marks");
}

int rx_pool_merge_sort(const rx_pool_desc* d, void* stream) {
  RX_POOL_CHECK(d);
  RX_REQUIRE(d->flags && d->tile_scratch && d->release_ring, "rx_pool_merge_sort: bad arguments");
  auto s = static_cast<hipStream_t>(stream);
  const int64_t nt = pool_tiles(d);
  const unsigned mb = static_cast<unsigned>(std::min<int64_t>(1024, (d->capacity + 255) / 256));
  hipLaunchKernelGGL(pool_mark_ring_kernel, dim3(mb), dim3(256), 0, s, pool_args(d), 0);
  hipLaunchKernelGGL(pool_mark_ring_kernel, dim3(mb), dim3(256), 0, s, pool_args(d), 1);
  hipLaunchKernelGGL(pool_tile_count_kernel, dim3(static_cast<unsigned>(nt)), dim3(256), 0, s, pool_args(d));
  hipLaunchKernelGGL(pool_tile_scan_kernel, dim3(1), dim3(256), 0, s, pool_args(d), nt, 1, 0);
  hipLaunchKernelGGL(pool_tile_scatter_kernel, dim3(static_cast<unsigned>(nt)), dim3(256), 0, s, pool_args(d), 0);
  return check_launch("rx_pool_merge_sort");
}

}  // extern "C"
