"""Scheduler-side slot allocation: the producer of ``out_cache_loc`` and of the req_to_token
rows the attention kernels read (srt/mem_cache/allocation.py: alloc_token_slots / evict-on-demand
:137-232, alloc_for_extend :303-403, alloc_for_decode :539-593, write_cache_indices :55-101).

Slots come from our allocators (HIP alloc kernels), rows are written by rx_write_req_to_token,
cached prefixes come from the native radix tree; when the allocator runs dry the tree is asked
to evict (the reference's evict_from_tree_cache)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import ops
from .radix_cache import EvictParams, Req


def evict_from_tree_cache(tree_cache, allocator, num_tokens: int):
    if tree_cache is None:
        return
    if allocator.available_size() < num_tokens:
        tree_cache.evict(EvictParams(num_tokens=num_tokens - allocator.available_size()))


def alloc_token_slots(tree_cache, allocator, num_tokens: int) -> torch.Tensor:
    evict_from_tree_cache(tree_cache, allocator, num_tokens)
    out = allocator.alloc(num_tokens)
    if out is None:
        raise RuntimeError(f"Out of memory: need {num_tokens} tokens, "
                           f"available {allocator.available_size()}")
    return out


def alloc_for_extend(reqs: Sequence[Req], prefix_lens: Sequence[int], seq_lens: Sequence[int],
                     req_to_token_pool, allocator, tree_cache=None):
    """Returns (out_cache_loc int64[sum extend], req_pool_indices int64[bs] on device).
    ``reqs[i].prefix_indices`` holds the cached slots of the first prefix_lens[i] tokens
    (from RadixCache.match_prefix); a request without a row gets one.

    The device side is ONE host-to-device copy of a packed [4, bs] table (row, prefix_len, seq_len, prefix pointer) and
    ONE native call (rx_pool_alloc_extend_rows: last_loc gather + alloc_extend / alloc_token_slots + write_cache_indices,
    srt/mem_cache/allocation.py:303-403) -- as alloc_for_decode is one call."""
    dev = req_to_token_pool.device
    bs = len(reqs)
    need_rows = [r for r in reqs if r.req_pool_idx is None]
    rows = req_to_token_pool.alloc(len(need_rows))
    if rows is None:
        raise RuntimeError("alloc_req_slots: req_to_token_pool exhausted")
    for r, row in zip(need_rows, rows):
        r.req_pool_idx = row
    keep = []  # int64 prefix tensors, read through raw pointers by a kernel on the current stream: the caching allocator
    ptrs = []  # only re-uses that memory for later work on the same stream, so dropping them after the call is safe
    for r, p in zip(reqs, prefix_lens):
        t = r.prefix_indices
        if t is None or p == 0:
            ptrs.append(0)
            continue
        if t.dtype != torch.int64 or not t.is_contiguous():
            t = t.to(torch.int64).contiguous()
        if t.numel() < p:
            raise ValueError("alloc_for_extend: prefix_indices shorter than prefix_len")
        keep.append(t)
        ptrs.append(t.data_ptr())
    table = torch.tensor([[r.req_pool_idx for r in reqs], list(prefix_lens), list(seq_lens), ptrs], dtype=torch.int64)
    n_ext = int((table[2] - table[1]).sum())
    ps = allocator.page_size
    evict_from_tree_cache(tree_cache, allocator, n_ext if ps == 1 else n_ext + bs * ps)
    out_cache_loc = allocator.alloc_extend_rows(req_to_token_pool.req_to_token, table, n_ext)
    if out_cache_loc is None:
        raise RuntimeError(f"{'Out of memory' if ps == 1 else 'Prefill out of memory'}: need {n_ext} tokens, "
                           f"available {allocator.available_size()}")
    rpi = table[0].to(dev, non_blocking=True)
    del keep
    return out_cache_loc, rpi


def alloc_for_decode(req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, seq_lens_cpu: torch.Tensor,
                     req_to_token_pool, allocator, tree_cache=None, token_per_req: int = 1):
    """seq_lens are the lengths BEFORE the new token; returns out_cache_loc int64[bs] and writes
    req_to_token[req, seq_len] = loc (allocation.py:578-580).  The last-slot gather, the allocation and the row
    write are one kernel on the device-resident free list (rx_pool_alloc_decode_rows)."""
    assert token_per_req == 1
    bs = seq_lens.shape[0]
    evict_from_tree_cache(tree_cache, allocator, bs * allocator.page_size)
    out_cache_loc = allocator.alloc_decode_rows(req_to_token_pool.req_to_token, req_pool_indices, seq_lens,
                                                seq_lens_cpu)
    if out_cache_loc is None:
        raise RuntimeError(f"Decode out of memory: available {allocator.available_size()}")
    return out_cache_loc
