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
    (from RadixCache.match_prefix); a request without a row gets one."""
    dev = req_to_token_pool.device
    bs = len(reqs)
    need_rows = [r for r in reqs if r.req_pool_idx is None]
    rows = req_to_token_pool.alloc(len(need_rows))
    if rows is None:
        raise RuntimeError("alloc_req_slots: req_to_token_pool exhausted")
    for r, row in zip(need_rows, rows):
        r.req_pool_idx = row
    prefix_cpu = torch.tensor(list(prefix_lens), dtype=torch.int64)
    seq_cpu = torch.tensor(list(seq_lens), dtype=torch.int64)
    ext_cpu = seq_cpu - prefix_cpu
    prefix_d, seq_d, ext_d = prefix_cpu.to(dev), seq_cpu.to(dev), ext_cpu.to(dev)
    rpi = torch.tensor([r.req_pool_idx for r in reqs], dtype=torch.int64, device=dev)
    n_ext = int(ext_cpu.sum())
    ps = allocator.page_size
    prefix_tensors = [
        (r.prefix_indices.to(torch.int64) if r.prefix_indices is not None and p > 0
         else torch.empty(0, dtype=torch.int64, device=dev))
        for r, p in zip(reqs, prefix_lens)]
    if ps == 1:
        out_cache_loc = alloc_token_slots(tree_cache, allocator, n_ext)
    else:
        last_loc = torch.cat([t[-1:] if len(t) > 0 else torch.full((1,), -1, dtype=torch.int64, device=dev)
                              for t in prefix_tensors])
        evict_from_tree_cache(tree_cache, allocator, n_ext + bs * ps)
        out_cache_loc = allocator.alloc_extend(prefix_d, prefix_cpu, seq_d, seq_cpu, last_loc, n_ext)
        if out_cache_loc is None:
            raise RuntimeError(f"Prefill out of memory: need {n_ext} tokens, "
                               f"available {allocator.available_size()}")
    keep = [t.contiguous() for t in prefix_tensors]
    ptrs = torch.tensor([t.data_ptr() if t.numel() else 0 for t in keep], dtype=torch.int64).to(dev)
    ops.write_req_to_token(req_to_token_pool.req_to_token, rpi, ptrs, prefix_d, seq_d, ext_d,
                           out_cache_loc)
    # `keep` is read through raw pointers by a kernel on the current stream; the caching allocator
    # only re-uses that memory for later work on the same stream, so dropping it here is safe.
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
