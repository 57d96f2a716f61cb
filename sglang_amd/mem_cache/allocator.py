"""Slot / page allocators with the observable contract of the reference's
TokenToKVPoolAllocator (srt/mem_cache/allocator/token.py:27-84) and
PagedTokenToKVPoolAllocator (allocator/paged.py:105-345): same free-list order (FIFO head
slice on alloc, LIFO prepend on paged free, optional sort-merge, slot/page 0 reserved), so
the KV page indices they hand out are bit-identical to the reference's.

The per-request index arithmetic of alloc_extend / alloc_decode runs in the HIP kernels
rx_alloc_extend / rx_alloc_decode (include/radix_hip.h); list bookkeeping is torch slicing.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import ops


class BaseTokenToKVPoolAllocator:
    """allocator/base.py:27-134."""

    def __init__(self, size: int, page_size: int, dtype, device, kvcache=None,
                 need_sort: bool = False):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        self.device = device
        self._kvcache = kvcache
        self.need_sort = need_sort
        self.free_pages: torch.Tensor = None
        self.release_pages: torch.Tensor = None
        self.is_not_in_free_group = True
        self.free_group: List[torch.Tensor] = []

    @property
    def size_full(self):
        return self.size

    def available_size(self):
        return (len(self.free_pages) + len(self.release_pages)) * self.page_size

    def get_kvcache(self):
        return self._kvcache

    def free_group_begin(self):
        self.is_not_in_free_group = False
        self.free_group = []

    def free_group_end(self):
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(torch.cat(self.free_group))

    def merge_and_sort_free(self):
        if len(self.release_pages) > 0:
            self.free_pages = torch.cat((self.free_pages, self.release_pages))
            self.free_pages, _ = torch.sort(self.free_pages)
            self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def alloc_extend(self, *args, **kwargs):
        raise NotImplementedError("alloc_extend is only for paged allocator")

    def alloc_decode(self, *args, **kwargs):
        raise NotImplementedError("alloc_decode is only for paged allocator")

    def free_segment(self, free_index: torch.Tensor, *, start_pos: int):
        self.free(free_index)

    def free_segments(self, segments):
        """allocator/base.py:115-134."""
        ps = self.page_size
        prev_end = None
        for free_index, start_pos in segments:
            n = free_index.numel()
            if n == 0:
                continue
            seg_end = start_pos + n
            if prev_end is not None and start_pos // ps == (prev_end - 1) // ps:
                boundary = (start_pos // ps + 1) * ps
                free_index = free_index[boundary - start_pos:]
                start_pos = boundary
            prev_end = seg_end
            self.free_segment(free_index, start_pos=start_pos)


class TokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """page_size == 1 (allocator/token.py:27-84)."""

    def __init__(self, size: int, dtype, device, kvcache=None, need_sort: bool = False):
        super().__init__(size, 1, dtype, device, kvcache, need_sort)
        self.clear()

    def clear(self):
        # slot 0 absorbs the writes of padded tokens (token.py:41-46)
        self.free_pages = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def available_size(self):
        return len(self.free_pages) + len(self.release_pages)

    def alloc(self, need_size: int):
        if self.need_sort and need_size > len(self.free_pages):
            self.merge_and_sort_free()
        if need_size > len(self.free_pages):
            return None
        select_index = self.free_pages[:need_size]
        self.free_pages = self.free_pages[need_size:]
        return select_index

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            if self.need_sort:
                self.release_pages = torch.cat((self.release_pages, free_index))
            else:
                self.free_pages = torch.cat((self.free_pages, free_index))
        else:
            self.free_group.append(free_index)


def get_num_new_pages(seq_lens: torch.Tensor, page_size: int,
                      prefix_lens: Optional[torch.Tensor] = None, decode: bool = False) -> int:
    """srt/utils/common.py:4298-4321 (CPU tensors, so no device sync)."""
    if prefix_lens is None or decode:
        assert decode
        return int((seq_lens % page_size == 1).int().sum().item())
    after = (seq_lens + page_size - 1) // page_size
    before = (prefix_lens + page_size - 1) // page_size
    return int(torch.sum(after - before).item())


class PagedTokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """allocator/paged.py:105-345."""

    def __init__(self, size: int, page_size: int, dtype, device, kvcache=None,
                 need_sort: bool = False, debug_mode: bool = False):
        super().__init__(size, page_size, dtype, device, kvcache, need_sort)
        self.num_pages = size // page_size
        self.debug_mode = debug_mode
        self.clear()

    def clear(self):
        # page 0 absorbs the writes of padded tokens (paged.py:329-337)
        self.free_pages = torch.arange(1, self.num_pages + 1, dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []
        self.free_page_reps_group: List[torch.Tensor] = []
        self.release_pages = torch.empty((0,), dtype=torch.int64, device=self.device)

    def alloc(self, need_size: int):
        if self.debug_mode:
            assert need_size % self.page_size == 0, "The allocation size should be page-aligned"
        num_pages = need_size // self.page_size
        if self.need_sort and num_pages > len(self.free_pages):
            self.merge_and_sort_free()
        if num_pages > len(self.free_pages):
            return None
        out_pages = self.free_pages[:num_pages]
        self.free_pages = self.free_pages[num_pages:]
        return (out_pages[:, None] * self.page_size
                + torch.arange(self.page_size, device=self.device)).reshape(-1)

    def alloc_extend(self, prefix_lens, prefix_lens_cpu, seq_lens, seq_lens_cpu, last_loc,
                     extend_num_tokens: int, num_new_pages: Optional[int] = None):
        if self.debug_mode:
            assert torch.all((last_loc + 1) % self.page_size == prefix_lens % self.page_size)
        bs = len(prefix_lens)
        if self.need_sort and extend_num_tokens // self.page_size + bs + 1 > len(self.free_pages):
            self.merge_and_sort_free()
        out_indices = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
        ops.alloc_extend(prefix_lens.to(torch.int64), seq_lens.to(torch.int64),
                         last_loc.to(torch.int64), self.free_pages, out_indices, self.page_size)
        if self.debug_mode:
            assert len(torch.unique(out_indices)) == len(out_indices)
        if num_new_pages is None:
            num_new_pages = get_num_new_pages(seq_lens_cpu, self.page_size, prefix_lens_cpu)
        if num_new_pages > len(self.free_pages):
            return None
        self.free_pages = self.free_pages[num_new_pages:]
        return out_indices

    def alloc_decode(self, seq_lens, seq_lens_cpu, last_loc):
        if self.debug_mode:
            assert torch.all((last_loc + 2) % self.page_size == seq_lens % self.page_size)
        bs = len(seq_lens)
        if self.need_sort and bs > len(self.free_pages):
            self.merge_and_sort_free()
        out_indices = torch.empty((bs,), dtype=torch.int64, device=self.device)
        ops.alloc_decode(seq_lens.to(torch.int64), last_loc.to(torch.int64), self.free_pages,
                         out_indices, self.page_size)
        if self.debug_mode:
            assert len(torch.unique(out_indices)) == len(out_indices)
        num_new_pages = get_num_new_pages(seq_lens_cpu, self.page_size, decode=True)
        if num_new_pages > len(self.free_pages):
            return None
        self.free_pages = self.free_pages[num_new_pages:]
        return out_indices

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            self._release_page_ids(torch.unique(free_index // self.page_size))
        else:
            self.free_group.append(free_index)
        if self.debug_mode:
            self._debug_check_no_duplicate_pages()

    def free_segment(self, free_index: torch.Tensor, *, start_pos: int):
        """Fixed-shape free: page representatives are stride slices (paged.py:273-301)."""
        if free_index.numel() == 0:
            return
        ps = self.page_size
        offset = start_pos % ps
        if offset == 0:
            pieces = (free_index[::ps],)
        else:
            pieces = (free_index[:1], free_index[ps - offset:: ps])
        if self.is_not_in_free_group:
            self._release_page_ids(*(p // ps for p in pieces))
            if self.debug_mode:
                self._debug_check_no_duplicate_pages()
        else:
            self.free_page_reps_group.extend(pieces)

    def _debug_check_no_duplicate_pages(self):
        pages = torch.cat((self.free_pages, self.release_pages))
        assert len(torch.unique(pages)) == len(pages)

    def _release_page_ids(self, *page_ids: torch.Tensor):
        if self.need_sort:
            self.release_pages = torch.cat((*page_ids, self.release_pages))
        else:
            self.free_pages = torch.cat((*page_ids, self.free_pages))

    def free_group_begin(self):
        super().free_group_begin()
        self.free_page_reps_group = []

    def free_group_end(self):
        super().free_group_end()
        if self.free_page_reps_group:
            self._release_page_ids(torch.cat(self.free_page_reps_group) // self.page_size)
            self.free_page_reps_group = []
        if self.debug_mode:
            self._debug_check_no_duplicate_pages()
