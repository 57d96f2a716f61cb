"""KV slot / page allocators over a DEVICE-RESIDENT free list (csrc/rx_pool.hip, ``rx_pool_*``).

Observable contract = the reference's TokenToKVPoolAllocator (srt/mem_cache/allocator/token.py:27-84)
and PagedTokenToKVPoolAllocator (allocator/paged.py:105-345): same method names, same return values
(``None`` when the request cannot be served), and -- what "KV page indices bit-exact" pins -- the same free-list
ORDER after every call, so a replayed call sequence hands out identical indices.

What differs is where the list lives.  The reference keeps ``free_pages`` as a torch tensor that every call
re-creates by slicing / ``torch.cat`` and whose paged ``free`` synchronises the host through ``torch.unique``
(paged.py:261-271).  Here the list is a ring in HBM (`DeviceFreeList`) that kernels update in place: an allocator
call is one or two launches on the current stream, allocates nothing but its result and never waits for the
GPU.  The host only mirrors the list LENGTHS; they are exact except right after a data-dependent ``free`` (how
many distinct pages a slot list touches), and are then re-read lazily -- two words -- the next time a decision
needs them (``available_size``, an out-of-memory check).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Tuple

import torch

from .. import lib as _L

FREE, RELEASE = 0, 1


def _i64(t: torch.Tensor) -> torch.Tensor:
    t = t if t.dtype == torch.int64 else t.to(torch.int64)
    return t if t.is_contiguous() else t.contiguous()


class DeviceFreeList:
    """The two id lists of an allocator (free, release) as rings in device memory + their state words.
    Thin wrapper over the rx_pool_* C ABI; every method enqueues kernels on the current stream."""

    def __init__(self, num_ids: int, device, with_release: bool):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("the allocator's free list lives in GPU memory: device must be a GPU "
                               "(there is no CPU fallback)")
        self._lib = _L.load()
        self.device, self.num_ids = dev, int(num_ids)
        cap = self.capacity = self.num_ids + 1
        self.free_ring = torch.empty(cap, dtype=torch.int64, device=dev)
        self.release_ring = torch.empty(cap, dtype=torch.int64, device=dev) if with_release else None
        self.flags = torch.zeros(self.num_ids + 1, dtype=torch.uint8, device=dev)
        self.tiles = torch.zeros(self._lib.rx_pool_tile_scratch_len(self.num_ids), dtype=torch.int64, device=dev)
        self.state = torch.zeros(self._lib.rx_pool_state_words(), dtype=torch.int64, device=dev)
        d = self.desc = _L.RxPoolDesc()
        d.free_ring, d.capacity = self.free_ring.data_ptr(), cap
        d.release_ring = None if self.release_ring is None else self.release_ring.data_ptr()
        d.flags, d.num_ids = self.flags.data_ptr(), self.num_ids
        d.tile_scratch, d.state = self.tiles.data_ptr(), self.state.data_ptr()
        self._ref = C.byref(d)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _call(self, name, *args):
        st = getattr(self._lib, name)(self._ref, *args, self._stream())
        if st:
            _L.check(st, name)

    def reset(self, first_id: int, n: int):
        self._call("rx_pool_reset", first_id, n)

    def load(self, which: int, ids: torch.Tensor):
        ids = _i64(ids)
        self._call("rx_pool_load", which, C.c_void_p(ids.data_ptr()), ids.numel())

    def snapshot(self, which: int, count: int) -> torch.Tensor:
        out = torch.empty(count, dtype=torch.int64, device=self.device)
        if count:
            self._call("rx_pool_snapshot", which, C.c_void_p(out.data_ptr()), count)
        return out

    def counts(self) -> Tuple[int, int, int]:
        """(free count, release count, refused allocations, list overflows): a device -> host read of the state
        words."""
        s = self.state.tolist()
        return int(s[1]), int(s[3]), int(s[4]), int(s[7])

    def take(self, num_pages: int, page_size: int) -> torch.Tensor:
        out = torch.empty(num_pages * page_size, dtype=torch.int64, device=self.device)
        self._call("rx_pool_alloc", num_pages, page_size, C.c_void_p(out.data_ptr()))
        return out

    def append(self, which: int, ids: torch.Tensor):
        ids = _i64(ids)
        self._call("rx_pool_append", which, C.c_void_p(ids.data_ptr()), ids.numel())

    def prepend_strided(self, which: int, idx: torch.Tensor, has_first: bool, start: int, stride: int,
                        page_size: int):
        idx = _i64(idx)
        self._call("rx_pool_prepend_strided", which, C.c_void_p(idx.data_ptr()), idx.numel(), int(has_first),
                   start, stride, page_size)

    def mark(self, idx: torch.Tensor, page_size: int):
        idx = _i64(idx)
        self._call("rx_pool_mark", C.c_void_p(idx.data_ptr()), idx.numel(), page_size)

    def flush_marks(self, which: int):
        self._call("rx_pool_flush_marks", which)

    def merge_sort(self):
        self._call("rx_pool_merge_sort")


class BaseTokenToKVPoolAllocator:
    """Shared bookkeeping (allocator/base.py:27-134): the host mirror of the list lengths, free groups,
    merge_and_sort_free, free_segments."""

    def __init__(self, size: int, page_size: int, dtype, device, kvcache=None, need_sort: bool = False):
        self.size, self.page_size, self.dtype, self.device = size, page_size, dtype, device
        self._kvcache = kvcache
        self.need_sort = need_sort
        self.num_ids = size // page_size
        self._list = DeviceFreeList(self.num_ids, device, with_release=True)
        self._grouping = False
        # host mirror: list lengths; None = unknown until the next read-back
        self._n_free: Optional[int] = 0
        self._n_release: Optional[int] = 0
        self._target = RELEASE if need_sort else FREE  # where frees go (base.py:70-76 drains RELEASE)

    # ---- host mirror -------------------------------------------------------------------------------------------
    def _sync_counts(self) -> None:
        nf, nr, refused, overflow = self._list.counts()
        if overflow:
            raise RuntimeError(f"allocator: {overflow} free(s) pushed the free + release lists past the pool's "
                               f"{self.num_ids} pages (a double free, or a page freed both by free() and free_segment())")
        if refused:
            raise RuntimeError(f"allocator: {refused} allocation(s) reached the device without enough free pages "
                               f"(host mirror out of step with the device list)")
        self._n_free, self._n_release = nf, nr

    def _free_count(self) -> int:
        if self._n_free is None:
            self._sync_counts()
        return self._n_free

    def _release_count(self) -> int:
        if self._n_release is None:
            self._sync_counts()
        return self._n_release

    def _bump(self, which: int, delta: Optional[int]) -> None:
        """The list `which` grew by delta entries (None: by a data-dependent amount)."""
        attr = "_n_free" if which == FREE else "_n_release"
        cur = getattr(self, attr)
        setattr(self, attr, None if (delta is None or cur is None) else cur + delta)

    # ---- queries -------------------------------------------------------------------------------------------------
    @property
    def size_full(self):
        return self.size

    def available_size(self):
        return (self._free_count() + self._release_count()) * self.page_size

    def get_kvcache(self):
        return self._kvcache

    @property
    def free_pages(self) -> torch.Tensor:
        """The free list as a tensor, head first (a copy; synchronises -- tests / debugging)."""
        return self._list.snapshot(FREE, self._free_count())

    @free_pages.setter
    def free_pages(self, ids: torch.Tensor):
        self._list.load(FREE, ids)
        self._n_free = int(ids.numel())

    @property
    def release_pages(self) -> torch.Tensor:
        return self._list.snapshot(RELEASE, self._release_count())

    @release_pages.setter
    def release_pages(self, ids: torch.Tensor):
        self._list.load(RELEASE, ids)
        self._n_release = int(ids.numel())

    def backup_state(self):
        return self.free_pages, self.release_pages

    def restore_state(self, state):
        self.free_pages, self.release_pages = state

    def clear(self):
        # id 0 is the padding sink: never handed out (token.py:42-49, paged.py:329-337)
        self._list.reset(1, self.num_ids)
        self._n_free, self._n_release = self.num_ids, 0
        self._grouping = False
        self._group_pending: List = []

    # ---- shared operations ---------------------------------------------------------------------------------------
    def _ensure(self, pages_needed: int) -> bool:
        """True when `pages_needed` ids can leave the free list (after a sort-merge if that is what it takes)."""
        if pages_needed > self._free_count() and self.need_sort and self._release_count() > 0:
            self.merge_and_sort_free()
        return pages_needed <= self._free_count()

    def merge_and_sort_free(self):
        if self._release_count() > 0:
            self._list.merge_sort()
            self._n_free, self._n_release = self._free_count() + self._release_count(), 0

    def alloc_decode_rows(self, req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                          seq_lens: torch.Tensor, seq_lens_cpu: torch.Tensor):
        """alloc_for_decode's three steps (srt/mem_cache/allocation.py:539-593: gather every request's last slot from
        its req_to_token row, alloc_decode, scatter the new slots back into the rows) as ONE launch.  seq_lens are
        the lengths BEFORE the new token.  Returns out_cache_loc int64[bs], or None when the pages run out."""
        ps, bs = self.page_size, seq_lens.shape[0]
        if req_to_token.dtype != torch.int32 or req_to_token.stride(-1) != 1:
            raise TypeError("req_to_token must be int32 with contiguous rows")
        need = bs if ps == 1 else int((seq_lens_cpu % ps == 0).sum())
        if self.need_sort and bs > self._free_count():   # alloc_decode's sort-merge trigger (paged.py:232-236)
            self.merge_and_sort_free()
        if not self._ensure(need):
            return None
        out = torch.empty((bs,), dtype=torch.int64, device=self.device)
        rpi64, seq64 = _i64(req_pool_indices), _i64(seq_lens)
        self._list._call("rx_pool_alloc_decode_rows", C.c_void_p(req_to_token.data_ptr()), req_to_token.stride(0),
                         C.c_void_p(rpi64.data_ptr()), C.c_void_p(seq64.data_ptr()), C.c_void_p(out.data_ptr()),
                         bs, ps, need)
        self._n_free -= need
        return out

    def alloc_extend_rows(self, req_to_token: torch.Tensor, table_cpu: torch.Tensor, extend_num_tokens: int):
        """alloc_for_extend's device side (srt/mem_cache/allocation.py:303-403: last_loc gather, alloc_extend or -- page
        size 1 -- alloc_token_slots, write_cache_indices) as ONE copy + ONE launch.  table_cpu: int64 [4, bs] on the HOST
        (ideally pinned) = req_pool_idx | prefix_len | seq_len | device address of the request's cached prefix slots
        (int64; 0 = none).  Returns out_cache_loc int64[extend_num_tokens], or None when the pages run out."""
        ps, bs = self.page_size, table_cpu.shape[1]
        if req_to_token.dtype != torch.int32 or req_to_token.stride(-1) != 1:
            raise TypeError("req_to_token must be int32 with contiguous rows")
        if table_cpu.dtype != torch.int64 or table_cpu.dim() != 2 or table_cpu.shape[0] != 4 or not table_cpu.is_contiguous():
            raise TypeError("table_cpu must be a contiguous host int64 [4, bs]")
        pre, seq = table_cpu[1], table_cpu[2]
        need = int(((seq + ps - 1) // ps - (pre + ps - 1) // ps).sum())
        # the reference's sort-merge triggers, so that the list order -- hence the indices -- evolves identically
        # (paged.py:188-193 on the estimate tokens / page + bs + 1; token.py:55-58 on the need itself)
        if ps > 1 and self.need_sort and extend_num_tokens // ps + bs + 1 > self._free_count():
            self.merge_and_sort_free()
        if not self._ensure(need):
            return None
        out = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
        table = table_cpu.to(self.device, non_blocking=True)
        self._list._call("rx_pool_alloc_extend_rows", C.c_void_p(req_to_token.data_ptr()), req_to_token.stride(0),
                         C.c_void_p(table.data_ptr()), C.c_void_p(out.data_ptr()), bs, ps, need)
        self._n_free -= need
        if getattr(self, "debug_mode", False):
            assert out.unique().numel() == out.numel(), "alloc_extend_rows handed out a slot twice"
        return out

    def free_group_begin(self):
        self._grouping = True
        self._group_pending = []

    def free_group_end(self):
        self._grouping = False
        pending, self._group_pending = self._group_pending, []
        self._flush_group(pending)

    def _release(self, piece) -> None:
        """Every give-back goes through here: queued while a group is open, applied to the device list otherwise."""
        if self._grouping:
            self._group_pending.append(piece)
        else:
            self._flush_group([piece])

    def alloc_extend(self, *args, **kwargs):
        raise NotImplementedError("alloc_extend is only for paged allocator")

    def alloc_decode(self, *args, **kwargs):
        raise NotImplementedError("alloc_decode is only for paged allocator")

    def free_segment(self, free_index: torch.Tensor, *, start_pos: int):
        self.free(free_index)

    def free_segments(self, segments: Iterable[Tuple[torch.Tensor, int]]):
        """Several runs of ONE request freed together (base.py:118-134): when a run starts inside the page the
        previous run ended in, that page already went with the previous run -- skip ahead to the next boundary."""
        last_end = None
        for idx, pos in segments:
            count = idx.numel()
            if count == 0:
                continue
            end = pos + count
            same_page = last_end is not None and pos // self.page_size == (last_end - 1) // self.page_size
            if same_page:
                nxt = (pos // self.page_size + 1) * self.page_size
                idx, pos = idx[nxt - pos:], nxt
            last_end = end
            self.free_segment(idx, start_pos=pos)


class TokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """page_size == 1 (token.py:27-84): frees go to the BACK of the list, in call order."""

    def __init__(self, size: int, dtype, device, kvcache=None, need_sort: bool = False):
        super().__init__(size, 1, dtype, device, kvcache, need_sort)
        self.clear()

    def available_size(self):
        return self._free_count() + self._release_count()

    def alloc(self, need_size: int):
        if not self._ensure(need_size):
            return None
        out = self._list.take(need_size, 1)
        self._n_free -= need_size
        return out

    def free(self, free_index: torch.Tensor):
        if free_index.numel():
            self._release(free_index)

    def _flush_group(self, pending):
        for idx in pending:  # cat(free_group) appended = the pieces appended in order
            self._list.append(self._target, idx)
            self._bump(self._target, idx.numel())


def get_num_new_pages(seq_lens: torch.Tensor, page_size: int, prefix_lens: Optional[torch.Tensor] = None,
                      decode: bool = False) -> int:
    """srt/utils/common.py:4298-4321, on the CPU copies of the lens (no device sync)."""
    if decode or prefix_lens is None:
        assert decode
        return int((seq_lens % page_size == 1).sum())
    up = lambda t: (t + (page_size - 1)) // page_size  # noqa: E731
    return int((up(seq_lens) - up(prefix_lens)).sum())


class PagedTokenToKVPoolAllocator(BaseTokenToKVPoolAllocator):
    """page_size > 1 (paged.py:105-345): the list holds PAGE ids; frees go to the FRONT."""

    def __init__(self, size: int, page_size: int, dtype, device, kvcache=None, need_sort: bool = False,
                 debug_mode: bool = False):
        super().__init__(size, page_size, dtype, device, kvcache, need_sort)
        self.num_pages = self.num_ids
        self.debug_mode = debug_mode
        self.clear()

    # ---- allocation ----------------------------------------------------------------------------------------------
    def alloc(self, need_size: int):
        if self.debug_mode and need_size % self.page_size:
            raise AssertionError("The allocation size should be page-aligned")
        pages = need_size // self.page_size
        if not self._ensure(pages):
            return None
        out = self._list.take(pages, self.page_size)
        self._n_free -= pages
        return out

    def alloc_extend(self, prefix_lens, prefix_lens_cpu, seq_lens, seq_lens_cpu, last_loc,
                     extend_num_tokens: int, num_new_pages: Optional[int] = None):
        ps = self.page_size
        if self.debug_mode and not bool(torch.all((last_loc + 1) % ps == prefix_lens % ps)):
            raise AssertionError("last_loc does not continue the cached prefix")
        bs = len(prefix_lens)
        if num_new_pages is None:
            num_new_pages = get_num_new_pages(seq_lens_cpu, ps, prefix_lens_cpu)
        # the reference sort-merges on the estimate tokens / page + bs + 1 (paged.py:188-193); keep its trigger so
        # that the list order -- hence the indices -- evolves identically
        if self.need_sort and extend_num_tokens // ps + bs + 1 > self._free_count():
            self.merge_and_sort_free()
        if num_new_pages > self._free_count():
            return None  # decided on the host BEFORE any launch: the kernel never sees a short list
        out = torch.empty((extend_num_tokens,), dtype=torch.int64, device=self.device)
        pre64, seq64, last64 = _i64(prefix_lens), _i64(seq_lens), _i64(last_loc)  # alive until the launch is queued
        self._list._call("rx_pool_alloc_extend", C.c_void_p(pre64.data_ptr()), C.c_void_p(seq64.data_ptr()),
                         C.c_void_p(last64.data_ptr()), C.c_void_p(out.data_ptr()), bs, ps, num_new_pages)
        self._n_free -= num_new_pages
        if self.debug_mode:
            assert out.unique().numel() == out.numel(), "alloc_extend handed out a slot twice"
        return out

    def alloc_decode(self, seq_lens, seq_lens_cpu, last_loc):
        ps = self.page_size
        if self.debug_mode and not bool(torch.all((last_loc + 2) % ps == seq_lens % ps)):
            raise AssertionError("last_loc does not precede the new token")
        bs = len(seq_lens)
        if self.need_sort and bs > self._free_count():
            self.merge_and_sort_free()
        num_new_pages = get_num_new_pages(seq_lens_cpu, ps, decode=True)
        if num_new_pages > self._free_count():
            return None
        out = torch.empty((bs,), dtype=torch.int64, device=self.device)
        seq64, last64 = _i64(seq_lens), _i64(last_loc)
        self._list._call("rx_pool_alloc_decode", C.c_void_p(seq64.data_ptr()), C.c_void_p(last64.data_ptr()),
                         C.c_void_p(out.data_ptr()), bs, ps, num_new_pages)
        self._n_free -= num_new_pages
        if self.debug_mode:
            assert out.unique().numel() == out.numel(), "alloc_decode handed out a slot twice"
        return out

    # ---- release -------------------------------------------------------------------------------------------------
    def free(self, free_index: torch.Tensor):
        """Slots of arbitrary pages: the SORTED SET of their pages goes to the front (paged.py:261-271).  The set
        is formed on the device (flag per page + ordered compaction); its size is not known here, so the mirror
        of that list's length becomes unknown until somebody needs it."""
        if free_index.numel():
            self._release(("slots", free_index))

    def free_segment(self, free_index: torch.Tensor, *, start_pos: int):
        """A run of consecutive positions starting at start_pos (paged.py:273-301): one representative slot per page
        at fixed strides -- the first slot, then every page_size-th from the next page boundary -- so the page
        count is known on the host and the pages go to the front in run order."""
        if free_index.numel():
            off = start_pos % self.page_size
            self._release(("segment", free_index, bool(off), (self.page_size - off) if off else 0))

    def _prepend_segment(self, idx: torch.Tensor, has_first: bool, start: int):
        ps, n = self.page_size, idx.numel()
        reps = (1 if has_first else 0) + (-(-(n - start) // ps) if start < n else 0)
        self._list.prepend_strided(self._target, idx, has_first, start, ps, ps)
        self._bump(self._target, reps)

    def _flush_group(self, pending):
        """free_group_end (paged.py:318-327): first the union of all grouped `free` calls as one sorted set, then
        the grouped segments' representatives in FRONT of that, in the order they were queued."""
        slots = [p[1] for p in pending if p[0] == "slots"]
        if slots:
            for idx in slots:
                self._list.mark(idx, self.page_size)
            self._list.flush_marks(self._target)
            self._bump(self._target, None)
        # front inserts compose right to left: the last queued segment goes in first
        for p in reversed([p for p in pending if p[0] == "segment"]):
            self._prepend_segment(*p[1:])
        if self.debug_mode:
            self._debug_check_no_duplicate_pages()

    def _debug_check_no_duplicate_pages(self):
        """debug_mode only (synchronises): after sorting, no two neighbours of free + release may be equal."""
        ids, _ = torch.sort(torch.cat((self._list.snapshot(FREE, self._free_count()),
                                       self._list.snapshot(RELEASE, self._release_count()))))
        if ids.numel() > 1 and bool((ids[1:] == ids[:-1]).any()):
            raise AssertionError("a page sits in the free lists twice")
