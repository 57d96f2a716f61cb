"""CPU oracle for the RadixAttention hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``sglang_amd``) never does: it fails
loudly when the HIP library is missing.

Every function restates, in numpy, the algorithm of one reference function
(path relative to /root/reference/python/sglang/).  Parity status: PINNED --
the golden fixtures under ``tests/golden`` were produced by running the
reference's own Triton kernels (TRITON_INTERPRET=1), paged allocators and
compiled C++ CPU kernels in the build container (``tests/golden/make_golden.py``),
and ``tests/test_oracle_golden.py`` checks this file against every one of them.
One section is pinned differently and says so in its own comment: the quick all-reduce (C3), whose reference kernel is
HIP-only -- it is held to the reference's own test properties and, for ``v_rcp_f16``, to a table read back from an MI355X.

bf16/fp16 tensors are passed as numpy ``uint16`` bit patterns (bf16) or
``float16``; helpers below convert.  All attention math is done in float64 so
the oracle is the "exact" answer the tolerance is measured from.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# dtype helpers
# --------------------------------------------------------------------------


def bf16_to_f32(x_u16: np.ndarray) -> np.ndarray:
    """bf16 bit patterns (uint16) -> float32."""
    x = np.ascontiguousarray(x_u16).astype(np.uint32) << 16
    return x.view(np.float32)


def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round-to-nearest-even (NaN kept quiet)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    nan = np.isnan(x)
    out = rounded.astype(np.uint16)
    out[nan] = 0x7FC0
    return out


# ---- OCP fp8 e4m3fn (torch.float8_e4m3fn; gfx950's native fp8) ----------------------------------
# The reference's fp8 KV pools are plain casts: ``cache_k.to(torch.float8_e4m3fn)``
# (srt/mem_cache/memory_pool.py:2334-2343 for MHA, :4022-4066 for MLA latent rows), i.e. torch's
# round-to-nearest-even with NaN above the largest finite value 448.  Pinned against torch's own
# CPU cast in tests/test_oracle_golden.py.
def fp8_e4m3fn_decode(u8: np.ndarray) -> np.ndarray:
    u = np.asarray(u8, dtype=np.uint8).astype(np.int32)
    sign = np.where(u & 0x80, -1.0, 1.0)
    e = (u >> 3) & 0xF
    m = u & 0x7
    val = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * np.exp2(e.astype(np.float64) - 7))
    val = np.where((e == 15) & (m == 7), np.nan, val)
    return (sign * val).astype(np.float32)


_FP8_POS = fp8_e4m3fn_decode(np.arange(0, 0x7F, dtype=np.uint8)).astype(np.float64)  # 0 .. 448, ascending


def fp8_e4m3fn_encode(x: np.ndarray) -> np.ndarray:
    """float -> e4m3fn byte, round to nearest, ties to the even mantissa; |x| beyond the rounding
    range of 448 and NaN give 0x7F | sign (torch); the gfx950 instruction saturates to 448 instead."""
    x = np.asarray(x, dtype=np.float64)
    a = np.abs(x)
    hi = np.searchsorted(_FP8_POS, a, side="left").clip(1, len(_FP8_POS) - 1)
    lo = hi - 1
    dlo, dhi = a - _FP8_POS[lo], _FP8_POS[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    code = np.where(a > 464.0, 0x7F, code)  # 464 = midpoint of 448 and the (absent) 480: the tie goes to 448
    code = np.where(np.isnan(x), 0x7F, code)
    return (code | np.where(np.signbit(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def quantize_kv_fp8(x: np.ndarray, scale: float, src_is_bf16: bool) -> np.ndarray:
    """set_kv_buffer's fp8 write (memory_pool.py:2334-2343): in-place ``x.div_(scale)`` on the 16-bit
    tensor (fp32 quotient rounded back to the source dtype), then ``.to(float8_e4m3fn)``."""
    x = np.asarray(x, dtype=np.float32)
    if scale != 1.0:
        t = x / np.float32(scale)
        x = bf16_to_f32(f32_to_bf16(t)) if src_is_bf16 else t.astype(np.float16).astype(np.float32)
    return fp8_e4m3fn_encode(x)



def quantize_fused_fp8(x: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """fused_fp8_qkv_kv_cache's per-element arithmetic (kernels/jit/csrc/attention/fused_fp8_qkv_kv_cache.cuh:33-55,64-66):
    ``static_cast<fp8_e4m3_t>(float(x) * inv_scale)`` with ``inv_scale = 1.0f / scale`` in fp32 -- CUDA's satfinite,
    round-to-nearest-even cast; the reference test states it as ``(x.float() * inv).clamp(-448, 448).to(float8_e4m3fn)``
    (test/registered/kernels/ops/kvcache/test_fused_fp8_qkv_kv_cache.py:13-16).  x: float32 values of the 16-bit source."""
    x = np.asarray(x, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(scale)
    y = np.clip(x * inv, np.float32(-448.0), np.float32(448.0))  # (np.clip keeps NaN)
    return fp8_e4m3fn_encode(y)


def to_f64(x: np.ndarray) -> np.ndarray:
    """Accept bf16-bits (uint16), float16, float32 -> float64."""
    if x.dtype == np.uint16:
        return bf16_to_f32(x).astype(np.float64)
    return x.astype(np.float64)


# --------------------------------------------------------------------------
# a4  ReqToTokenPool            srt/mem_cache/memory_pool.py:256-326
# --------------------------------------------------------------------------


class ReqToTokenPoolOracle:
    """req_to_token int32[size+1, max_context_len]; row 0 is the padding sink
    (memory_pool.py:273-281); free-list of slots 1..size, FIFO alloc
    (memory_pool.py:306-307), append on free (:318)."""

    def __init__(self, size: int, max_context_len: int):
        self.size = size
        self.max_context_len = max_context_len
        self.req_to_token = np.zeros((size + 1, max_context_len), dtype=np.int32)
        self.free_slots = list(range(1, size + 1))

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        sel = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return sel

    def free(self, idx: int) -> None:
        self.free_slots.append(idx)

    def write(self, indices, values) -> None:
        self.req_to_token[indices] = values

    def clear(self) -> None:
        self.free_slots = list(range(1, self.size + 1))


# --------------------------------------------------------------------------
# a5  allocators     srt/mem_cache/allocator/{base,token,paged}.py
#                    kernels/ops/memory/allocator.py:16-135
# --------------------------------------------------------------------------


class TokenAllocatorOracle:
    """TokenToKVPoolAllocator (allocator/token.py:27-84): page_size == 1,
    free_pages = arange(1, size+1) (:41-46), alloc = head slice (:53-62),
    free = append (or to release_pages when need_sort) (:64-74)."""

    def __init__(self, size: int, need_sort: bool = False):
        self.size = size
        self.page_size = 1
        self.need_sort = need_sort
        self.clear()

    def clear(self):
        self.free_pages = np.arange(1, self.size + 1, dtype=np.int64)
        self.release_pages = np.empty((0,), dtype=np.int64)
        self.is_not_in_free_group = True
        self.free_group: List[np.ndarray] = []

    def available_size(self):
        return len(self.free_pages) + len(self.release_pages)

    def merge_and_sort_free(self):  # allocator/base.py:70-76
        if len(self.release_pages) > 0:
            self.free_pages = np.sort(
                np.concatenate((self.free_pages, self.release_pages)), kind="stable"
            )
            self.release_pages = np.empty((0,), dtype=np.int64)

    def alloc(self, need_size: int):
        if self.need_sort and need_size > len(self.free_pages):
            self.merge_and_sort_free()
        if need_size > len(self.free_pages):
            return None
        sel = self.free_pages[:need_size].copy()
        self.free_pages = self.free_pages[need_size:]
        return sel

    def free(self, free_index: np.ndarray):
        free_index = np.asarray(free_index, dtype=np.int64)
        if free_index.size == 0:
            return
        if self.is_not_in_free_group:
            if self.need_sort:
                self.release_pages = np.concatenate((self.release_pages, free_index))
            else:
                self.free_pages = np.concatenate((self.free_pages, free_index))
        else:
            self.free_group.append(free_index)

    def free_group_begin(self):  # base.py:60-62
        self.is_not_in_free_group = False
        self.free_group = []

    def free_group_end(self):  # base.py:64-67
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(np.concatenate(self.free_group))


def alloc_extend_ref(prefix_lens, seq_lens, last_loc, free_pages, page_size):
    """alloc_extend_kernel (kernels/ops/memory/allocator.py:16-100) ==
    alloc_extend_naive (allocator/paged.py:45-102).  Returns out_indices int64
    of length sum(seq_lens - prefix_lens).  Three parts per request: fill the
    old partial page after last_loc, whole new pages taken in order from
    free_pages, then the new partial page."""
    prefix_lens = np.asarray(prefix_lens, dtype=np.int64)
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    last_loc = np.asarray(last_loc, dtype=np.int64)
    free_pages = np.asarray(free_pages, dtype=np.int64)
    ps = page_size
    extend_lens = seq_lens - prefix_lens
    out = np.empty((int(extend_lens.sum()),), dtype=np.int64)
    pages_after = (seq_lens + ps - 1) // ps
    pages_before = (prefix_lens + ps - 1) // ps
    new_pages = pages_after - pages_before
    out_start = np.cumsum(extend_lens) - extend_lens
    page_start = np.cumsum(new_pages) - new_pages
    for i in range(len(seq_lens)):
        pre, seq = int(prefix_lens[i]), int(seq_lens[i])
        o = int(out_start[i])
        n1 = min(seq, (pre + ps - 1) // ps * ps) - pre
        out[o : o + n1] = last_loc[i] + 1 + np.arange(n1)
        if pre + n1 == seq:
            continue
        n2 = seq // ps * ps - (pre + ps - 1) // ps * ps
        off = np.arange(n2)
        out[o + n1 : o + n1 + n2] = (
            free_pages[int(page_start[i]) + off // ps] * ps + off % ps
        )
        if pre + n1 + n2 == seq:
            continue
        n3 = seq - seq // ps * ps
        start = free_pages[int(page_start[i]) + int(new_pages[i]) - 1]
        out[o + n1 + n2 : o + n1 + n2 + n3] = start * ps + np.arange(n3)
    return out


def alloc_decode_ref(seq_lens, last_loc, free_pages, page_size):
    """alloc_decode_kernel (kernels/ops/memory/allocator.py:103-135): a request
    whose new token opens a page ((seq_len-1) % page == 0) takes the next free
    page, otherwise last_loc + 1."""
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    last_loc = np.asarray(last_loc, dtype=np.int64)
    free_pages = np.asarray(free_pages, dtype=np.int64)
    ps = page_size
    pre = seq_lens - 1
    new_pages = (seq_lens + ps - 1) // ps - (pre + ps - 1) // ps
    page_start = np.cumsum(new_pages) - new_pages
    out = np.empty((len(seq_lens),), dtype=np.int64)
    for i in range(len(seq_lens)):
        if new_pages[i] == 0:
            out[i] = last_loc[i] + 1
        else:
            out[i] = free_pages[int(page_start[i])] * ps
    return out


class PagedAllocatorOracle:
    """PagedTokenToKVPoolAllocator (allocator/paged.py:105-345)."""

    def __init__(self, size: int, page_size: int, need_sort: bool = False):
        self.size = size
        self.page_size = page_size
        self.num_pages = size // page_size
        self.need_sort = need_sort
        self.clear()

    def clear(self):  # paged.py:329-337: page 0 reserved for padded writes
        self.free_pages = np.arange(1, self.num_pages + 1, dtype=np.int64)
        self.release_pages = np.empty((0,), dtype=np.int64)
        self.is_not_in_free_group = True
        self.free_group: List[np.ndarray] = []
        self.free_page_reps_group: List[np.ndarray] = []

    def available_size(self):  # base.py:54-55
        return (len(self.free_pages) + len(self.release_pages)) * self.page_size

    def merge_and_sort_free(self):
        if len(self.release_pages) > 0:
            self.free_pages = np.sort(
                np.concatenate((self.free_pages, self.release_pages)), kind="stable"
            )
            self.release_pages = np.empty((0,), dtype=np.int64)

    def alloc(self, need_size: int):  # paged.py:149-170
        num_pages = need_size // self.page_size
        if self.need_sort and num_pages > len(self.free_pages):
            self.merge_and_sort_free()
        if num_pages > len(self.free_pages):
            return None
        pages = self.free_pages[:num_pages]
        self.free_pages = self.free_pages[num_pages:]
        return (pages[:, None] * self.page_size + np.arange(self.page_size)).reshape(-1)

    def alloc_extend(self, prefix_lens, seq_lens, last_loc):  # paged.py:172-220
        prefix_lens = np.asarray(prefix_lens, dtype=np.int64)
        seq_lens = np.asarray(seq_lens, dtype=np.int64)
        ps = self.page_size
        extend_num_tokens = int((seq_lens - prefix_lens).sum())
        bs = len(prefix_lens)
        if self.need_sort and extend_num_tokens // ps + bs + 1 > len(self.free_pages):
            self.merge_and_sort_free()
        num_new = int(((seq_lens + ps - 1) // ps - (prefix_lens + ps - 1) // ps).sum())
        if num_new > len(self.free_pages):
            return None
        out = alloc_extend_ref(prefix_lens, seq_lens, last_loc, self.free_pages, ps)
        self.free_pages = self.free_pages[num_new:]
        return out

    def alloc_decode(self, seq_lens, last_loc):  # paged.py:222-259
        seq_lens = np.asarray(seq_lens, dtype=np.int64)
        ps = self.page_size
        bs = len(seq_lens)
        if self.need_sort and bs > len(self.free_pages):
            self.merge_and_sort_free()
        # get_num_new_pages(decode=True), srt/utils/common.py:4311-4314
        num_new = int((seq_lens % ps == 1).sum())
        if num_new > len(self.free_pages):
            return None
        out = alloc_decode_ref(seq_lens, last_loc, self.free_pages, ps)
        self.free_pages = self.free_pages[num_new:]
        return out

    def _release_page_ids(self, *page_ids):  # paged.py:308-312: LIFO prepend
        if self.need_sort:
            self.release_pages = np.concatenate((*page_ids, self.release_pages))
        else:
            self.free_pages = np.concatenate((*page_ids, self.free_pages))

    def free(self, free_index):  # paged.py:261-271 (torch.unique == sorted unique)
        free_index = np.asarray(free_index, dtype=np.int64)
        if free_index.size == 0:
            return
        if self.is_not_in_free_group:
            self._release_page_ids(np.unique(free_index // self.page_size))
        else:
            self.free_group.append(free_index)

    def free_segment(self, free_index, start_pos: int):  # paged.py:273-301
        free_index = np.asarray(free_index, dtype=np.int64)
        if free_index.size == 0:
            return
        ps = self.page_size
        offset = start_pos % ps
        if offset == 0:
            pieces = (free_index[::ps],)
        else:
            pieces = (free_index[:1], free_index[ps - offset :: ps])
        if self.is_not_in_free_group:
            self._release_page_ids(*(p // ps for p in pieces))
        else:
            self.free_page_reps_group.extend(pieces)

    def free_group_begin(self):  # paged.py:314-316
        self.is_not_in_free_group = False
        self.free_group = []
        self.free_page_reps_group = []

    def free_group_end(self):  # paged.py:318-327
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(np.concatenate(self.free_group))
        if self.free_page_reps_group:
            self._release_page_ids(
                np.concatenate(self.free_page_reps_group) // self.page_size
            )
            self.free_page_reps_group = []


# --------------------------------------------------------------------------
# a10 kv-index builder    kernels/ops/kvcache/kv_indices.py:8-46
#                         + cumsum of triton_backend.py:386-404
# --------------------------------------------------------------------------


def build_kv_indices(req_to_token, req_pool_indices, lens, kv_start=None,
                     out_dtype=np.int64):
    """kv_indptr[0]=0, kv_indptr[1:]=cumsum(lens) (int32, triton_backend.py:393-394);
    kv_indices[kv_indptr[i]:kv_indptr[i+1]] = req_to_token[req_pool_indices[i],
    kv_start[i] : kv_start[i]+lens[i]] (kv_indices.py:22-46)."""
    req_to_token = np.asarray(req_to_token)
    lens = np.asarray(lens, dtype=np.int64)
    bs = len(lens)
    kv_indptr = np.zeros((bs + 1,), dtype=np.int32)
    kv_indptr[1:] = np.cumsum(lens)
    kv_indices = np.empty((int(kv_indptr[-1]),), dtype=out_dtype)
    for i in range(bs):
        s = 0 if kv_start is None else int(kv_start[i])
        kv_indices[kv_indptr[i] : kv_indptr[i + 1]] = req_to_token[
            int(req_pool_indices[i]), s : s + int(lens[i])
        ]
    return kv_indptr, kv_indices


def build_unified_kv_indices(prefix_kv_indptr, prefix_kv_indices, extend_start_loc, extend_seq_lens, extend_kv_indices, bs):
    """build_unified_kv_indices (kernels/ops/attention/extend_attention.py:193-238; _copy_unified_indices_kernel
    :135-190): per request the prefix slots, then the new tokens' slots.  Returns (unified_kv_indptr int32[bs + 1],
    unified_kv_indices int64[indptr[bs]], prefix_lens int32[bs]) -- the reference's buffer is len(prefix) + len(extend)
    long and undefined past indptr[bs]."""
    pre = np.asarray(prefix_kv_indptr, dtype=np.int64)
    prefix_lens = (pre[1: bs + 1] - pre[:bs]).astype(np.int32)
    ext = np.asarray(extend_seq_lens, dtype=np.int64)[:bs]
    indptr = np.concatenate([[0], np.cumsum(prefix_lens.astype(np.int64) + ext)]).astype(np.int32)
    out = np.zeros(int(indptr[-1]), dtype=np.int64)
    for i in range(bs):
        u0, pl, e0 = int(indptr[i]), int(prefix_lens[i]), int(extend_start_loc[i])
        out[u0: u0 + pl] = np.asarray(prefix_kv_indices[int(pre[i]): int(pre[i]) + pl], dtype=np.int64)
        out[u0 + pl: u0 + pl + int(ext[i])] = np.asarray(extend_kv_indices[e0: e0 + int(ext[i])], dtype=np.int64)
    return indptr, out, prefix_lens


def draft_decode_kv_indices(req_to_token, req_pool_indices, seq_lens, positions, topk, num_steps, page_size,
                            kv_indices_width, kv_indptr_width, fill=-1):
    """generate_draft_decode_kv_indices (kernels/ops/speculative/cache_locs.py:56-141), the per-step page tables of EAGLE's
    multi-step draft decode (TritonMultiStepDraftBackend.common_template, triton_backend.py:1929-1945).  For step i
    (iters = i + 1), request b and branch k (of topk) the branch's keys are the request's seq_len cached tokens plus the
    iters draft tokens this branch has written so far:
        offset(i, b, k) = sum(seq_lens[:b]) * topk + b * iters * topk + k * (seq_len_b + iters)      (:89)
        kv_indices[i, offset : offset + seq_len_b]       = req_to_token[row_b, : seq_len_b]             (:94-100)
        kv_indices[i, offset + seq_len_b + j], j < iters = req_to_token[row_b, start + j]               (:102-123)
    with start = seq_len_b + k * num_steps when page_size == 1 or topk == 1, else -- every branch on pages of its own --
    seq_len_b // page * page + k * ceil((seq_len_b % page + num_steps) / page) * page + seq_len_b % page.
        kv_indptr[i, z] = sum(positions[:z]) + z * iters for z = b * topk + k, z = 0 standing for num_seqs * topk (:125-134;
    kv_indptr[i, 0] is never written).  Returns (kv_indices int64 [num_steps, kv_indices_width] pre-filled with `fill`,
    kv_indptr int32 [num_steps, kv_indptr_width] pre-filled with 0)."""
    req_to_token = np.asarray(req_to_token)
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    positions = np.asarray(positions, dtype=np.int64)
    num_seqs = len(seq_lens)
    kv_indices = np.full((num_steps, kv_indices_width), fill, dtype=np.int64)
    kv_indptr = np.zeros((num_steps, kv_indptr_width), dtype=np.int32)
    cum = np.concatenate([[0], np.cumsum(seq_lens)])
    for i in range(num_steps):
        iters = i + 1
        for b in range(num_seqs):
            row = req_to_token[int(req_pool_indices[b])]
            n = int(seq_lens[b])
            for k in range(topk):
                off = int(cum[b]) * topk + b * iters * topk + k * (n + iters)
                kv_indices[i, off: off + n] = row[:n]
                if page_size == 1 or topk == 1:
                    start = n + k * num_steps
                else:
                    last = n % page_size
                    new_pages = (last + num_steps + page_size - 1) // page_size
                    start = n // page_size * page_size + k * new_pages * page_size + last
                kv_indices[i, off + n: off + n + iters] = row[start: start + iters]
                z = b * topk + k
                if z == 0:
                    z = num_seqs * topk
                kv_indptr[i, z] = int(positions[:z].sum()) + z * iters
    return kv_indices, kv_indptr


# --------------------------------------------------------------------------
# a11 kv-split scheduler   kernels/ops/attention/metadata.py:11-60
# --------------------------------------------------------------------------


def _cdiv(a, b):
    return -(-a // b)


def num_kv_splits(seq_lens, num_group, num_head, num_kv_head, max_kv_splits,
                  device_core_count):
    """get_num_kv_splits_triton (metadata.py:11-60).  Output int32[num_seq*num_group],
    entry (i*num_group + g) = splits of sequence i.  float32 log2 as in Triton."""
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    num_seq = len(seq_lens)
    max_seq_len = int(seq_lens.max())
    min_seq_len = int(seq_lens.min())
    if max_seq_len * 8 < min_seq_len * 10:
        min_seq_len = max_seq_len
    max_kv_splits_1 = min(_cdiv(max_seq_len, min_seq_len), max_kv_splits)
    kv_chunk_size_1 = _cdiv(max_seq_len, max_kv_splits_1)
    ext_seq_len = np.float32(max_seq_len) / np.float32(64.0)
    ext_device_core_count = int(
        np.float32(device_core_count)
        * np.maximum(np.log2(ext_seq_len, dtype=np.float32), np.float32(1.0))
    )
    block_h, num_kv_group = 16, num_head // num_kv_head
    if num_kv_group == 1:
        token_grid = num_seq * num_group * num_head
    else:
        block_h = min(block_h, num_kv_group)
        token_grid = num_seq * num_group * _cdiv(num_head, block_h)
    max_kv_splits_2 = min(_cdiv(ext_device_core_count, token_grid), max_kv_splits)
    kv_chunk_size_2 = _cdiv(max_seq_len, max_kv_splits_2)
    splits = np.maximum(_cdiv(seq_lens, kv_chunk_size_1), _cdiv(seq_lens, kv_chunk_size_2))
    return np.repeat(splits.astype(np.int32), num_group)


# --------------------------------------------------------------------------
# a8  KV store       kernels/jit/csrc/elementwise/kvcache.cuh:189-219
#                    wrapper kernels/ops/kvcache/kvcache.py:57-110
# --------------------------------------------------------------------------


def store_kv(k, v, k_cache, v_cache, loc, size_limit=0, reserved_skip_index=0):
    """k_cache[loc[i]] = k[i]; v_cache[loc[i]] = v[i] (row copy, in place);
    rows whose loc == reserved_skip_index (default slot 0) are skipped
    (kvcache.cuh:215-217); loc outside [0, size_limit) is an error
    (device assert kvcache.cuh:209)."""
    loc = np.asarray(loc).astype(np.int64)
    if size_limit <= 0:
        size_limit = k_cache.shape[0]
    if loc.size and (loc.min() < 0 or loc.max() >= size_limit):
        raise IndexError("store_kv: loc out of [0, size_limit)")
    kc = k_cache.reshape(k_cache.shape[0], -1)
    vc = v_cache.reshape(v_cache.shape[0], -1)
    k2 = k.reshape(k.shape[0], -1)
    v2 = v.reshape(v.shape[0], -1)
    for i, idx in enumerate(loc):
        if idx == reserved_skip_index:
            continue
        kc[idx] = k2[i]
        vc[idx] = v2[i]


def move_kv(buffers: Sequence[np.ndarray], tgt_loc, src_loc):
    """move_kv_cache (memory_pool.py:2775-2842 -> cache_move.py:60-133): for every
    K and V buffer of every layer, buf[tgt_loc[i]] = buf[src_loc[i]].  Reads are
    taken before writes per row (the kernel copies row by row; tgt/src are
    disjoint in the reference's callers)."""
    tgt = np.asarray(tgt_loc).astype(np.int64)
    src = np.asarray(src_loc).astype(np.int64)
    for buf in buffers:
        rows = buf.reshape(buf.shape[0], -1)
        rows[tgt] = rows[src].copy()


# --------------------------------------------------------------------------
# a12 decode attention   kernels/ops/attention/decode_attention.py:383-805, 968-1044
# --------------------------------------------------------------------------


def _tanh_cap(x, cap):
    return cap * np.tanh(x / cap) if cap > 0 else x


def _rel_bias(bias_row, q_pos, kv_pos):
    """relative_bias_score_mod (kernels/ops/attention/score_mod.py:44-56): bias_row[q_pos - kv_pos] where
    0 <= q_pos - kv_pos < len(bias_row), else 0.  bias_row float64 [extent]; kv_pos an integer array."""
    rel = q_pos - np.asarray(kv_pos, dtype=np.int64)
    ok = (rel >= 0) & (rel < bias_row.shape[0])
    return np.where(ok, bias_row[np.clip(rel, 0, bias_row.shape[0] - 1)], 0.0)


def _gather_kv(buf, slots, kv_head):
    """buf [slots, Hkv, D] (NHD, memory_pool.py:2030-2041) -> [n, D] float64."""
    return to_f64(buf[slots, kv_head])


def decode_attention(q, k_buffer, v_buffer, kv_indptr, kv_indices, sm_scale,
                     k_scale=1.0, v_scale=1.0, logit_cap=0.0, sinks=None,
                     return_lse=False, xai_temperature_len=-1, score_bias=None, return_absw=False):
    """return_absw: also return A = sum_j p_j |v_j| (same softmax, |V| in place of V: parity_util.check_out's `absw`)
    as the LAST element of the result -- one pass instead of a second call on abs_values(v).
    score_bias [bs, Hq, extent] (already decoded to float): the reference's score_mod = relative_bias_score_mod with
    aux_tensors = [score_bias] (decode_attention.py:215-227,539-551: q_pos = seq_len - 1, kv_pos = list position,
    q_idx = request), added after scale, cap and temperature.
    Semantics of decode_attention_fwd (decode_attention.py:968-1044):
    o[b,h] = softmax(q[b,h]·K[idx]^T * sm_scale*k_scale [tanh cap]) · V[idx] * v_scale,
    idx = kv_indices[kv_indptr[b]:kv_indptr[b+1]], kv head = h // (Hq/Hkv).
    Optional per-head sink logit joins the denominator (stage2 :796-798).
    Returns o float64 [bs,Hq,Dv] (+ natural-log lse [bs,Hq])."""
    bs, hq, _ = q.shape
    hkv = k_buffer.shape[-2]
    dv = v_buffer.shape[-1]
    group = hq // hkv
    qf = to_f64(q)
    o = np.zeros((bs, hq, dv), dtype=np.float64)
    oa = np.zeros((bs, hq, dv), dtype=np.float64) if return_absw else None
    lse = np.full((bs, hq), -np.inf, dtype=np.float64)
    for b in range(bs):
        idx = np.asarray(kv_indices[kv_indptr[b] : kv_indptr[b + 1]]).astype(np.int64)
        if idx.size == 0:
            continue
        xai = 1.0  # Grok temperature (decode_attention.py:156-160,212-213): the query sits at seq_len - 1
        if xai_temperature_len > 0 and idx.size - 1 > xai_temperature_len:
            xai = math.log2(idx.size - 1) / math.log2(float(xai_temperature_len))
        for kvh in range(hkv):
            kk = _gather_kv(k_buffer, idx, kvh)
            vv = _gather_kv(v_buffer, idx, kvh)
            for h in range(kvh * group, (kvh + 1) * group):
                s = _tanh_cap(kk @ qf[b, h] * (sm_scale * k_scale), logit_cap) * xai
                if score_bias is not None:
                    s = s + _rel_bias(np.asarray(score_bias[b, h], dtype=np.float64), idx.size - 1, np.arange(idx.size))
                m = s.max()
                p = np.exp(s - m)
                den = p.sum()
                if sinks is not None:
                    den = den + math.exp(float(sinks[h]) - m)
                o[b, h] = (p @ vv) / den * v_scale
                if return_absw:
                    oa[b, h] = (p @ np.abs(vv)) / den * v_scale
                lse[b, h] = m + math.log(p.sum())
    res = (o, lse) if return_lse else o
    if return_absw:
        return (res + (oa,)) if return_lse else (o, oa)
    return res


_MIN_BLOCK_KV = 32  # decode_attention.py:36


def decode_attention_split(q, k_buffer, v_buffer, kv_indptr, kv_indices,
                           num_kv_splits_arr, max_kv_splits, sm_scale,
                           k_scale=1.0, v_scale=1.0, logit_cap=0.0):
    """Restates the two-stage split-KV layout: stage 1 (decode_attention.py:
    466-594) writes, for split s of request b covering
    [s*L, min((s+1)*L, seq)) with L = cdiv(cdiv(seq, splits), 32)*32,
    attn_logits[b,h,s,:] = acc/e_sum and attn_lse[b,h,s] = e_max + log(e_sum);
    stage 2 (:731-805) merges.  Returns (attn_logits, attn_lse, o); untouched
    split entries are NaN so tests can check which entries must be written."""
    bs, hq, _ = q.shape
    hkv = k_buffer.shape[-2]
    dv = v_buffer.shape[-1]
    group = hq // hkv
    qf = to_f64(q)
    logits = np.full((bs, hq, max_kv_splits, dv), np.nan, dtype=np.float64)
    lse = np.full((bs, hq, max_kv_splits), np.nan, dtype=np.float64)
    o = np.zeros((bs, hq, dv), dtype=np.float64)
    for b in range(bs):
        idx = np.asarray(kv_indices[kv_indptr[b] : kv_indptr[b + 1]]).astype(np.int64)
        seq = idx.size
        splits = int(num_kv_splits_arr[b])
        per = _cdiv(_cdiv(seq, splits), _MIN_BLOCK_KV) * _MIN_BLOCK_KV if seq else 0
        for h in range(hq):
            kvh = h // group
            e_max, e_sum, acc = -np.inf, 0.0, np.zeros(dv)
            for s in range(max_kv_splits):
                lo, hi = per * s, min(per * (s + 1), seq)
                if hi <= lo:
                    continue
                kk = _gather_kv(k_buffer, idx[lo:hi], kvh)
                vv = _gather_kv(v_buffer, idx[lo:hi], kvh)
                sc = _tanh_cap(kk @ qf[b, h] * (sm_scale * k_scale), logit_cap)
                m = sc.max()
                p = np.exp(sc - m)
                logits[b, h, s] = (p @ vv) / p.sum()
                lse[b, h, s] = m + math.log(p.sum())
                n_max = max(lse[b, h, s], e_max)
                old = math.exp(e_max - n_max)
                w = math.exp(lse[b, h, s] - n_max)
                acc = acc * old + w * logits[b, h, s]
                e_sum = e_sum * old + w
                e_max = n_max
            if e_sum > 0:
                o[b, h] = acc / e_sum * v_scale
    return logits, lse, o


# --------------------------------------------------------------------------
# a13 extend attention   kernels/ops/attention/extend_attention.py:241-661, 664-812
# --------------------------------------------------------------------------


def extend_attention(q_extend, k_extend, v_extend, k_buffer, v_buffer, qo_indptr,
                     kv_indptr, kv_indices, is_causal=True, sm_scale=None,
                     k_scale=1.0, v_scale=1.0, logit_cap=0.0,
                     sliding_window_size=-1, sinks=None, skip_prefix=False,
                     skip_extend=False, return_lse=False, custom_mask=None, mask_indptr=None,
                     skip_prefix_custom_mask=True, window_kv_offsets=None, xai_temperature_len=-1, score_bias=None,
                     return_absw=False):
    """return_absw: also return sum_j p_j |v_j| (see decode_attention) as the last element of the result.
    score_bias [T, Hq, extent] (float): score_mod = relative_bias_score_mod, aux_tensors = [score_bias]
    (extend_attention.py:463-476 prefix stage: q_pos = P + m, kv_pos = list position; :594-607 extend stage:
    kv_pos = P + n; q_idx = global query token), added after scale, cap and temperature, before the masks.
    Semantics of extend_attention_fwd (extend_attention.py:664-812).  Request i
    has prefix tokens kv_indices[kv_indptr[i]:kv_indptr[i+1]] read from the cache
    (stage 1, :372-510; scaled by k_scale / v_scale) and E_i = qo_indptr[i+1]-
    qo_indptr[i] new tokens whose K/V are the contiguous k_extend/v_extend rows
    (stage 2, :512-631).  Query m (0-based inside the extend part) sees every
    prefix token and extend tokens n <= m when causal, all E_i otherwise.
    sliding window W>0: q_abs <= kv_abs + W (:385-390, :556-561).
    custom_mask (speculative tree attention, :320-326, :378-390, :525-539): request i owns the flat
    bytes mask[mask_indptr[i]:], a row-major [E_i, woff_i + P_i + E_i] matrix (woff = window_kv_offsets,
    0 without SWA); in the extend part it REPLACES the causal mask, in the prefix part it applies
    unless skip_prefix_custom_mask.  xai_temperature_len L>0 (:336-343, :460, :591): scores of the
    query at absolute position a = P_i + m are multiplied by log2(a)/log2(L) when a > L, after scale
    and cap.  Rows with nothing visible come out NaN in the reference (0/0); here they stay 0.
    Returns o float64 [T,Hq,Dv] (+ lse [T,Hq])."""
    t, hq, dq = q_extend.shape
    hkv = k_extend.shape[1]
    dv = v_extend.shape[-1]
    group = hq // hkv
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(dq)
    qf, kf, vf = to_f64(q_extend), to_f64(k_extend), to_f64(v_extend)
    o = np.zeros((t, hq, dv), dtype=np.float64)
    oa = np.zeros((t, hq, dv), dtype=np.float64) if return_absw else None
    lse = np.full((t, hq), -np.inf, dtype=np.float64)
    bs = len(qo_indptr) - 1
    for i in range(bs):
        q0, q1 = int(qo_indptr[i]), int(qo_indptr[i + 1])
        e = q1 - q0
        idx = np.asarray(kv_indices[kv_indptr[i] : kv_indptr[i + 1]]).astype(np.int64)
        p_len = idx.size
        cm = None
        if custom_mask is not None:
            woff = int(window_kv_offsets[i]) if window_kv_offsets is not None else 0
            row = woff + p_len + e
            m0 = int(mask_indptr[i])
            cm = np.asarray(custom_mask[m0 : m0 + e * row]).astype(bool).reshape(e, row)[:, woff:]
        for kvh in range(hkv):
            if p_len and not skip_prefix:
                kp = _gather_kv(k_buffer, idx, kvh)
                vp = _gather_kv(v_buffer, idx, kvh)
            else:
                kp = np.zeros((0, dq)); vp = np.zeros((0, dv))
            ke = kf[q0:q1, kvh]
            ve = vf[q0:q1, kvh]
            for h in range(kvh * group, (kvh + 1) * group):
                # All query rows of the (request, head) at once -- the same arithmetic as one row at a time
                # (_extend_attention_rows below keeps that form; tests/test_oracle_golden.py holds the two together).
                if e == 0:
                    continue
                M = np.arange(e)
                Q = qf[q0:q1, h]
                xai = np.ones(e)
                if xai_temperature_len > 0:
                    a = (p_len + M).astype(np.float64)
                    xai = np.where(a > xai_temperature_len, np.log2(np.maximum(a, 1.0)) / math.log2(float(xai_temperature_len)), 1.0)
                bias = None if score_bias is None else np.asarray(score_bias[q0:q1, h], dtype=np.float64)

                def _bias(rel):  # [e, n] distances -> the rows' bias values, 0 outside [0, extent)
                    ok = (rel >= 0) & (rel < bias.shape[1])
                    return np.where(ok, np.take_along_axis(bias, np.clip(rel, 0, bias.shape[1] - 1), axis=1), 0.0)

                parts_s, parts_v = [], []
                if kp.shape[0]:
                    N = np.arange(p_len)
                    s1 = _tanh_cap((Q @ kp.T) * (sm_scale * k_scale), logit_cap) * xai[:, None]
                    if bias is not None:
                        s1 = s1 + _bias((p_len + M)[:, None] - N[None, :])
                    if cm is not None and not skip_prefix_custom_mask:
                        s1 = np.where(cm[:, :p_len], s1, -np.inf)
                    if sliding_window_size > 0:
                        s1 = np.where((p_len + M)[:, None] <= N[None, :] + sliding_window_size, s1, -np.inf)
                    parts_s.append(s1); parts_v.append(vp * v_scale)
                if not skip_extend:
                    N = np.arange(e)
                    s2 = _tanh_cap((Q @ ke.T) * sm_scale, logit_cap) * xai[:, None]
                    if bias is not None:
                        s2 = s2 + _bias(M[:, None] - N[None, :])
                    if cm is not None:
                        s2 = np.where(cm[:, p_len: p_len + e], s2, -np.inf)
                    elif is_causal:  # (row m's list ends at key m)
                        s2 = np.where(N[None, :] <= M[:, None], s2, -np.inf)
                    if sliding_window_size > 0:
                        s2 = np.where(M[:, None] <= N[None, :] + sliding_window_size, s2, -np.inf)
                    parts_s.append(s2); parts_v.append(ve)
                if not parts_s:
                    continue
                sc = np.concatenate(parts_s, axis=1)
                vv = np.concatenate(parts_v, axis=0)
                if sc.shape[1] == 0:
                    continue
                mx = sc.max(axis=1)
                live = np.isfinite(mx)
                pm = np.exp(sc - np.where(live, mx, 0.0)[:, None])
                den = pm.sum(axis=1)
                rows = q0 + M[live]
                with np.errstate(divide="ignore"):
                    lse[rows, h] = (mx + np.log(den))[live]
                if sinks is not None:
                    den = den + np.exp(float(sinks[h]) - np.where(live, mx, 0.0))
                den = np.where(live, den, 1.0)
                o[rows, h] = ((pm @ vv) / den[:, None])[live]
                if return_absw:
                    oa[rows, h] = ((pm @ np.abs(vv)) / den[:, None])[live]
    res = (o, lse) if return_lse else o
    if return_absw:
        return (res + (oa,)) if return_lse else (o, oa)
    return res


def _extend_attention_rows(q_extend, k_extend, v_extend, k_buffer, v_buffer, qo_indptr,
                     kv_indptr, kv_indices, is_causal=True, sm_scale=None,
                     k_scale=1.0, v_scale=1.0, logit_cap=0.0,
                     sliding_window_size=-1, sinks=None, skip_prefix=False,
                     skip_extend=False, return_lse=False, custom_mask=None, mask_indptr=None,
                     skip_prefix_custom_mask=True, window_kv_offsets=None, xai_temperature_len=-1, score_bias=None,
                     return_absw=False):
    """extend_attention one query row at a time: the form the restatement was pinned in (rounds 1-5); kept as the cross-check of
    the row-vectorised function above (same arguments, same results up to fp64 summation order).
    return_absw: also return sum_j p_j |v_j| (see decode_attention) as the last element of the result.
    score_bias [T, Hq, extent] (float): score_mod = relative_bias_score_mod, aux_tensors = [score_bias]
    (extend_attention.py:463-476 prefix stage: q_pos = P + m, kv_pos = list position; :594-607 extend stage:
    kv_pos = P + n; q_idx = global query token), added after scale, cap and temperature, before the masks.
    Semantics of extend_attention_fwd (extend_attention.py:664-812).  Request i
    has prefix tokens kv_indices[kv_indptr[i]:kv_indptr[i+1]] read from the cache
    (stage 1, :372-510; scaled by k_scale / v_scale) and E_i = qo_indptr[i+1]-
    qo_indptr[i] new tokens whose K/V are the contiguous k_extend/v_extend rows
    (stage 2, :512-631).  Query m (0-based inside the extend part) sees every
    prefix token and extend tokens n <= m when causal, all E_i otherwise.
    sliding window W>0: q_abs <= kv_abs + W (:385-390, :556-561).
    custom_mask (speculative tree attention, :320-326, :378-390, :525-539): request i owns the flat
    bytes mask[mask_indptr[i]:], a row-major [E_i, woff_i + P_i + E_i] matrix (woff = window_kv_offsets,
    0 without SWA); in the extend part it REPLACES the causal mask, in the prefix part it applies
    unless skip_prefix_custom_mask.  xai_temperature_len L>0 (:336-343, :460, :591): scores of the
    query at absolute position a = P_i + m are multiplied by log2(a)/log2(L) when a > L, after scale
    and cap.  Rows with nothing visible come out NaN in the reference (0/0); here they stay 0.
    Returns o float64 [T,Hq,Dv] (+ lse [T,Hq])."""
    t, hq, dq = q_extend.shape
    hkv = k_extend.shape[1]
    dv = v_extend.shape[-1]
    group = hq // hkv
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(dq)
    qf, kf, vf = to_f64(q_extend), to_f64(k_extend), to_f64(v_extend)
    o = np.zeros((t, hq, dv), dtype=np.float64)
    oa = np.zeros((t, hq, dv), dtype=np.float64) if return_absw else None
    lse = np.full((t, hq), -np.inf, dtype=np.float64)
    bs = len(qo_indptr) - 1
    for i in range(bs):
        q0, q1 = int(qo_indptr[i]), int(qo_indptr[i + 1])
        e = q1 - q0
        idx = np.asarray(kv_indices[kv_indptr[i] : kv_indptr[i + 1]]).astype(np.int64)
        p_len = idx.size
        cm = None
        if custom_mask is not None:
            woff = int(window_kv_offsets[i]) if window_kv_offsets is not None else 0
            row = woff + p_len + e
            m0 = int(mask_indptr[i])
            cm = np.asarray(custom_mask[m0 : m0 + e * row]).astype(bool).reshape(e, row)[:, woff:]
        for kvh in range(hkv):
            if p_len and not skip_prefix:
                kp = _gather_kv(k_buffer, idx, kvh)
                vp = _gather_kv(v_buffer, idx, kvh)
            else:
                kp = np.zeros((0, dq)); vp = np.zeros((0, dv))
            ke = kf[q0:q1, kvh]
            ve = vf[q0:q1, kvh]
            for h in range(kvh * group, (kvh + 1) * group):
                for m in range(e):
                    parts_s, parts_v = [], []
                    xai = 1.0
                    if xai_temperature_len > 0 and (p_len + m) > xai_temperature_len:
                        xai = math.log2(p_len + m) / math.log2(float(xai_temperature_len))
                    if kp.shape[0]:
                        s1 = _tanh_cap(kp @ qf[q0 + m, h] * (sm_scale * k_scale), logit_cap) * xai
                        if score_bias is not None:
                            s1 = s1 + _rel_bias(np.asarray(score_bias[q0 + m, h], dtype=np.float64), p_len + m, np.arange(p_len))
                        if cm is not None and not skip_prefix_custom_mask:
                            s1 = np.where(cm[m, :p_len], s1, -np.inf)
                        if sliding_window_size > 0:
                            keep = (p_len + m) <= (np.arange(p_len) + sliding_window_size)
                            s1 = np.where(keep, s1, -np.inf)
                        parts_s.append(s1); parts_v.append(vp * v_scale)
                    if not skip_extend:
                        n_end = (m + 1) if (is_causal and cm is None) else e
                        s2 = _tanh_cap(ke[:n_end] @ qf[q0 + m, h] * sm_scale, logit_cap) * xai
                        if score_bias is not None:
                            s2 = s2 + _rel_bias(np.asarray(score_bias[q0 + m, h], dtype=np.float64), p_len + m, p_len + np.arange(n_end))
                        if cm is not None:
                            s2 = np.where(cm[m, p_len : p_len + n_end], s2, -np.inf)
                        if sliding_window_size > 0:
                            keep = m <= (np.arange(n_end) + sliding_window_size)
                            s2 = np.where(keep, s2, -np.inf)
                        parts_s.append(s2); parts_v.append(ve[:n_end])
                    if not parts_s:
                        continue
                    s = np.concatenate(parts_s)
                    vv = np.concatenate(parts_v, axis=0)
                    mx = s.max()
                    if not np.isfinite(mx):
                        continue
                    p = np.exp(s - mx)
                    den = p.sum()
                    lse[q0 + m, h] = mx + math.log(den)
                    if sinks is not None:
                        den = den + math.exp(float(sinks[h]) - mx)
                    o[q0 + m, h] = (p @ vv) / den
                    if return_absw:
                        oa[q0 + m, h] = (p @ np.abs(vv)) / den
    res = (o, lse) if return_lse else o
    if return_absw:
        return (res + (oa,)) if return_lse else (o, oa)
    return res


def extend_attention_unified(q, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, prefix_lens, sm_scale=None,
                             k_scale=1.0, v_scale=1.0, logit_cap=0.0, is_causal=True, sliding_window_size=-1,
                             sinks=None, custom_mask=None, mask_indptr=None, xai_temperature_len=-1, score_bias=None,
                             return_absw=False):
    """return_absw: also return sum_j p_j |v_j| (see decode_attention).
    score_bias [T, Hq, extent]: relative_bias_score_mod through :1093-1104 (q_pos = prefix_i + m, kv_pos = list position).
    Semantics of extend_attention_fwd_unified / _fwd_kernel_unified (extend_attention.py:852-1158): one pass
    over a kv list holding prefix + new tokens.  Query m of request i sees list position n iff n < prefix_i or
    n - prefix_i <= m (causal, :993-1008), window prefix_i + m <= n + W (:1010-1027); a custom mask row is
    kv_len wide and replaces the causal rule (:980-990); xai factor L / (prefix_i + m + 1) once
    prefix_i + m >= L (:940-946)."""
    t, hq, dq = q.shape
    hkv = k_buffer.shape[-2]
    dv = v_buffer.shape[-1]
    group = hq // hkv
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(dq)
    qf = to_f64(q)
    o = np.zeros((t, hq, dv), dtype=np.float64)
    oa = np.zeros((t, hq, dv), dtype=np.float64) if return_absw else None
    for i in range(len(qo_indptr) - 1):
        q0, q1 = int(qo_indptr[i]), int(qo_indptr[i + 1])
        e = q1 - q0
        idx = np.asarray(kv_indices[kv_indptr[i]: kv_indptr[i + 1]]).astype(np.int64)
        n_kv, pre = idx.size, int(prefix_lens[i])
        cm = None
        if custom_mask is not None:
            m0 = int(mask_indptr[i])
            cm = np.asarray(custom_mask[m0: m0 + e * n_kv]).astype(bool).reshape(e, n_kv)
        pos = np.arange(n_kv)
        for kvh in range(hkv):
            kk, vv = _gather_kv(k_buffer, idx, kvh), _gather_kv(v_buffer, idx, kvh)
            for h in range(kvh * group, (kvh + 1) * group):
                if e == 0 or n_kv == 0:
                    continue
                M = np.arange(e)
                Q = qf[q0:q1, h]
                xai = np.ones(e)
                if xai_temperature_len > 0:
                    xai = np.where(pre + M >= xai_temperature_len, xai_temperature_len / (pre + M + 1.0), 1.0)
                sc = _tanh_cap((Q @ kk.T) * (sm_scale * k_scale), logit_cap) * xai[:, None]
                if score_bias is not None:
                    bias = np.asarray(score_bias[q0:q1, h], dtype=np.float64)
                    rel = (pre + M)[:, None] - pos[None, :]
                    ok = (rel >= 0) & (rel < bias.shape[1])
                    sc = sc + np.where(ok, np.take_along_axis(bias, np.clip(rel, 0, bias.shape[1] - 1), axis=1), 0.0)
                keep = np.ones((e, n_kv), dtype=bool)
                if cm is not None:
                    keep &= cm
                elif is_causal:
                    keep &= (pos[None, :] < pre) | (pos[None, :] - pre <= M[:, None])
                if sliding_window_size > 0:
                    keep &= (pre + M)[:, None] <= (pos[None, :] + sliding_window_size)
                sc = np.where(keep, sc, -np.inf)
                mx = sc.max(axis=1)
                live = np.isfinite(mx)
                pm = np.exp(sc - np.where(live, mx, 0.0)[:, None])
                den = pm.sum(axis=1)
                if sinks is not None:
                    den = den + np.exp(float(sinks[h]) - np.where(live, mx, 0.0))
                den = np.where(live, den, 1.0)
                rows = q0 + M[live]
                o[rows, h] = ((pm @ vv) / den[:, None] * v_scale)[live]
                if return_absw:
                    oa[rows, h] = ((pm @ np.abs(vv)) / den[:, None] * v_scale)[live]
    return (o, oa) if return_absw else o


def _extend_attention_unified_rows(q, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, prefix_lens, sm_scale=None,
                             k_scale=1.0, v_scale=1.0, logit_cap=0.0, is_causal=True, sliding_window_size=-1,
                             sinks=None, custom_mask=None, mask_indptr=None, xai_temperature_len=-1, score_bias=None,
                             return_absw=False):
    """extend_attention_unified one query row at a time (the form it was pinned in; the cross-check of the row-vectorised
    function above).  return_absw: also return sum_j p_j |v_j| (see decode_attention).
    score_bias [T, Hq, extent]: relative_bias_score_mod through :1093-1104 (q_pos = prefix_i + m, kv_pos = list position).
    Semantics of extend_attention_fwd_unified / _fwd_kernel_unified (extend_attention.py:852-1158): one pass
    over a kv list holding prefix + new tokens.  Query m of request i sees list position n iff n < prefix_i or
    n - prefix_i <= m (causal, :993-1008), window prefix_i + m <= n + W (:1010-1027); a custom mask row is
    kv_len wide and replaces the causal rule (:980-990); xai factor L / (prefix_i + m + 1) once
    prefix_i + m >= L (:940-946)."""
    t, hq, dq = q.shape
    hkv = k_buffer.shape[-2]
    dv = v_buffer.shape[-1]
    group = hq // hkv
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(dq)
    qf = to_f64(q)
    o = np.zeros((t, hq, dv), dtype=np.float64)
    oa = np.zeros((t, hq, dv), dtype=np.float64) if return_absw else None
    for i in range(len(qo_indptr) - 1):
        q0, q1 = int(qo_indptr[i]), int(qo_indptr[i + 1])
        e = q1 - q0
        idx = np.asarray(kv_indices[kv_indptr[i]: kv_indptr[i + 1]]).astype(np.int64)
        n_kv, pre = idx.size, int(prefix_lens[i])
        cm = None
        if custom_mask is not None:
            m0 = int(mask_indptr[i])
            cm = np.asarray(custom_mask[m0: m0 + e * n_kv]).astype(bool).reshape(e, n_kv)
        pos = np.arange(n_kv)
        for kvh in range(hkv):
            kk, vv = _gather_kv(k_buffer, idx, kvh), _gather_kv(v_buffer, idx, kvh)
            for h in range(kvh * group, (kvh + 1) * group):
                for m in range(e):
                    xai = 1.0
                    if xai_temperature_len > 0 and pre + m >= xai_temperature_len:
                        xai = xai_temperature_len / (pre + m + 1.0)
                    s = _tanh_cap(kk @ qf[q0 + m, h] * (sm_scale * k_scale), logit_cap) * xai
                    if score_bias is not None:
                        s = s + _rel_bias(np.asarray(score_bias[q0 + m, h], dtype=np.float64), pre + m, pos)
                    keep = np.ones(n_kv, dtype=bool)
                    if cm is not None:
                        keep &= cm[m]
                    elif is_causal:
                        keep &= (pos < pre) | (pos - pre <= m)
                    if sliding_window_size > 0:
                        keep &= (pre + m) <= (pos + sliding_window_size)
                    s = np.where(keep, s, -np.inf)
                    mx = s.max() if n_kv else -np.inf
                    if not np.isfinite(mx):
                        continue
                    p = np.exp(s - mx)
                    den = p.sum()
                    if sinks is not None:
                        den = den + math.exp(float(sinks[h]) - mx)
                    o[q0 + m, h] = (p @ vv) / den * v_scale
                    if return_absw:
                        oa[q0 + m, h] = (p @ np.abs(vv)) / den * v_scale
    return (o, oa) if return_absw else o


def rope(x, positions, cos_sin_cache, is_neox, rotary_dim=None):
    """RotaryEmbedding._apply_rotary_emb semantics (srt/layers/rotary_embedding: cos_sin_cache[pos] =
    [cos(rot/2) | sin(rot/2)]; neox pairs (i, i+rot/2), gptj pairs (2i, 2i+1)) in float64.
    x [n, H, D] -> float64 [n, H, D]; columns >= rotary_dim pass through."""
    x = to_f64(x).copy()
    cs = np.asarray(cos_sin_cache, dtype=np.float64)[np.asarray(positions)]
    rot = int(rotary_dim or cs.shape[-1])
    half = rot // 2
    cos, sin = cs[:, None, :half], cs[:, None, half:rot]
    if is_neox:
        x0, x1 = x[..., :half].copy(), x[..., half:rot].copy()
        x[..., :half], x[..., half:rot] = x0 * cos - x1 * sin, x1 * cos + x0 * sin
    else:
        x0, x1 = x[..., 0:rot:2].copy(), x[..., 1:rot:2].copy()
        x[..., 0:rot:2], x[..., 1:rot:2] = x0 * cos - x1 * sin, x1 * cos + x0 * sin
    return x


def qknorm_rope_freqs(rotary_dim, base, factor=1.0, low=0.0, high=0.0):
    """compute_freq of the reference kernel (kernels/jit/csrc/elementwise/fused_qknorm_rope.cuh:42-63): freq_p =
    base^(-2 p / rotary_dim) for p < rotary_dim / 2; with YaRN (factor != 1) blended with freq_p / factor by the ramp
    clamp((p - low) / (high' - low), 0, 1), high' = high + 0.001 when |low - high| <= 1e-6.  float64 [rotary_dim / 2]."""
    p = np.arange(int(rotary_dim) // 2, dtype=np.float64)
    freq = np.power(float(base), -2.0 * p / float(rotary_dim))
    if float(factor) != 1.0:
        high_adj = float(high) + 0.001 if abs(float(low) - float(high)) <= 1e-6 else float(high)
        ramp = np.clip((p - float(low)) / (high_adj - float(low)), 0.0, 1.0)
        extr = 1.0 - ramp
        freq = (freq / float(factor)) * (1.0 - extr) + freq * extr
    return freq


def fused_qk_norm_rope(q, k, q_weight, k_weight, positions, eps, base, is_neox, factor=1.0, low=0.0, high=0.0,
                       attention_factor=1.0, rotary_dim=None, cos_sin_cache=None):
    """fused_qk_norm_rope (kernels/ops/attention/fused_qknorm_rope.py:37-100; kernel fused_qknorm_rope.cuh:78-246), in
    float64 and WITHOUT the final rounding: per (token, head) x * rsqrt(mean(x^2) + eps) * w (:131-155), then RoPE on
    the first rotary_dim columns -- neox pairs (p, p + rot/2) (:206-230), else (2p, 2p + 1) (:168-203) -- with
    angle = position * freq_p (qknorm_rope_freqs) and the rotated pair times attention_factor; columns >= rotary_dim
    stay normalised only (:236-245).  cos_sin_cache (float [max_pos, rot] = [cos | sin]): the angles' cos / sin come from
    there instead (this library's second form).  q [n, Hq, D], k [n, Hkv, D] (uint16 = bf16 bits, or float) ->
    (q', k') float64."""
    pos = np.asarray(positions).astype(np.int64)
    outs = []
    for x, w in ((q, q_weight), (k, k_weight)):
        xf, wf = to_f64(x), to_f64(w).reshape(-1)
        d = xf.shape[-1]
        rot = int(rotary_dim or d)
        half = rot // 2
        y = xf / np.sqrt((xf * xf).mean(axis=-1, keepdims=True) + float(eps)) * wf
        if cos_sin_cache is not None:
            cs = np.asarray(cos_sin_cache, dtype=np.float64)[pos]
            cos, sin = cs[:, None, :half], cs[:, None, half:rot]
        else:
            ang = pos[:, None].astype(np.float64) * qknorm_rope_freqs(rot, base, factor, low, high)[None, :]
            cos, sin = np.cos(ang)[:, None, :], np.sin(ang)[:, None, :]
        o = y.copy()
        if is_neox:
            y0, y1 = y[..., :half], y[..., half:rot]
            o[..., :half], o[..., half:rot] = (y0 * cos - y1 * sin) * attention_factor, (y1 * cos + y0 * sin) * attention_factor
        else:
            y0, y1 = y[..., 0:rot:2], y[..., 1:rot:2]
            o[..., 0:rot:2], o[..., 1:rot:2] = (y0 * cos - y1 * sin) * attention_factor, (y1 * cos + y0 * sin) * attention_factor
        outs.append(o)
    return outs[0], outs[1]


def decode_attention_grouped_rope(q, k_buffer, kv_indptr, kv_indices, cos_sin_cache, positions, sm_scale,
                                  kv_lora_rank=512, is_neox=True, logit_cap=0.0, new_rows=None):
    """Semantics of decode_attention_fwd_grouped_rope (kernels/ops/attention/rocm_mla_decode_rope.py:45-439) with
    use_rope: q [bs, Hq, c + r] holds [q_nope | q_pe NOT rotated]; k_buffer rows [c + r] hold the latent and the k_pe --
    rotated for every cached token EXCEPT each request's newest one (the last index of its kv_indices list), which the
    kernel rotates itself (:180-199, used at :228-236) at positions[b], like q_pe (:119-178).  Scores
    (q_nope . kv + q_pe' . k_pe') sm_scale (:238-252), values = the first c columns of the same rows.
    new_rows [bs, c + r] (this library's k_new form): the newest rows come from there instead of the pool.
    Returns (o float64 [bs, Hq, c], k_pe_out float64 [bs, r]: the newest tokens' rotated k_pe, :285-292)."""
    qf, kb = to_f64(q), to_f64(k_buffer)
    kb = kb.reshape(kb.shape[0], -1).copy()
    bs, hq, dk = qf.shape
    c = int(kv_lora_rank)
    r = dk - c
    pos = np.asarray(positions)
    q_rot = qf.copy()
    q_rot[..., c:] = rope(qf[..., c:], pos, cos_sin_cache, is_neox, r)
    o = np.zeros((bs, hq, c))
    kpe_out = np.zeros((bs, r))
    for b in range(bs):
        idx = np.asarray(kv_indices[int(kv_indptr[b]): int(kv_indptr[b + 1])]).astype(np.int64)
        if idx.size == 0:
            continue
        rows = kb[idx].copy()
        newest = rows[-1] if new_rows is None else to_f64(new_rows)[b].reshape(-1).copy()
        newest[c:] = rope(newest[None, None, c:], pos[b: b + 1], cos_sin_cache, is_neox, r)[0, 0]
        rows[-1] = newest
        kpe_out[b] = newest[c:]
        s_ = (q_rot[b] @ rows.T) * sm_scale
        if logit_cap > 0:
            s_ = logit_cap * np.tanh(s_ / logit_cap)
        s_ -= s_.max(axis=-1, keepdims=True)
        p_ = np.exp(s_)
        o[b] = (p_ / p_.sum(axis=-1, keepdims=True)) @ rows[:, :c]
    return o, kpe_out


def merge_state(a, lse_a, b, lse_b):
    """merge_state_triton (kernels/ops/attention/merge_state.py:8-64): LSE-weighted blend of two partial
    attention outputs; a +inf LSE is read as -inf.  Returns (out float64, out_lse float64)."""
    a, b = to_f64(a), to_f64(b)
    la = np.asarray(lse_a, dtype=np.float64).copy()
    lb = np.asarray(lse_b, dtype=np.float64).copy()
    la[la == np.inf] = -np.inf
    lb[lb == np.inf] = -np.inf
    m = np.maximum(la, lb)
    with np.errstate(invalid="ignore"):
        wa, wb = np.exp(la - m), np.exp(lb - m)
    se = wa + wb
    out = a * (wa / se)[..., None] + b * (wb / se)[..., None]
    return out, np.log(se) + m


# --------------------------------------------------------------------------
# decode context parallel (8e)   srt/layers/dcp/layout.py, kernels/ops/attention/dcp_kernels.py, srt/layers/dcp/comm.py
# --------------------------------------------------------------------------
def dcp_lens(lens, dcp_size, dcp_rank, start=None):
    """get_dcp_lens (layout.py:23-41): tokens of [start, start + lens) at positions p with p % dcp_size == dcp_rank."""
    lens = np.asarray(lens, dtype=np.int64)
    if dcp_size == 1:
        return lens
    if start is None:
        return lens // dcp_size + (dcp_rank < lens % dcp_size)
    start = np.asarray(start, dtype=np.int64)
    first = start + np.remainder(dcp_rank - start, dcp_size)
    remaining = start + lens - first
    return np.maximum((remaining + dcp_size - 1) // dcp_size, 0)


def dcp_kv_indices(req_to_token, req_pool_indices, lens, dcp_size, dcp_rank, kv_start=None, out_dtype=np.int64):
    """TritonAttnBackend._dcp_kv_indices (triton_backend.py:356-384) = dcp_lens + cumsum + the strided gather of
    create_triton_kv_indices_for_dcp_triton (dcp_kernels.py:34-76): the rank's tokens of every request as LOCAL slots
    (virtual slot // dcp_size).  Returns (kv_indptr int32, kv_indices, dcp_lens)."""
    req_to_token = np.asarray(req_to_token)
    dl = dcp_lens(lens, dcp_size, dcp_rank, kv_start)
    bs = len(dl)
    kv_indptr = np.zeros((bs + 1,), dtype=np.int32)
    kv_indptr[1:] = np.cumsum(dl)
    kv_indices = np.empty((int(kv_indptr[-1]),), dtype=out_dtype)
    for i in range(bs):
        s = 0 if kv_start is None else int(kv_start[i])
        first = s + (dcp_rank - s) % dcp_size
        pos = first + np.arange(int(dl[i]), dtype=np.int64) * dcp_size
        kv_indices[kv_indptr[i]: kv_indptr[i + 1]] = req_to_token[int(req_pool_indices[i]), pos] // dcp_size
    return kv_indptr, kv_indices, dl.astype(np.int32)


def dcp_store_loc(out_cache_loc, positions, dcp_size, dcp_rank, skip_index=0):
    """_set_kv_buffer's DCP branch (triton_backend.py:1227-1239): local slot for owned tokens, skip_index otherwise."""
    loc = np.asarray(out_cache_loc, dtype=np.int64)
    pos = np.asarray(positions, dtype=np.int64)
    return np.where(pos % dcp_size == dcp_rank, loc // dcp_size, skip_index).astype(np.int64)


def dcp_merge(outs, lses):
    """cp_lse_ag_out_rs_mha (comm.py:82-108) for all ranks at once: outs [dcp, T, H, D], lses [dcp, T, H] (natural log;
    -inf = the rank saw no token; NaN / inf outputs of such rows count as 0).  Returns (scaled [dcp, T, H, D] -- what
    each rank contributes to the all-reduce --, summed [T, H, D], global_lse [T, H]); rank r keeps heads
    [r * H / dcp, (r + 1) * H / dcp)."""
    outs = np.asarray(outs, dtype=np.float64)
    lses = np.asarray(lses, dtype=np.float64)
    m = lses.max(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        g = np.where(np.isfinite(m), m + np.log(np.exp(lses - np.where(np.isfinite(m), m, 0.0)).sum(axis=0)), m)
        scale = np.exp(lses - g)
    scale = np.where(np.isfinite(scale), scale, 0.0)
    scaled = np.where(np.isfinite(outs), outs, 0.0) * scale[..., None]
    return scaled, scaled.sum(axis=0), g


# --------------------------------------------------------------------------
# C3 quick all-reduce   kernels/aot/csrc/allreduce/quick_all_reduce.cuh:52-632 (codecs, two-shot schedule),
#                       quick_all_reduce_base.h:98-300 (packed 16-bit arithmetic, group_abs_max)
# Parity status of THIS section: the reference's kernel is HIP-only, so nothing here could be pinned against an execution of
# it in the build container.  It is pinned to (a) the reference's own test properties (test/manual/test_quick_allreduce.py:
# level FP is exact on small integers, INT4 of all-ones is exactly `world`, zeros stay zeros, integers in [1, 23) land within
# atol 1.25 W / rtol 0.5 W at every level) and (b) the hardware: the one instruction whose result the ISA leaves open,
# v_rcp_f16, enters as a TABLE (tests/golden/rcp_f16_gfx950.npy, read back from an MI355X through rx_rcp_f16_table; the GPU
# test re-reads it live), every other step is an IEEE operation with one rounding.
# --------------------------------------------------------------------------
QR_FP, QR_INT8, QR_INT6, QR_INT4 = 0, 1, 2, 3   # QuickReduceRegime, quick_all_reduce.py:41-46
_QR_BITS = {QR_FP: 16, QR_INT8: 8, QR_INT6: 6, QR_INT4: 4}


def rcp_f16_correctly_rounded() -> np.ndarray:
    """1 / x for all 65536 fp16 bit patterns, correctly rounded (the stand-in where no hardware table is at hand)."""
    with np.errstate(all="ignore"):
        x = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float64)
        return (1.0 / x).astype(np.float16).view(np.uint16)


class _QrF16:
    """Packed-fp16 instruction semantics on uint16 bit patterns: exact in float64, ONE rounding to fp16."""
    eps = np.uint16(0x0001)   # kScaleEpsilon: fp16's smallest subnormal

    def __init__(self, rcp_table=None):
        self.table = rcp_f16_correctly_rounded() if rcp_table is None else np.asarray(rcp_table, dtype=np.uint16)

    @staticmethod
    def val(b):
        return np.asarray(b, dtype=np.uint16).view(np.float16).astype(np.float64)

    @staticmethod
    def rnd(x):
        return np.asarray(x, dtype=np.float64).astype(np.float16).view(np.uint16)

    def const(self, v):
        return self.rnd(np.float64(v))

    def mul(self, a, b):
        return self.rnd(self.val(a) * self.val(b))

    def add(self, a, b):
        return self.rnd(self.val(a) + self.val(b))

    @staticmethod
    def _key(b):   # total order of the non-NaN patterns: -0 below +0 (v_pk_max_f16 / v_pk_min_f16 order the zeros)
        b = np.asarray(b, dtype=np.uint16).astype(np.int32)
        return np.where(b & 0x8000, 0xFFFF - b, b | 0x8000)

    def vmax(self, a, b):
        a, b = np.broadcast_arrays(np.asarray(a, np.uint16), np.asarray(b, np.uint16))
        na, nb = np.isnan(self.val(a)), np.isnan(self.val(b))
        return np.where(na, b, np.where(nb, a, np.where(self._key(a) >= self._key(b), a, b))).astype(np.uint16)

    def vmin(self, a, b):
        a, b = np.broadcast_arrays(np.asarray(a, np.uint16), np.asarray(b, np.uint16))
        na, nb = np.isnan(self.val(a)), np.isnan(self.val(b))
        return np.where(na, b, np.where(nb, a, np.where(self._key(a) <= self._key(b), a, b))).astype(np.uint16)

    def rcp(self, a):
        return self.table[np.asarray(a, dtype=np.uint16)]

    def pick_abs(self, a, b):   # __hgt(|a|, |b|) ? a : b
        return np.where(np.abs(self.val(a)) > np.abs(self.val(b)), a, b).astype(np.uint16)


class _QrBF16:
    """The HIP bf16 operators: fp32 arithmetic, round to nearest even after every operation."""
    eps = np.uint16(0x33D7)

    @staticmethod
    def val(b):
        return bf16_to_f32(np.asarray(b, dtype=np.uint16))

    @staticmethod
    def rnd(x):
        return f32_to_bf16(np.asarray(x, dtype=np.float32))

    def const(self, v):
        return self.rnd(np.float32(v))

    def mul(self, a, b):
        return self.rnd(self.val(a) * self.val(b))

    def add(self, a, b):
        return self.rnd(self.val(a) + self.val(b))

    def vmax(self, a, b):   # __hmax: a NaN loses; a > b ? a : b -- the SECOND operand on equality (+0 / -0)
        a, b = np.broadcast_arrays(np.asarray(a, np.uint16), np.asarray(b, np.uint16))
        va, vb = self.val(a), self.val(b)
        return np.where(np.isnan(va), b, np.where(np.isnan(vb), a, np.where(va > vb, a, b))).astype(np.uint16)

    def vmin(self, a, b):
        a, b = np.broadcast_arrays(np.asarray(a, np.uint16), np.asarray(b, np.uint16))
        va, vb = self.val(a), self.val(b)
        return np.where(np.isnan(va), b, np.where(np.isnan(vb), a, np.where(va < vb, a, b))).astype(np.uint16)

    def rcp(self, a):   # hrcp: __float2bfloat16(1.0f / x)
        return self.rnd(np.float32(1.0) / self.val(a))

    def pick_abs(self, a, b):
        return np.where(np.abs(self.val(a)) > np.abs(self.val(b)), a, b).astype(np.uint16)


def _qr_block_scale(num, x, bits):
    """x: uint16 [G, 64] (groups of 64 consecutive elements) -> decode scale per (group, parity) [G, 2].  A lane holds 8
    consecutive elements as 4 packed pairs; the comparisons run pair 0-1, pair 2-3, then both, then a binary tree over the 8
    lanes towards lane 0, own value first (group_abs_max, quick_all_reduce_base.h:268-300)."""
    v = x.reshape(-1, 8, 4, 2)   # [group, lane, pair, parity]

    def reduce(op):
        m = op(op(v[:, :, 0], v[:, :, 1]), op(v[:, :, 2], v[:, :, 3]))   # [G, lane, parity]
        m = op(m[:, 0:7], m[:, 1:8])        # lane l <- (l, l + 1); lanes 0, 2, 4, 6 are used
        m = op(m[:, 0:5], m[:, 2:7])        # lane l <- (l, l + 2); lanes 0, 4
        return op(m[:, 0], m[:, 4])         # [G, parity]

    mx, mn = reduce(num.vmax), reduce(num.vmin)
    return num.mul(num.pick_abs(mx, mn), num.const(-1.0 / (1 << (bits - 1))))


def qr_codec_roundtrip(num, x, bits):
    """decode(encode(x)) for one rank's padded message x (uint16 bits, a multiple of 64 elements), and the codes."""
    r = 1 << (bits - 1)
    g = x.reshape(-1, 64)
    d = _qr_block_scale(num, g, bits)                               # [G, 2]
    e = num.rcp(num.add(d, num.eps))
    v = g.reshape(-1, 32, 2)
    w = num.vmin(num.vmax(num.mul(v, e[:, None, :]), num.const(-r)), num.const(r - 1))
    code = np.rint(num.val(w).astype(np.float64)).astype(np.int64) + r      # rintf: ties to even
    out = num.mul(num.rnd((code - r).astype(np.float64 if isinstance(num, _QrF16) else np.float32)), d[:, None, :])
    return out.reshape(-1), code.reshape(-1)


def quick_allreduce(parts: Sequence[np.ndarray], is_bf16: bool, level: int, cast_bf16_to_fp16: bool = False,
                    rcp_f16_table: Optional[np.ndarray] = None) -> np.ndarray:
    """AllReduceTwoshot<T, Codec, cast_bf2half>::run (quick_all_reduce.cuh:448-632) for all ranks at once.
    parts: the `world` inputs as uint16 bit patterns of the tensor's dtype (fp16, or bf16 with is_bf16), equal lengths (a
    multiple of 8).  Returns the bits every rank ends with.  See the section comment for the arithmetic and its pinning."""
    bits = _QR_BITS[level]
    n = len(parts[0])
    assert n % 8 == 0 and all(len(p) == n for p in parts)
    with np.errstate(all="ignore"):
        as_f16 = (not is_bf16) or cast_bf16_to_fp16
        num = _QrF16(rcp_f16_table) if as_f16 else _QrBF16()
        xs = []
        for p in parts:
            p = np.ascontiguousarray(p, dtype=np.uint16).reshape(-1)
            if is_bf16 and cast_bf16_to_fp16:   # __float22half2_rn(__bfloat1622float2(x))
                p = bf16_to_f32(p).astype(np.float16).view(np.uint16)
            xs.append(np.concatenate([p, np.zeros((-n) % 64, dtype=np.uint16)]))
        acc = np.zeros_like(xs[0])              # +0
        for x in xs:                            # rank order, every partial sum rounded to the type
            acc = num.add(acc, x if bits == 16 else qr_codec_roundtrip(num, x, bits)[0])
        out = acc if bits == 16 else qr_codec_roundtrip(num, acc, bits)[0]
        out = out[:n]
        if is_bf16 and cast_bf16_to_fp16:       # __float22bfloat162_rn(__half22float2(x))
            out = f32_to_bf16(out.view(np.float16).astype(np.float32))
        return out


# --------------------------------------------------------------------------
# a14 torch-native semantics   srt/layers/attention/torch_native_backend.py:61-277
# --------------------------------------------------------------------------


def sdpa_decode_req_to_token(q, k_cache, v_cache, req_to_token, req_pool_indices,
                             seq_lens, scaling, sliding_window_size=-1):
    """_run_sdpa_forward_decode (:176-277): per request, K/V =
    cache[req_to_token[req_pool_idx, :seq_len]], one query token at position seq_len - 1, no mask; with a sliding
    window W the mask of _make_sliding_window_mask (:36-48): keys q_pos - W .. q_pos (W + 1 of them)."""
    seq_lens = np.asarray(seq_lens, dtype=np.int64)
    if sliding_window_size is None or sliding_window_size < 0:
        kv_indptr, kv_indices = build_kv_indices(req_to_token, req_pool_indices, seq_lens)
    else:
        wl = np.minimum(seq_lens, sliding_window_size + 1)
        kv_indptr, kv_indices = build_kv_indices(req_to_token, req_pool_indices, wl, kv_start=seq_lens - wl)
    return decode_attention(q, k_cache, v_cache, kv_indptr, kv_indices, scaling)


def sdpa_extend_req_to_token(q, k_cache, v_cache, req_to_token, req_pool_indices,
                             seq_lens, extend_prefix_lens, extend_seq_lens, scaling,
                             causal=True, sliding_window_size=-1):
    """_run_sdpa_forward_extend (:61-174): the new tokens' K/V are read back from
    the cache (they were stored before the call), queries sit at positions
    prefix..seq-1 of a causal mask over the whole sequence; with a sliding window W the mask
    (k_pos <= q_pos) & (k_pos >= q_pos - W) replaces the causal one (:147-157)."""
    t, hq, _ = q.shape
    hkv = k_cache.shape[-2]
    group = hq // hkv
    dv = v_cache.shape[-1]
    qf = to_f64(q)
    o = np.zeros((t, hq, dv), dtype=np.float64)
    windowed = sliding_window_size is not None and sliding_window_size >= 0
    start = 0
    for i in range(len(seq_lens)):
        e, pre, seq = int(extend_seq_lens[i]), int(extend_prefix_lens[i]), int(seq_lens[i])
        idx = np.asarray(req_to_token[int(req_pool_indices[i]), :seq]).astype(np.int64)
        for kvh in range(hkv):
            kk = _gather_kv(k_cache, idx, kvh)
            vv = _gather_kv(v_cache, idx, kvh)
            for h in range(kvh * group, (kvh + 1) * group):
                for m in range(e):
                    n_end = pre + m + 1 if (causal or windowed) else seq
                    n_lo = max(0, pre + m - sliding_window_size) if windowed else 0
                    s = kk[n_lo:n_end] @ qf[start + m, h] * scaling
                    p = np.exp(s - s.max())
                    o[start + m, h] = (p @ vv[n_lo:n_end]) / p.sum()
        start += e
    return o


# --------------------------------------------------------------------------
# a16 RadixCache     srt/mem_cache/radix_cache.py:279-812, evict_policy.py
# --------------------------------------------------------------------------


class _RNode:
    __slots__ = ("id", "parent", "key", "value", "children", "lock_ref", "hit_count", "priority",
                 "last_access", "creation", "extra")

    def __init__(self, nid, now, priority=0):
        self.id, self.parent, self.key, self.value = nid, None, [], []
        self.children, self.lock_ref, self.hit_count, self.priority = {}, 0, 0, priority
        self.last_access = self.creation = now
        self.extra = None


class RadixTreeOracle:
    """Pure-Python restatement of RadixCache's tree logic (no torch): match_prefix with node
    split (:352-410, :642-694), insert (:704-757), lock refs (:592-626), heap eviction over
    evictable leaves (:562-590) with the policy keys of evict_policy.py.  Time is a logical
    clock: (tick, node id) reproduces the order of the reference's time.monotonic() stamps."""

    def __init__(self, page_size=1, policy="lru"):
        self.page_size, self.policy = page_size, policy
        self._ids = 0
        self._clock = 0
        self.reset()

    def _now(self):
        self._clock += 1
        return self._clock

    def _new(self, priority=0):
        n = _RNode(self._ids, self._now(), priority)
        self._ids += 1
        return n

    def reset(self):
        self.root = self._new(-(1 << 62))
        self.root.lock_ref = 1
        self.evictable_size_ = 0
        self.protected_size_ = 0
        self.leaves = set()

    def _ck(self, extra, toks):
        return (extra, tuple(toks[: self.page_size]))

    def _match(self, a, b):
        n = min(len(a), len(b))
        i = 0
        while i < n and a[i] == b[i]:
            i += 1
        return i // self.page_size * self.page_size

    def _leaf(self, n):
        if n.lock_ref > 0 or n.children:
            self.leaves.discard(n)
        else:
            self.leaves.add(n)

    def _split(self, child, k):
        nn = self._new(child.priority)
        nn.hit_count, nn.extra, nn.parent, nn.lock_ref = child.hit_count, child.extra, child.parent, child.lock_ref
        old = self._ck(child.extra, child.key)
        nn.key, nn.value = child.key[:k], child.value[:k]
        child.key, child.value = child.key[k:], child.value[k:]
        nn.children[self._ck(child.extra, child.key)] = child
        child.parent = nn
        nn.parent.children[old] = nn
        return nn

    def match_prefix(self, tokens, extra=None):
        tokens = list(tokens)[: len(tokens) // self.page_size * self.page_size]
        node = self.root
        if not tokens:
            return [], node
        now = self._now()
        node.last_access = now
        out = []
        while tokens:
            child = node.children.get(self._ck(extra, tokens))
            if child is None:
                break
            child.last_access = now
            pl = self._match(child.key, tokens)
            if pl < len(child.key):
                nn = self._split(child, pl)
                out += nn.value
                node = nn
                break
            out += child.value
            node = child
            tokens = tokens[pl:]
        return out, node

    def insert(self, tokens, values, extra=None, priority=0, chunked=False):
        n = len(tokens) // self.page_size * self.page_size
        tokens, values = list(tokens)[:n], list(values)[:n]
        node = self.root
        now = self._now()
        node.last_access = now
        node.priority = max(node.priority, priority)
        total = 0
        while tokens:
            child = node.children.get(self._ck(extra, tokens))
            if child is None:
                break
            node = child
            node.last_access = now
            pl = self._match(node.key, tokens)
            total += pl
            tokens, values = tokens[pl:], values[pl:]
            if pl < len(node.key):
                node = self._split(node, pl)
            node.priority = max(node.priority, priority)
            if not chunked:
                node.hit_count += 1
        if tokens:
            nn = self._new(priority)
            nn.parent, nn.extra, nn.key, nn.value = node, extra, tokens, values
            if not chunked:
                nn.hit_count += 1
            node.children[self._ck(extra, tokens)] = nn
            self.evictable_size_ += len(tokens)
            self._leaf(node)
            self._leaf(nn)
            node = nn
        return total, node

    def inc_lock_ref(self, node):
        delta = 0
        while node is not self.root:
            if node.lock_ref == 0:
                self.evictable_size_ -= len(node.key)
                self.protected_size_ += len(node.key)
                delta -= len(node.key)
            node.lock_ref += 1
            self._leaf(node)
            node = node.parent
        return delta

    def dec_lock_ref(self, node):
        delta = 0
        while node is not self.root:
            if node.lock_ref == 1:
                self.evictable_size_ += len(node.key)
                self.protected_size_ -= len(node.key)
                delta += len(node.key)
            node.lock_ref -= 1
            self._leaf(node)
            node = node.parent
        return delta

    def _prio(self, n):
        p = self.policy
        if p == "lfu":
            return (n.hit_count, n.last_access, n.id)
        if p == "fifo":
            return (n.creation, 0, n.id)
        if p == "mru":
            return (-n.last_access, 0, n.id)
        if p == "filo":
            return (-n.creation, 0, n.id)
        if p == "priority":
            return (n.priority, n.last_access, n.id)
        if p == "slru":
            return (1 if n.hit_count >= 2 else 0, n.last_access, n.id)
        return (n.last_access, 0, n.id)

    def evict(self, num_tokens):
        import heapq

        heap = [(self._prio(n), n.id, n) for n in self.leaves]
        heapq.heapify(heap)
        segments, evicted = [], 0
        while evicted < num_tokens and heap:
            _, _, x = heapq.heappop(heap)
            segments.append(list(x.value))
            evicted += len(x.value)
            par = x.parent
            del par.children[self._ck(x.extra, x.key)]
            self.evictable_size_ -= len(x.key)
            self.leaves.discard(x)
            self._leaf(par)
            if not par.children and par.lock_ref == 0:
                heapq.heappush(heap, (self._prio(par), par.id, par))
        return evicted, segments

    def total_size(self):
        total, stack = 0, [self.root]
        while stack:
            n = stack.pop()
            total += len(n.value)
            stack.extend(n.children.values())
        return total
