"""Request->token table and the physical KV pool, with the contract of the reference's
ReqToTokenPool (srt/mem_cache/memory_pool.py:256-326) and MHATokenToKVPool (:1740-2842):
per layer ``k_buffer[l]``, ``v_buffer[l]`` of shape [size + page_size, Hkv, D] (NHD, :2030-2041)
or [pages, Hkv, page, D] (HND, :2032-2036); slot 0 / page 0 is the padding sink.

HBM layout for MI355X: one allocation per (layer, K|V) so a layer's K rows are a contiguous
2 KiB * slots span (Llama-3-8B TP1); 288 GB holds 32 layers x (256 x 4096 + 16) slots = 128 GiB
resident at the metric shape.  Writes go through rx_store_kv, moves through rx_move_kv.
"""
from __future__ import annotations

import collections
from dataclasses import dataclass
from typing import List, Optional

import torch

from .. import ops


@dataclass
class KVWriteLoc:
    """memory_pool.py:1531-1564; only ``loc`` is meaningful for the plain MHA pool."""

    loc: torch.Tensor
    swa_loc: Optional[torch.Tensor] = None
    full_loc: Optional[torch.Tensor] = None


def unwrap_write_loc(loc_info):
    if isinstance(loc_info, KVWriteLoc):
        return loc_info.loc, loc_info.swa_loc, loc_info.full_loc
    return loc_info, None, None


class ReqToTokenPool:
    """req_to_token int32[size + 1, max_context_len]; row 0 is the padding row that
    graph-padded batches read (memory_pool.py:273-281)."""

    def __init__(self, size: int, max_context_len: int, device: str):
        self.size, self.max_context_len, self.device = size, max_context_len, device
        self._alloc_size = size + 1  # + the padding row
        self.req_to_token = torch.zeros((self._alloc_size, max_context_len), dtype=torch.int32, device=device)
        self.clear()

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def available_size(self):
        return len(self.free_slots)

    def alloc(self, need_size: int = 1) -> Optional[List[int]]:
        """Hands out ``need_size`` row ids FIFO (the reference takes Req objects and stamps
        req.req_pool_idx; the id order is the same, memory_pool.py:306-314)."""
        if need_size > len(self.free_slots):
            return None
        return [self.free_slots.popleft() for _ in range(need_size)]

    def free(self, free_index):
        if isinstance(free_index, int):
            free_index = (free_index,)
        self.free_slots.extend(free_index)

    def clear(self):
        self.free_slots = collections.deque(range(1, self._alloc_size))


class MHATokenToKVPool:
    """Physical K/V storage for multi-head / grouped-query attention."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int,
                 layer_num: int, device: str, v_head_dim: Optional[int] = None,
                 start_layer: int = 0, use_hnd: bool = False):
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        # fp8 pools are stored as uint8 and viewed as fp8 by the getters (memory_pool.py:1753-1760)
        self.is_fp8 = dtype in ops.FP8_DTYPES
        if not self.is_fp8 and dtype not in (torch.bfloat16, torch.float16):
            raise NotImplementedError(f"MHATokenToKVPool: dtype {dtype} (bf16 / fp16 / float8_e4m3fn)")
        self.store_dtype = torch.uint8 if self.is_fp8 else dtype
        self.head_num = head_num
        self.head_dim = head_dim
        self.v_head_dim = head_dim if v_head_dim is None else v_head_dim
        self.layer_num = layer_num
        self.device = device
        self.start_layer = start_layer
        self.use_hnd = use_hnd
        self.row_dim = head_num * head_dim
        self.v_row_dim = head_num * self.v_head_dim
        self.num_pages = (size + page_size) // page_size
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=device)
        self._create_buffers()
        self._build_ptr_tables()
        self._store_launchers = [None] * layer_num

    def _kv_buffer_shapes(self):
        if self.use_hnd:
            return ((self.num_pages, self.head_num, self.page_size, self.head_dim),
                    (self.num_pages, self.head_num, self.page_size, self.v_head_dim))
        rows = self.size + self.page_size
        return ((rows, self.head_num, self.head_dim), (rows, self.head_num, self.v_head_dim))

    def _create_buffers(self):
        k_shape, v_shape = self._kv_buffer_shapes()
        self.k_buffer = [torch.zeros(k_shape, dtype=self.store_dtype, device=self.device)
                         for _ in range(self.layer_num)]
        self.v_buffer = [torch.zeros(v_shape, dtype=self.store_dtype, device=self.device)
                         for _ in range(self.layer_num)]

    def _build_ptr_tables(self):
        """data_ptrs / data_strides tables of memory_pool.py:2005-2028 (k0..kL-1, v0..vL-1)."""
        bufs = self.k_buffer + self.v_buffer
        self.data_ptrs = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64,
                                      device=self.device)
        self.data_strides = torch.tensor(
            [b.stride(0) * b.element_size() for b in bufs], dtype=torch.int64, device=self.device)
        if self.use_hnd:  # [pages, Hkv, page, D]: {page, head, token} strides and one head's piece, in bytes
            self.data_geom = torch.tensor(
                [[b.stride(0) * b.element_size(), b.stride(1) * b.element_size(), b.stride(2) * b.element_size(),
                  b.shape[3] * b.element_size()] for b in bufs], dtype=torch.int64, device=self.device)

    def get_kv_size_bytes(self):
        k = sum(b.numel() * b.element_size() for b in self.k_buffer)
        v = sum(b.numel() * b.element_size() for b in self.v_buffer)
        return k, v

    def get_key_buffer(self, layer_id: int):
        b = self.k_buffer[layer_id - self.start_layer]
        return b.view(self.dtype) if self.is_fp8 else b

    def get_value_buffer(self, layer_id: int):
        b = self.v_buffer[layer_id - self.start_layer]
        return b.view(self.dtype) if self.is_fp8 else b

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def get_v_head_dim(self):
        return self.v_head_dim

    def get_contiguous_buf_infos(self):
        bufs = self.k_buffer + self.v_buffer
        return ([b.data_ptr() for b in bufs], [b.nbytes for b in bufs],
                [b[0].nbytes * (1 if not self.use_hnd else 1) for b in bufs])

    def set_kv_buffer(self, layer, loc_info, cache_k: torch.Tensor, cache_v: torch.Tensor,
                      k_scale: Optional[float] = None, v_scale: Optional[float] = None,
                      layer_id_override: Optional[int] = None, dcp_kv_mask: Optional[torch.Tensor] = None):
        """memory_pool.py:2305-2381 -> _store_kv_layer (:2383-2430) -> rx_store_kv.  ``dcp_kv_mask`` (decode context
        parallel, :2351-2369 masked_set_kv_buffer_kernel): rows whose mask is 0 are not written -- they take the
        reserved slot 0, which the store kernels skip."""
        loc, _, _ = unwrap_write_loc(loc_info)
        if dcp_kv_mask is not None:
            loc = torch.where(dcp_kv_mask.to(torch.bool), loc, torch.zeros_like(loc))
        layer_id = layer_id_override if layer_id_override is not None else layer.layer_id
        li = layer_id - self.start_layer
        if self.is_fp8:
            # quant-on-write in ONE kernel (the reference: in-place div_ + .to(fp8) + store,
            # memory_pool.py:2334-2343): t = x / scale rounded to x's dtype, then RNE to e4m3fn
            n = loc.shape[0]
            if not (cache_k.is_cuda and cache_v.is_cuda and loc.is_cuda):
                raise RuntimeError("set_kv_buffer: k, v and loc must be GPU tensors (there is no CPU fallback)")
            lay = self._fp8_layouts[li] if hasattr(self, "_fp8_layouts") else None
            if lay is None:
                if not hasattr(self, "_fp8_layouts"):
                    self._fp8_layouts = [None] * self.layer_num
                kb, vb = self.k_buffer[li], self.v_buffer[li]
                lay = ops.kv_layout_hnd(kb, vb) if self.use_hnd else ops._kv_layout(kb, vb, self.page_size)
                self._fp8_layouts[li] = lay
            ops.store_cache_fp8(cache_k.reshape(n, self.row_dim), cache_v.reshape(n, self.v_row_dim), lay,
                                loc, self.head_num, self.head_dim, self.v_head_dim,
                                size_limit=(self.num_pages * self.page_size if self.use_hnd
                                            else self.size + self.page_size),
                                k_scale=1.0 if k_scale is None else float(k_scale),
                                v_scale=1.0 if v_scale is None else float(v_scale),
                                err_flag=self.err_flag)
            return
        if cache_k.dtype != self.dtype:
            cache_k = cache_k.to(self.dtype)
            cache_v = cache_v.to(self.dtype)
        n = loc.shape[0]
        k2 = cache_k.reshape(n, self.row_dim) if cache_k.dim() == 3 else cache_k
        v2 = cache_v.reshape(n, self.v_row_dim) if cache_v.dim() == 3 else cache_v
        if not (k2.is_cuda and v2.is_cuda and loc.is_cuda):
            raise RuntimeError("set_kv_buffer: k, v and loc must be GPU tensors (there is no CPU fallback)")
        if loc.dim() != 1 or loc.dtype not in (torch.int32, torch.int64):
            raise TypeError("set_kv_buffer: loc must be a 1-D int32/int64 tensor")
        if k2.stride(1) != 1:
            k2 = k2.contiguous()
        if v2.stride(1) != 1:
            v2 = v2.contiguous()
        if not loc.is_contiguous():
            loc = loc.contiguous()
        stream = torch.cuda.current_stream(k2.device).cuda_stream
        if self.use_hnd:
            # a slot is [page, :, off, :] (memory_pool.py:2372-2379): HIP scatter by (page, off, head)
            launcher = self._store_launchers[li]
            if launcher is None:
                lay = ops.kv_layout_hnd(self.k_buffer[li], self.v_buffer[li])
                launcher = self._store_launchers[li] = ops.StoreLayoutLauncher(
                    lay, self.head_num, self.head_dim, self.v_head_dim,
                    self.num_pages * self.page_size, self.err_flag)
            launcher(k2, v2, loc, stream)
            return
        launcher = self._store_launchers[li]
        if launcher is None:
            launcher = self._store_launchers[li] = ops.StoreLauncher(
                self.k_buffer[li].view(-1, self.row_dim), self.v_buffer[li].view(-1, self.v_row_dim),
                self.size + self.page_size, self.err_flag)
        launcher(k2, v2, loc, stream)

    def move_kv_cache(self, tgt_loc: torch.Tensor, src_loc: torch.Tensor):
        """memory_pool.py:2775-2842: every layer's K and V rows src -> tgt in one launch."""
        if tgt_loc.numel() == 0:
            return
        if self.use_hnd:
            ops.move_kv_layout(self.data_ptrs, self.data_geom, self.page_size, self.head_num,
                               tgt_loc.to(torch.int64), src_loc.to(torch.int64))
            return
        ops.move_kv(self.data_ptrs, self.data_strides, tgt_loc.to(torch.int64),
                    src_loc.to(torch.int64))

    def check_errors(self) -> int:
        """Host poll of the device error word (replaces the reference's device assert)."""
        v = int(self.err_flag.item())
        if v:
            self.err_flag.zero_()
        return v


class MLATokenToKVPool:
    """Latent-KV pool for MLA models (memory_pool.py:3906-4179): ONE buffer per layer,
    [size + page_size, 1, kv_lora_rank + qk_rope_head_dim]; the value view is the first
    kv_lora_rank columns of the same rows (:4006-4014), so decode reads each row once for both
    products.  Rows are bf16 / fp16, or fp8 e4m3fn bytes (576-B rows: 16-bit -> fp8 cast fused into the
    write, exact upcast inside the MLA decode kernel's staging)."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, kv_lora_rank: int,
                 qk_rope_head_dim: int, layer_num: int, device: str, start_layer: int = 0):
        self.is_fp8 = dtype in ops.FP8_DTYPES
        if not self.is_fp8 and dtype not in (torch.bfloat16, torch.float16):
            raise NotImplementedError(f"MLATokenToKVPool: dtype {dtype} (bf16 / fp16 / float8_e4m3fn)")
        self.size, self.page_size, self.dtype = size, page_size, dtype
        self.store_dtype = torch.uint8 if self.is_fp8 else dtype
        self.kv_lora_rank, self.qk_rope_head_dim = kv_lora_rank, qk_rope_head_dim
        self.kv_cache_dim = kv_lora_rank + qk_rope_head_dim
        self.layer_num, self.device, self.start_layer = layer_num, device, start_layer
        self.use_hnd = False
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=device)
        # slot 0 absorbs the writes of padded tokens (:3967)
        self.kv_buffer = [torch.zeros((size + page_size, 1, self.kv_cache_dim), dtype=self.store_dtype,
                                      device=device) for _ in range(layer_num)]
        self._fp8_layouts = [None] * layer_num
        self.data_ptrs = torch.tensor([b.data_ptr() for b in self.kv_buffer], dtype=torch.int64, device=device)
        self.data_strides = torch.tensor([b.stride(0) * b.element_size() for b in self.kv_buffer],
                                         dtype=torch.int64, device=device)

    def get_kv_size_bytes(self):
        return sum(b.numel() * b.element_size() for b in self.kv_buffer)

    def get_key_buffer(self, layer_id: int):
        b = self.kv_buffer[layer_id - self.start_layer]
        return b.view(self.dtype) if self.is_fp8 else b

    def get_value_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id)[..., : self.kv_lora_rank]

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def get_v_head_dim(self):
        return self.kv_lora_rank

    def _write_two(self, layer_id, loc, a, b):
        """dst[loc, :a_cols] = a ; dst[loc, a_cols:] = b  in one rx_store_kv launch."""
        li = layer_id - self.start_layer
        buf = self.kv_buffer[li].view(-1, self.kv_cache_dim)
        n = loc.shape[0]
        a2, b2 = a.reshape(n, -1), b.reshape(n, -1)
        if self.is_fp8:  # fused 16-bit -> fp8 cast + paged write (set_mla_kv_buffer_triton_fp8_quant, :4046-4056)
            lay = self._fp8_layouts[li]
            if lay is None:
                rows = self.kv_buffer[li]  # [slots, 1, 576] uint8
                lay = self._fp8_layouts[li] = ops._kv_layout(rows[..., : a2.shape[1]], rows[..., a2.shape[1]:], 1)
            ops.store_cache_fp8(a2, b2, lay, loc, 1, a2.shape[1], b2.shape[1],
                                size_limit=self.size + self.page_size, reserved_skip_index=-1,
                                err_flag=self.err_flag)
            return
        ops.store_cache(a2, b2, buf[:, : a2.shape[1]], buf[:, a2.shape[1]:], loc,
                        size_limit=self.size + self.page_size, reserved_skip_index=-1,
                        err_flag=self.err_flag)

    def set_kv_buffer(self, layer, loc_info, cache_k: torch.Tensor, cache_v: torch.Tensor = None, *_, **__):
        """memory_pool.py:4022-4044: the whole latent row comes in as cache_k; cache_v is unused."""
        loc, _, _ = unwrap_write_loc(loc_info)
        if not self.is_fp8 and cache_k.dtype != self.dtype:
            cache_k = cache_k.to(self.dtype)
        k2 = cache_k.reshape(loc.shape[0], self.kv_cache_dim)
        self._write_two(layer.layer_id, loc, k2[:, : self.kv_lora_rank], k2[:, self.kv_lora_rank:])

    def set_mla_kv_buffer(self, layer, loc: torch.Tensor, cache_k_nope: torch.Tensor,
                          cache_k_rope: torch.Tensor):
        """memory_pool.py:4095-4115 (two-tensor write of [nope | rope])."""
        if not self.is_fp8 and cache_k_nope.dtype != self.dtype:
            cache_k_nope, cache_k_rope = cache_k_nope.to(self.dtype), cache_k_rope.to(self.dtype)
        self._write_two(layer.layer_id, loc, cache_k_nope, cache_k_rope)

    def get_mla_kv_buffer(self, layer, loc: torch.Tensor, dst_dtype: Optional[torch.dtype] = None):
        """memory_pool.py:4117-4138."""
        dst_dtype = dst_dtype or (torch.bfloat16 if self.is_fp8 else self.dtype)
        return ops.get_mla_kv(self.kv_buffer[layer.layer_id - self.start_layer], loc, self.kv_lora_rank,
                              self.qk_rope_head_dim, dst_dtype, size_limit=self.size + self.page_size,
                              err_flag=self.err_flag)

    def move_kv_cache(self, tgt_loc: torch.Tensor, src_loc: torch.Tensor):
        if tgt_loc.numel():
            ops.move_kv(self.data_ptrs, self.data_strides, tgt_loc.to(torch.int64), src_loc.to(torch.int64))

    def check_errors(self) -> int:
        v = int(self.err_flag.item())
        if v:
            self.err_flag.zero_()
        return v
