"""PyTorch custom-op binding of the C ABI: ``torch.ops.radix_hip.*``.

The reference registers its kernels with ``register_custom_op(mutates_args=[...])``
(srt/utils/custom_op.py:57-…; e.g. ``store_cache`` kernels/ops/kvcache/kvcache.py:57 and
``unified_attention_with_output`` srt/layers/radix_attention.py:399-436) so that Dynamo and graph
capture treat them as opaque, in-place operators.  This module does the same for libradix_hip.so:
every op mutates caller-owned tensors, returns nothing, and has a fake (meta) implementation that
does nothing -- shapes never change, so tracing needs no more.

The real implementations call ``sglang_amd.ops`` (ctypes -> C ABI); there is no CPU kernel behind
them: called on CPU tensors they raise, exactly like ``ops``.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops

NS = "radix_hip"


def _op(name, mutates):
    return torch.library.custom_op(f"{NS}::{name}", mutates_args=mutates)


# K1 ---------------------------------------------------------------------------------------------
@_op("store_cache", ("k_cache", "v_cache"))
def store_cache(k: Tensor, v: Tensor, k_cache: Tensor, v_cache: Tensor, indices: Tensor,
                size_limit: int = 0, reserved_skip_index: int = 0) -> None:
    ops.store_cache(k, v, k_cache, v_cache, indices, size_limit=size_limit,
                    reserved_skip_index=reserved_skip_index)


# K2 / K3 ----------------------------------------------------------------------------------------
@_op("build_kv_indices", ("kv_indptr", "kv_indices"))
def build_kv_indices(req_to_token: Tensor, req_pool_indices: Tensor, lens: Tensor, kv_indptr: Tensor,
                     kv_indices: Tensor, kv_start: Optional[Tensor] = None) -> None:
    ops.build_kv_indices(req_to_token, req_pool_indices, lens, kv_indptr, kv_indices, kv_start)


@_op("get_num_kv_splits", ("num_kv_splits",))
def get_num_kv_splits(num_kv_splits: Tensor, seq_lens: Tensor, num_head: int, num_kv_head: int,
                      max_kv_splits: int, device_core_count: int) -> None:
    ops.get_num_kv_splits(num_kv_splits, seq_lens, num_head, num_kv_head, max_kv_splits, device_core_count)


# K4-K6 ------------------------------------------------------------------------------------------
@_op("decode_attention", ("o", "attn_logits", "attn_lse"))
def decode_attention(q: Tensor, k_buffer: Tensor, v_buffer: Tensor, o: Tensor, kv_indptr: Tensor,
                     kv_indices: Tensor, attn_logits: Tensor, attn_lse: Tensor,
                     num_kv_splits: Optional[Tensor], max_kv_splits: int, sm_scale: float,
                     k_scale: float = 1.0, v_scale: float = 1.0, logit_cap: float = 0.0,
                     sinks: Optional[Tensor] = None, page_size: int = 1) -> None:
    ops.decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse,
                             num_kv_splits, max_kv_splits, sm_scale, k_scale, v_scale, logit_cap,
                             sinks=sinks, page_size=page_size)


@_op("decode_attention_paged", ("o", "attn_logits", "attn_lse"))
def decode_attention_paged(q: Tensor, k_buffer: Tensor, v_buffer: Tensor, o: Tensor, req_to_token: Tensor,
                           req_pool_indices: Tensor, seq_lens: Tensor, attn_logits: Tensor,
                           attn_lse: Tensor, num_kv_splits: Optional[Tensor], max_kv_splits: int,
                           sm_scale: float, k_scale: float = 1.0, v_scale: float = 1.0,
                           logit_cap: float = 0.0, sinks: Optional[Tensor] = None,
                           page_size: int = 1) -> None:
    ops.decode_attention_fwd_paged(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens,
                                   attn_logits, attn_lse, num_kv_splits, max_kv_splits, sm_scale,
                                   k_scale, v_scale, logit_cap, sinks, page_size)


# K7 ---------------------------------------------------------------------------------------------
# (torch.library cannot take an OPTIONAL mutated argument that the caller omits, so the LSE output
# has its own op)
def _extend(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
            is_causal, max_len_extend, k_scale, v_scale, sm_scale, logit_cap, sliding_window_size, sinks,
            lse_extend, skip_prefix, skip_extend, page_size, custom_mask, mask_indptr, skip_prefix_custom_mask,
            window_kv_offsets, xai_temperature_len):
    ops.extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
                             kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr, max_len_extend,
                             k_scale, v_scale, sm_scale=sm_scale, logit_cap=logit_cap,
                             skip_prefix_custom_mask=skip_prefix_custom_mask,
                             sliding_window_size=sliding_window_size, sinks=sinks,
                             window_kv_offsets=window_kv_offsets, xai_temperature_len=xai_temperature_len,
                             lse_extend=lse_extend, skip_prefix=skip_prefix, skip_extend=skip_extend,
                             page_size=page_size)


@_op("extend_attention", ("o_extend",))
def extend_attention(q_extend: Tensor, k_extend: Tensor, v_extend: Tensor, o_extend: Tensor,
                     k_buffer: Optional[Tensor], v_buffer: Optional[Tensor], qo_indptr: Tensor,
                     kv_indptr: Tensor, kv_indices: Optional[Tensor], is_causal: bool,
                     max_len_extend: int, k_scale: float = 1.0, v_scale: float = 1.0,
                     sm_scale: Optional[float] = None, logit_cap: float = 0.0,
                     sliding_window_size: int = -1, sinks: Optional[Tensor] = None,
                     skip_prefix: bool = False, skip_extend: bool = False, page_size: int = 1,
                     custom_mask: Optional[Tensor] = None, mask_indptr: Optional[Tensor] = None,
                     skip_prefix_custom_mask: bool = True, window_kv_offsets: Optional[Tensor] = None,
                     xai_temperature_len: int = -1) -> None:
    _extend(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
            is_causal, max_len_extend, k_scale, v_scale, sm_scale, logit_cap, sliding_window_size, sinks, None,
            skip_prefix, skip_extend, page_size, custom_mask, mask_indptr, skip_prefix_custom_mask,
            window_kv_offsets, xai_temperature_len)


@_op("extend_attention_lse", ("o_extend", "lse_extend"))
def extend_attention_lse(q_extend: Tensor, k_extend: Tensor, v_extend: Tensor, o_extend: Tensor,
                         lse_extend: Tensor, k_buffer: Optional[Tensor], v_buffer: Optional[Tensor],
                         qo_indptr: Tensor, kv_indptr: Tensor, kv_indices: Optional[Tensor], is_causal: bool,
                         max_len_extend: int, k_scale: float = 1.0, v_scale: float = 1.0,
                         sm_scale: Optional[float] = None, logit_cap: float = 0.0,
                         sliding_window_size: int = -1, sinks: Optional[Tensor] = None,
                         skip_prefix: bool = False, skip_extend: bool = False, page_size: int = 1) -> None:
    _extend(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
            is_causal, max_len_extend, k_scale, v_scale, sm_scale, logit_cap, sliding_window_size, sinks,
            lse_extend, skip_prefix, skip_extend, page_size, None, None, True, None, -1)


# K9 / K10 / K11 ---------------------------------------------------------------------------------
@_op("alloc_extend", ("free_pages", "out_indices"))
def alloc_extend(prefix_lens: Tensor, seq_lens: Tensor, last_loc: Tensor, free_pages: Tensor,
                 out_indices: Tensor, page_size: int) -> None:
    ops.alloc_extend(prefix_lens, seq_lens, last_loc, free_pages, out_indices, page_size)


@_op("alloc_decode", ("free_pages", "out_indices"))
def alloc_decode(seq_lens: Tensor, last_loc: Tensor, free_pages: Tensor, out_indices: Tensor,
                 page_size: int) -> None:
    ops.alloc_decode(seq_lens, last_loc, free_pages, out_indices, page_size)


@_op("write_req_to_token", ("req_to_token",))
def write_req_to_token(req_to_token: Tensor, req_pool_indices: Tensor, prefix_ptrs: Tensor,
                       pre_lens: Tensor, seq_lens: Tensor, extend_lens: Tensor,
                       out_cache_loc: Tensor) -> None:
    ops.write_req_to_token(req_to_token, req_pool_indices, prefix_ptrs, pre_lens, seq_lens, extend_lens,
                           out_cache_loc)


# The moved buffers are reached through the pointer table, not through tensor arguments: like the
# reference's copy_all_layer_kv_cache_tiled (kernels/ops/kvcache/cache_move.py:60-133) the op declares
# no mutated argument, so keep it out of functionalised graphs (eager / HIP-graph capture only).
@_op("move_kv", ())
def move_kv(data_ptrs: Tensor, row_bytes: Tensor, tgt_loc: Tensor, src_loc: Tensor) -> None:
    ops.move_kv(data_ptrs, row_bytes, tgt_loc, src_loc)


# merge_state_triton (kernels/ops/attention/merge_state.py:66-96) as an out-variant op
@_op("merge_state", ("output", "output_lse"))
def merge_state(prefix_output: Tensor, prefix_lse: Tensor, suffix_output: Tensor, suffix_lse: Tensor,
                output: Tensor, output_lse: Tensor) -> None:
    ops.merge_state(prefix_output, prefix_lse, suffix_output, suffix_lse, output, output_lse)


# shared-prefix decode plan (include/radix_hip.h: rx_shared_prefix_plan); all outputs int32
@_op("shared_prefix_plan", ("plan", "chunk_indptr", "shared_indices", "kv_start", "suffix_lens"))
def shared_prefix_plan(req_to_token: Tensor, req_pool_indices: Tensor, seq_lens: Tensor, plan: Tensor,
                       chunk_indptr: Tensor, shared_indices: Tensor, kv_start: Tensor, suffix_lens: Tensor,
                       min_shared: int = 0, chunk_align: int = 64) -> None:
    ops.shared_prefix_plan(req_to_token, req_pool_indices, seq_lens, plan, chunk_indptr, shared_indices,
                           kv_start, suffix_lens, min_shared=min_shared, chunk_align=chunk_align)


# fused_qk_norm_rope_out (kernels/ops/attention/fused_qknorm_rope.py:33-100: op_name, argument order, mutates qkv)
@_op("fused_qk_norm_rope_out", ("qkv",))
def fused_qk_norm_rope_out(qkv: Tensor, q_weight: Tensor, k_weight: Tensor, position_ids: Tensor, num_heads_q: int,
                           num_heads_k: int, num_heads_v: int, head_dim: int, eps: float, base: float, is_neox: bool,
                           factor: float, low: float, high: float, attention_factor: float, rotary_dim: int) -> None:
    ops.fused_qk_norm_rope(qkv, num_heads_q, num_heads_k, num_heads_v, head_dim, eps, q_weight, k_weight, base, is_neox,
                           position_ids, factor, low, high, attention_factor, rotary_dim)


# fused_fp8_qkv_kv_cache (kernels/ops/kvcache/fused_fp8_qkv_kv_cache.py:35-80) as out-variant ops: the reference's function
# allocates and returns q's fp8 copy; behind torch.library the caller owns it (an optional mutated argument cannot be omitted,
# so the form without q is its own op)
@_op("fused_fp8_qkv_kv_cache_out", ("q_out", "k_cache", "v_cache"))
def fused_fp8_qkv_kv_cache_out(q: Tensor, k: Tensor, v: Tensor, q_out: Tensor, k_cache: Tensor, v_cache: Tensor,
                               cache_loc: Tensor, k_scale: Optional[Tensor] = None, v_scale: Optional[Tensor] = None) -> None:
    ops.fused_fp8_qkv_kv_cache(q, k, v, k_cache, v_cache, cache_loc, k_scale, v_scale, q_out=q_out)


@_op("fused_fp8_kv_cache", ("k_cache", "v_cache"))
def fused_fp8_kv_cache(k: Tensor, v: Tensor, k_cache: Tensor, v_cache: Tensor, cache_loc: Tensor,
                       k_scale: Optional[Tensor] = None, v_scale: Optional[Tensor] = None) -> None:
    ops.fused_fp8_qkv_kv_cache(None, k, v, k_cache, v_cache, cache_loc, k_scale, v_scale)


ALL_OPS = (fused_fp8_qkv_kv_cache_out, fused_fp8_kv_cache, fused_qk_norm_rope_out, merge_state, shared_prefix_plan, store_cache, build_kv_indices, get_num_kv_splits, decode_attention, decode_attention_paged,
           extend_attention, extend_attention_lse, alloc_extend, alloc_decode, write_req_to_token, move_kv)

for _o in ALL_OPS:  # in-place ops: the fake implementation has nothing to compute
    _o.register_fake(lambda *a, **k: None)
