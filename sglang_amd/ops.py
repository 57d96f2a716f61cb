"""Operator-level host API: same names, argument meaning and error behaviour as the
reference's Python operators for this path, implemented by calls into libradix_hip.so.

torch is used for device memory and streams only.  Each wrapper cites the reference
operator it replaces (paths relative to /root/reference/python/sglang/).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import lib as _L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _is64(t: torch.Tensor, name: str) -> int:
    if t.dtype == torch.int64:
        return 1
    if t.dtype == torch.int32:
        return 0
    raise TypeError(f"{name} must be int32 or int64, got {t.dtype}")


def _rx_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _L.RX_BF16
    if t.dtype == torch.float16:
        return _L.RX_F16
    raise TypeError(f"unsupported dtype {t.dtype}: the HIP path computes in bf16/fp16")


FP8_DTYPES = (torch.float8_e4m3fn,)  # gfx950 pools are OCP e4m3fn (MI300's fnuz encoding is not supported)


def _is_fp8_pool(buf: torch.Tensor) -> bool:
    """fp8 pools arrive as float8_e4m3fn views of the uint8 store (KVCache.get_key_buffer,
    srt/mem_cache/memory_pool.py:2273-2290); a raw uint8 store buffer is accepted too."""
    return buf.dtype in FP8_DTYPES or buf.dtype == torch.uint8


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "sglang_amd ops run on the GPU through libradix_hip.so only; got a "
                f"{t.device} tensor (there is no CPU fallback)")


# --------------------------------------------------------------------------------------
# K1  store_cache            kernels/ops/kvcache/kvcache.py:57-110
# --------------------------------------------------------------------------------------
def store_cache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                indices: torch.Tensor, *, row_bytes: int = 0, v_row_bytes: int = 0,
                size_limit: int = 0, reserved_skip_index: int = 0,
                err_flag: Optional[torch.Tensor] = None) -> None:
    """k_cache[indices[i]] = k[i]; v_cache[indices[i]] = v[i].  k/v are (N, H*D) views whose
    rows may be strided; caches are (rows, H*D).  Writes to ``reserved_skip_index`` (slot 0)
    are skipped; an index outside [0, size_limit) is dropped and flagged in ``err_flag``."""
    _require_cuda(k, v, k_cache, v_cache, indices)
    k2 = k.reshape(k.shape[0], -1) if k.dim() != 2 else k
    v2 = v.reshape(v.shape[0], -1) if v.dim() != 2 else v
    kc = k_cache.view(k_cache.shape[0], -1) if k_cache.dim() != 2 else k_cache
    vc = v_cache.view(v_cache.shape[0], -1) if v_cache.dim() != 2 else v_cache
    if k2.stride(-1) != 1 or v2.stride(-1) != 1 or kc.stride(-1) != 1 or vc.stride(-1) != 1:
        raise ValueError("store_cache: innermost dimension must be contiguous")
    if indices.dim() != 1 or indices.shape[0] != k2.shape[0] or v2.shape[0] != k2.shape[0]:
        raise ValueError("store_cache: indices / k / v batch sizes differ")
    es = k2.element_size()
    row_bytes = row_bytes or k2.shape[-1] * es
    v_row_bytes = v_row_bytes or v2.shape[-1] * v2.element_size()
    if size_limit <= 0:
        size_limit = kc.shape[0]
    idx = indices if indices.is_contiguous() else indices.contiguous()
    st = _L.load().rx_store_kv(
        _ptr(k2), _ptr(v2), _ptr(kc), _ptr(vc), _ptr(idx), k2.shape[0], row_bytes, v_row_bytes,
        k2.stride(0) * es, v2.stride(0) * v2.element_size(), kc.stride(0) * kc.element_size(),
        vc.stride(0) * vc.element_size(), _is64(idx, "indices"), size_limit, reserved_skip_index,
        _ptr(err_flag), _stream(k))
    _L.check(st, "rx_store_kv")


def store_cache_layout(k, v, layout: "_L.RxKvLayout", indices, num_kv_heads, head_dim, v_head_dim, *,
                       size_limit: int, reserved_skip_index: int = 0, err_flag=None) -> None:
    """KV store into a pool addressed by an rx_kv_layout (HND pools): k [n, Hkv*Dk], v [n, Hkv*Dv]."""
    _require_cuda(k, v, indices)
    k2 = k.reshape(k.shape[0], -1)
    v2 = v.reshape(v.shape[0], -1)
    if k2.stride(-1) != 1 or v2.stride(-1) != 1:
        raise ValueError("store_cache_layout: innermost dimension must be contiguous")
    idx = indices if indices.is_contiguous() else indices.contiguous()
    st = _L.load().rx_store_kv_layout(_ptr(k2), _ptr(v2), C.byref(layout), _ptr(idx), k2.shape[0],
                                      num_kv_heads, head_dim, v_head_dim, k2.stride(0), v2.stride(0),
                                      _is64(idx, "indices"), size_limit, reserved_skip_index,
                                      _ptr(err_flag), _stream(k2))
    _L.check(st, "rx_store_kv_layout")


def store_cache_fp8(k, v, layout: "_L.RxKvLayout", indices, num_kv_heads, head_dim, v_head_dim, *,
                    size_limit: int, k_scale: float = 1.0, v_scale: float = 1.0,
                    reserved_skip_index: int = 0, err_flag=None) -> None:
    """Quantising KV store into an fp8 e4m3fn pool (set_kv_buffer with an fp8 store dtype,
    srt/mem_cache/memory_pool.py:2305-2381: ``cache_k.div_(k_scale)`` then ``.to(fp8)``; the MLA
    two-tensor fp8 write, :4046-4066).  k [n, Hkv*Dk], v [n, Hkv*Dv] are 16-bit."""
    _require_cuda(k, v, indices)
    k2 = k.reshape(k.shape[0], -1)
    v2 = v.reshape(v.shape[0], -1)
    if k2.stride(-1) != 1 or v2.stride(-1) != 1:
        raise ValueError("store_cache_fp8: innermost dimension must be contiguous")
    if k2.dtype != v2.dtype:
        raise TypeError("store_cache_fp8: k and v dtypes differ")
    idx = indices if indices.is_contiguous() else indices.contiguous()
    st = _L.load().rx_store_kv_fp8(_ptr(k2), _ptr(v2), C.byref(layout), _ptr(idx), k2.shape[0],
                                   num_kv_heads, head_dim, v_head_dim, k2.stride(0), v2.stride(0),
                                   _rx_dtype(k2), float(k_scale), float(v_scale), _is64(idx, "indices"),
                                   size_limit, reserved_skip_index, _ptr(err_flag), _stream(k2))
    _L.check(st, "rx_store_kv_fp8")


def fused_fp8_qkv_kv_cache(q: Optional[torch.Tensor], k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor,
                           v_cache: torch.Tensor, cache_loc: torch.Tensor, k_scale: Optional[torch.Tensor] = None,
                           v_scale: Optional[torch.Tensor] = None, *, kv_layout=None, page_size: int = 1,
                           err_flag: Optional[torch.Tensor] = None, q_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """fused_fp8_qkv_kv_cache (kernels/ops/kvcache/fused_fp8_qkv_kv_cache.py:35-80), same name, argument order and
    return: fp8 e4m3fn quantisation of K / V into the paged cache at ``cache_loc`` and -- when ``q`` is given -- of q into
    a fresh dense fp8 tensor that is returned (``None`` otherwise).  k / v: 16-bit [n, Hkv, D] or [n, Hkv*D], rows may be
    strided (slices of a fused qkv tensor); k_cache / v_cache: fp8 (or uint8) pools ``[slots, Hkv, D]`` -- or an HND pool
    with ``kv_layout=ops.kv_layout_hnd(...)``; k_scale / v_scale: fp32 DEVICE scalars or None (= 1).  y = x * (1 / scale),
    saturated to +-448 (the reference kernel's arithmetic; rx_fused_fp8_qkv_kv_cache).  ``q_out``: a caller-owned fp8 output for q
    (the custom-op out-variant, torch.ops.radix_hip.fused_fp8_qkv_kv_cache_out); by default a fresh tensor as the reference."""
    if k.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError(f"Unsupported dtype {k.dtype}. Supported: bfloat16, float16")
    _require_cuda(q, k, v, k_cache, v_cache, cache_loc, k_scale, v_scale)
    n = k.shape[0]
    k2, v2 = k.reshape(n, -1), v.reshape(n, -1)
    if v2.dtype != k2.dtype or k2.stride(-1) != 1 or v2.stride(-1) != 1 or cache_loc.numel() != n:
        raise ValueError("fused_fp8_qkv_kv_cache: k / v must share a dtype and be contiguous in the last dim; one cache_loc per token")
    if not (_is_fp8_pool(k_cache) and _is_fp8_pool(v_cache)):
        raise TypeError("fused_fp8_qkv_kv_cache: k_cache / v_cache must be fp8 e4m3fn (or uint8) pools")
    if kv_layout is not None:
        lay, hkv, dk, dv = kv_layout, k_cache.shape[1], k_cache.shape[-1], v_cache.shape[-1]
        slots = k_cache.shape[0] * k_cache.shape[2]
    else:
        kc = k_cache if k_cache.dim() == 3 else k_cache.view(k_cache.shape[0], 1, -1)
        vc = v_cache if v_cache.dim() == 3 else v_cache.view(v_cache.shape[0], 1, -1)
        lay, hkv, dk, dv = _kv_layout(kc, vc, page_size), kc.shape[1], kc.shape[2], vc.shape[2]
        slots = kc.shape[0]
    if k2.shape[1] != hkv * dk or v2.shape[1] != hkv * dv:
        raise ValueError("fused_fp8_qkv_kv_cache: k / v row width does not match the cache's [Hkv, D]")

    def scale_ptr(sc):
        if sc is None:
            return None, None
        if sc.dtype != torch.float32 or sc.numel() != 1:
            sc = sc.to(torch.float32).reshape(1)  # (the reference's _scale_to_f32)
        return sc, sc.data_ptr()

    ks, ksp = scale_ptr(k_scale)
    vs, vsp = scale_ptr(v_scale)
    q2 = None
    if q is None:
        q_out = None
    if q is not None:
        q2 = q.reshape(n, -1)
        if q2.dtype != k2.dtype or q2.stride(-1) != 1:
            raise ValueError("fused_fp8_qkv_kv_cache: q must have k's dtype and be contiguous in the last dim")
        if q_out is None:
            q_out = torch.empty(q2.shape, dtype=torch.float8_e4m3fn, device=q.device)
        elif (not _is_fp8_pool(q_out) or q_out.numel() != q2.numel() or not q_out.is_contiguous() or not q_out.is_cuda):
            raise ValueError("fused_fp8_qkv_kv_cache: q_out must be a contiguous fp8 GPU tensor of q's element count")
    if n == 0:
        return q_out
    idx = cache_loc if cache_loc.is_contiguous() else cache_loc.contiguous()
    st = _L.load().rx_fused_fp8_qkv_kv_cache(_ptr(q2), _ptr(k2), _ptr(v2), _ptr(q_out), C.byref(lay), _ptr(idx), _is64(idx, "cache_loc"),
                                             ksp, vsp, n, 0 if q2 is None else q2.shape[1], hkv, dk, dv,
                                             0 if q2 is None else q2.stride(0), k2.stride(0), v2.stride(0), _rx_dtype(k2), slots,
                                             _ptr(err_flag), _stream(k2))
    _L.check(st, "rx_fused_fp8_qkv_kv_cache")
    return q_out


def get_mla_kv(kv_buffer: torch.Tensor, loc: torch.Tensor, nope_cols: int, rope_cols: int,
               dst_dtype: torch.dtype, *, size_limit: int, err_flag=None):
    """get_mla_kv_buffer_triton (kernels/ops/kvcache/mla_buffer.py): gather latent rows into dense
    (nope [n,1,nope_cols], rope [n,1,rope_cols]) of dst_dtype; fp8 rows are upcast exactly."""
    _require_cuda(kv_buffer, loc)
    if dst_dtype not in (torch.bfloat16, torch.float16):
        raise TypeError(f"get_mla_kv: dst_dtype {dst_dtype}")
    fp8 = _is_fp8_pool(kv_buffer)
    if not fp8 and kv_buffer.dtype != dst_dtype:
        raise TypeError("get_mla_kv: a 16-bit pool is copied, not converted: dst_dtype must equal the pool dtype")
    rows = kv_buffer.view(kv_buffer.shape[0], -1)
    n = loc.shape[0]
    nope = torch.empty((n, 1, nope_cols), dtype=dst_dtype, device=kv_buffer.device)
    rope = torch.empty((n, 1, rope_cols), dtype=dst_dtype, device=kv_buffer.device)
    idx = loc if loc.is_contiguous() else loc.contiguous()
    st = _L.load().rx_get_mla_kv(_ptr(rows), rows.stride(0), int(fp8), _ptr(idx), _is64(idx, "loc"), n,
                                 nope_cols, rope_cols, _ptr(nope), _ptr(rope),
                                 _L.RX_BF16 if dst_dtype == torch.bfloat16 else _L.RX_F16,
                                 size_limit, _ptr(err_flag), _stream(kv_buffer))
    _L.check(st, "rx_get_mla_kv")
    return nope, rope


# --------------------------------------------------------------------------------------
# K2  kv-index build         kernels/ops/kvcache/kv_indices.py:8-46 (+ triton_backend.py:386-404)
# --------------------------------------------------------------------------------------
def build_kv_indices(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                     lens: torch.Tensor, kv_indptr: torch.Tensor,
                     kv_indices: Optional[torch.Tensor], kv_start: Optional[torch.Tensor] = None):
    """Fills kv_indptr[:bs+1] (int32) with the exclusive scan of ``lens`` and, when given,
    kv_indices with the ragged gather of req_to_token rows."""
    _require_cuda(req_to_token, req_pool_indices, lens, kv_indptr, kv_indices, kv_start)
    bs = lens.shape[0]
    if req_to_token.dtype != torch.int32 or kv_indptr.dtype != torch.int32:
        raise TypeError("req_to_token and kv_indptr must be int32")
    if kv_indptr.numel() < bs + 1:
        raise ValueError("kv_indptr too small")
    if kv_start is not None and kv_start.dtype != torch.int32:
        kv_start = kv_start.to(torch.int32)
    st = _L.load().rx_build_kv_indices(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices),
        _is64(req_pool_indices, "req_pool_indices"), _ptr(lens), _is64(lens, "lens"),
        _ptr(kv_start), _ptr(kv_indptr), _ptr(kv_indices),
        0 if kv_indices is None else _is64(kv_indices, "kv_indices"), bs, _stream(req_to_token))
    _L.check(st, "rx_build_kv_indices")
    return kv_indptr[: bs + 1]


def build_unified_kv_indices(prefix_kv_indptr: torch.Tensor, prefix_kv_indices: Optional[torch.Tensor],
                             extend_start_loc: torch.Tensor, extend_seq_lens: torch.Tensor,
                             extend_kv_indices: torch.Tensor, bs: int, out_indptr: Optional[torch.Tensor] = None,
                             out_indices: Optional[torch.Tensor] = None, max_tokens_per_request: int = 0):
    """build_unified_kv_indices (kernels/ops/attention/extend_attention.py:193-238): the kv list of the one-stage extend
    -- every request's prefix slots followed by its new tokens' slots.  Same arguments and returns as the reference:
    ``(unified_kv_indptr int32[bs + 1], unified_kv_indices int64[len(prefix) + len(extend)], prefix_lens int32[bs])``;
    out_indptr / out_indices let a caller keep address-stable buffers."""
    _require_cuda(prefix_kv_indptr, prefix_kv_indices, extend_start_loc, extend_seq_lens, extend_kv_indices)
    if prefix_kv_indptr.dtype != torch.int32:
        raise TypeError("prefix_kv_indptr must be int32")
    dev = prefix_kv_indptr.device
    n_pre = 0 if prefix_kv_indices is None else prefix_kv_indices.numel()
    total = n_pre + extend_kv_indices.numel()
    indptr = out_indptr if out_indptr is not None else torch.empty(bs + 1, dtype=torch.int32, device=dev)
    indices = out_indices if out_indices is not None else torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    if indptr.dtype != torch.int32 or indptr.numel() < bs + 1 or indices.dtype != torch.int64 or indices.numel() < total:
        raise ValueError("build_unified_kv_indices: out_indptr int32[bs + 1] / out_indices int64[len(prefix) + len(extend)]")
    prefix_lens = torch.empty(bs, dtype=torch.int32, device=dev)
    for name, t in (("extend_start_loc", extend_start_loc), ("extend_seq_lens", extend_seq_lens)):
        if t.numel() < bs or not t.is_contiguous():
            raise ValueError(f"build_unified_kv_indices: {name} must be a contiguous vector of at least bs entries")
    st = _L.load().rx_build_unified_kv_indices(
        _ptr(prefix_kv_indptr), _ptr(prefix_kv_indices) if n_pre else None,
        _is64(prefix_kv_indices, "prefix_kv_indices") if n_pre else 0,
        _ptr(extend_start_loc), _is64(extend_start_loc, "extend_start_loc"), _ptr(extend_seq_lens),
        _is64(extend_seq_lens, "extend_seq_lens"), _ptr(extend_kv_indices), _is64(extend_kv_indices, "extend_kv_indices"),
        int(bs), int(max_tokens_per_request), _ptr(indptr), _ptr(indices), _ptr(prefix_lens), _stream(prefix_kv_indptr))
    _L.check(st, "rx_build_unified_kv_indices")
    return indptr[: bs + 1], indices[:total] if out_indices is None else indices, prefix_lens


# --------------------------------------------------------------------------------------
# K3  get_num_kv_splits      kernels/ops/attention/metadata.py:11-60
# --------------------------------------------------------------------------------------
def get_num_kv_splits(num_kv_splits: torch.Tensor, seq_lens: torch.Tensor, num_head: int,
                      num_kv_head: int, max_kv_splits: int, device_core_count: int) -> None:
    _require_cuda(num_kv_splits, seq_lens)
    num_token, num_seq = num_kv_splits.shape[0], seq_lens.shape[0]
    num_group = num_token // num_seq
    assert num_group * num_seq == num_token, (
        f"num_seq({num_seq}), num_token({num_token}), something goes wrong!")
    if num_kv_splits.dtype != torch.int32:
        raise TypeError("num_kv_splits must be int32")
    st = _L.load().rx_num_kv_splits(_ptr(seq_lens), _is64(seq_lens, "seq_lens"), num_seq,
                                    num_group, num_head, num_kv_head, max_kv_splits,
                                    device_core_count, _ptr(num_kv_splits), _stream(seq_lens))
    _L.check(st, "rx_num_kv_splits")


def generate_draft_decode_kv_indices(req_pool_indices, req_to_token, paged_kernel_lens, kv_indices, kv_indptr, positions,
                                     topk: int, num_steps: int, page_size: int) -> None:
    """generate_draft_decode_kv_indices (kernels/ops/speculative/cache_locs.py:56-141) with the reference launch's grid
    (speculative_num_steps, num_seqs, topk) folded into the arguments: fills kv_indices [num_steps, width] (int64 or
    int32) and kv_indptr [num_steps, >= num_seqs * topk + 1] (int32, entry 0 untouched) for every step, request and
    top-k branch.  include/radix_hip.h: rx_draft_decode_kv_indices."""
    _require_cuda(req_pool_indices, req_to_token, paged_kernel_lens, kv_indices, kv_indptr, positions)
    if req_to_token.dtype != torch.int32 or kv_indptr.dtype != torch.int32:
        raise TypeError("req_to_token / kv_indptr must be int32")
    if kv_indices.dim() != 2 or kv_indptr.dim() != 2 or kv_indices.shape[0] < num_steps or kv_indptr.shape[0] < num_steps:
        raise ValueError("kv_indices / kv_indptr must be [num_steps, ...]")
    if kv_indices.stride(1) != 1 or kv_indptr.stride(1) != 1:
        raise ValueError("kv_indices / kv_indptr rows must be contiguous")
    st = _L.load().rx_draft_decode_kv_indices(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _is64(req_pool_indices, "req_pool_indices"),
        _ptr(paged_kernel_lens), _is64(paged_kernel_lens, "paged_kernel_lens"), _ptr(positions), _is64(positions, "positions"),
        paged_kernel_lens.shape[0], int(topk), int(num_steps), int(page_size), _ptr(kv_indices),
        _is64(kv_indices, "kv_indices"), kv_indices.stride(0), _ptr(kv_indptr), kv_indptr.stride(0), _stream(kv_indices))
    _L.check(st, "rx_draft_decode_kv_indices")


def native_max_kv_splits(bs: int, num_head: int, num_kv_head: int, cu_count: int, cap: int) -> int:
    """Uniform split count of the MI355X-native schedule (include/radix_hip.h, rx_num_kv_splits_native)."""
    group = max(1, num_head // num_kv_head)
    wgs = max(1, bs * num_kv_head * ((group + 15) // 16))
    return max(1, min(cap, -(-cu_count // wgs)))


def get_num_kv_splits_native(num_kv_splits: torch.Tensor, seq_lens: torch.Tensor, num_head: int, num_kv_head: int,
                             max_kv_splits: int, device_core_count: int, min_tokens_per_split: int = 128) -> None:
    _require_cuda(num_kv_splits, seq_lens)
    if num_kv_splits.dtype != torch.int32:
        raise TypeError("num_kv_splits must be int32")
    group = max(1, num_head // num_kv_head)
    st = _L.load().rx_num_kv_splits_native(_ptr(seq_lens), _is64(seq_lens, "seq_lens"), seq_lens.shape[0],
                                           num_kv_head * ((group + 15) // 16), device_core_count, max_kv_splits,
                                           min_tokens_per_split, _ptr(num_kv_splits), _stream(seq_lens))
    _L.check(st, "rx_num_kv_splits_native")


def balanced_kv_splits_host(lens, num_head: int, num_kv_head: int, max_kv_splits: int, wg_target: int,
                            min_tokens_per_split: int = 128, wg_target_mixed: int = 0) -> np.ndarray:
    """Host mirror of rx_num_kv_splits_balanced (include/radix_hip.h) on the CPU copy of the lengths: what the eager
    metadata path uses to size the split slots, the (request, split) table and the grid.  The device pass MUST run
    with the same ``max_kv_splits`` cap: under the rounds rule the cap feeds the round tests, so a smaller cap on the
    device can hand out more pairs than counted here."""
    lens = np.maximum(np.asarray(lens, dtype=np.int64), 0)
    group = max(1, num_head // num_kv_head)
    wgpr = num_kv_head * ((group + 15) // 16)
    work = int(lens.sum()) * wgpr

    def counts(tstar):
        n = np.where(2 * lens > 3 * tstar, np.minimum(max_kv_splits, -(-lens // tstar)), 1)
        return np.maximum(n, 1).astype(np.int32)

    if wg_target_mixed != 0:  # the fill rule (rx_misc.hip: near-uniform batches of ~1-3 whole-request workgroups per CU)
        live, total, mx = int((lens > 0).sum()), int(lens.sum()), int(lens.max()) if lens.size else 0
        cus, blocks = max(1, wg_target // 2), live * wgpr
        if 0 < wg_target_mixed <= wg_target:  # no live-pairs grid (MLA kernels): whole requests from 0.8 blocks per CU up
            if live > 0 and 2 * mx * live <= 3 * total and 10 * blocks >= 8 * cus:
                return np.ones(lens.shape, dtype=np.int32)
        elif live > 0 and 2 * mx * live <= 3 * total and 10 * blocks < 7 * cus:  # everybody is cut: a count that covers the chip evenly
            mt = int(min_tokens_per_split)
            mean = total // live
            ts = max(mt, -(-work // wg_target))
            n0 = max(1, min(int(max_kv_splits), -(-mean // ts)))
            smax = min(int(max_kv_splits), max(1, mean // mt))

            def fills(sp):
                w = blocks * sp
                return 2 * w >= 3 * cus and 100 * w >= 85 * -(-w // cus) * cus

            S = 0
            if not fills(n0):
                for dlt in range(1, 9):
                    if n0 - dlt >= 2 and fills(n0 - dlt):
                        S = n0 - dlt
                    elif n0 + dlt <= smax and fills(n0 + dlt):
                        S = n0 + dlt
                    if S:
                        break
            if S:
                return np.maximum(1, np.minimum(S, lens // mt)).astype(np.int32)
        elif live > 0 and 2 * mx * live <= 3 * total and blocks < 3 * cus:
            S = 1
            if blocks > cus:
                S, best, bn, bd = 0, 1, 0, 1
                for sp in range(1, min(int(max_kv_splits), 6) + 1):
                    rounds = -(-blocks * sp // cus)
                    if 100 * blocks * sp >= 85 * rounds * cus:
                        S = sp
                        break
                    if blocks * sp * bd > bn * rounds * cus:
                        best, bn, bd = sp, blocks * sp, rounds * cus
                S = S or best
            return np.maximum(1, np.minimum(S, lens // 256)).astype(np.int32)
    n = counts(max(int(min_tokens_per_split), -(-work // wg_target)))
    if wg_target_mixed < 0 and (n > 1).any() and (n == 1).any():  # the rounds rule (two workgroups per CU)
        split = n > 1
        if -(-int(n.sum()) * wgpr // wg_target) < 2:
            return n
        cu = int((~split).sum())
        a = max(1, -(-int(lens[~split].sum()) // cu))
        t1 = max(int(min_tokens_per_split), -(-work // wg_target))
        for R in (1, 2, 3, 4):
            p = max(R * a, t1)
            pieces = np.minimum(max_kv_splits, -(-lens[split] // p))
            if -(-(cu + int(pieces.sum())) * wgpr // wg_target) <= R:
                break
        n = n.copy()
        n[split] = np.maximum(1, pieces).astype(np.int32)
        return n
    if wg_target_mixed > wg_target and (n > 1).any() and (n == 1).any():  # a mixed batch: the live-pairs grid's budget
        tstar = max(int(min_tokens_per_split), -(-work // wg_target_mixed))
        wgs = int(counts(tstar).sum()) * wgpr
        if wgs > wg_target_mixed:
            tstar = -(-tstar * wgs // wg_target_mixed)
        n = counts(tstar)
    return n


def get_num_kv_splits_balanced(num_kv_splits: torch.Tensor, seq_lens: torch.Tensor, num_head: int, num_kv_head: int,
                               max_kv_splits: int, wg_target: int, min_tokens_per_split: int = 128,
                               wg_target_mixed: int = 0) -> None:
    _require_cuda(num_kv_splits, seq_lens)
    if num_kv_splits.dtype != torch.int32:
        raise TypeError("num_kv_splits must be int32")
    group = max(1, num_head // num_kv_head)
    st = _L.load().rx_num_kv_splits_balanced(_ptr(seq_lens), _is64(seq_lens, "seq_lens"), seq_lens.shape[0],
                                             num_kv_head * ((group + 15) // 16), int(wg_target), int(max_kv_splits),
                                             int(min_tokens_per_split), int(wg_target_mixed), _ptr(num_kv_splits),
                                             _stream(seq_lens))
    _L.check(st, "rx_num_kv_splits_balanced")


# --------------------------------------------------------------------------------------
# KV buffer strides          _extract_kv_strides, kernels/ops/attention/decode_attention.py:39-88
# --------------------------------------------------------------------------------------
def _kv_layout(k_buffer: torch.Tensor, v_buffer: torch.Tensor, page_size: int) -> _L.RxKvLayout:
    lay = _L.RxKvLayout()
    lay.k_buf = k_buffer.data_ptr()
    lay.v_buf = v_buffer.data_ptr()
    lay.page_size = page_size

    def strides(buf):
        if buf.dim() == 4:  # [pages, page_size, H, D]  (page-major shared view)
            assert buf.shape[1] == page_size
            return buf.stride(0), buf.stride(1), buf.stride(2)
        if buf.dim() == 3:  # [slots, H, D]
            return buf.stride(0) * page_size, buf.stride(0), buf.stride(1)
        raise ValueError(f"unexpected KV buffer ndim={buf.dim()}, shape={tuple(buf.shape)}")

    lay.k_page_stride, lay.k_tok_stride, lay.k_head_stride = strides(k_buffer)
    lay.v_page_stride, lay.v_tok_stride, lay.v_head_stride = strides(v_buffer)
    if k_buffer.stride(-1) != 1 or v_buffer.stride(-1) != 1:
        raise ValueError("KV buffers must be contiguous in head_dim")
    lay.kv_fp8 = int(_is_fp8_pool(k_buffer))
    return lay


def kv_layout_hnd(k_buffer: torch.Tensor, v_buffer: torch.Tensor) -> _L.RxKvLayout:
    """HND pool [pages, H, page, D] (memory_pool.py:2032-2036)."""
    lay = _L.RxKvLayout()
    lay.k_buf = k_buffer.data_ptr()
    lay.v_buf = v_buffer.data_ptr()
    lay.page_size = k_buffer.shape[2]
    lay.k_page_stride, lay.k_head_stride, lay.k_tok_stride = (k_buffer.stride(0), k_buffer.stride(1),
                                                              k_buffer.stride(2))
    lay.v_page_stride, lay.v_head_stride, lay.v_tok_stride = (v_buffer.stride(0), v_buffer.stride(1),
                                                              v_buffer.stride(2))
    lay.kv_fp8 = int(_is_fp8_pool(k_buffer))
    return lay


# --------------------------------------------------------------------------------------
# score_mod / aux_tensors       kernels/ops/attention/score_mod.py:30-56
# --------------------------------------------------------------------------------------
def relative_bias_score_mod(*_args, **_kwargs):
    """Marker with the name of the reference's one Triton score_mod (score_mod.py:44-56: qk + Aux0[q_idx, head,
    q_pos - kv_pos] inside [0, aux0_len)).  Pass it -- or the reference's own ``relative_bias_score_mod`` object -- as
    ``score_mod`` with ``aux_tensors = [rel_logits]``; the kernels implement it natively (rx_*_params.score_bias).  Any
    other callable cannot cross a C ABI and raises NotImplementedError."""
    raise RuntimeError("relative_bias_score_mod is a marker for the HIP kernels' built-in bias; it is never called")


def _is_relative_bias(score_mod) -> bool:
    if score_mod is relative_bias_score_mod:
        return True
    for obj in (score_mod, getattr(score_mod, "fn", None)):  # (a triton JITFunction keeps the python function in .fn)
        if getattr(obj, "__name__", None) == "relative_bias_score_mod":
            return True
    return False


def _set_score_bias(p, score_mod, aux_tensors, q: torch.Tensor) -> None:
    """unpack_aux_tensors (score_mod.py:30-41), into rx_{decode,extend}_params.score_bias*."""
    if score_mod is None:
        return  # (the reference ignores aux_tensors without a score_mod)
    if not _is_relative_bias(score_mod):
        raise NotImplementedError("score_mod: the HIP path implements relative_bias_score_mod only (a Triton callable "
                                  "cannot be inlined behind a C ABI)")
    assert aux_tensors is not None and len(aux_tensors) == 1, "Triton score_mod currently requires exactly one aux tensor"
    aux0 = aux_tensors[0]
    assert aux0.dim() == 3 and aux0.stride(2) == 1, (
        f"aux_tensors[0] must be 3D with a contiguous last dim, got shape={tuple(aux0.shape)} stride={aux0.stride()}")
    _require_cuda(aux0)
    if aux0.dtype not in (torch.float32, q.dtype):
        raise TypeError(f"aux_tensors[0] must be float32 or {q.dtype}, got {aux0.dtype}")
    if aux0.shape[0] < q.shape[0] or aux0.shape[1] != q.shape[1]:
        raise ValueError(f"aux_tensors[0] {tuple(aux0.shape)} does not cover q {tuple(q.shape)}")
    p.score_bias = aux0.data_ptr()
    p.score_bias_is_f32 = int(aux0.dtype == torch.float32)
    p.score_bias_len = int(aux0.shape[2])
    p.score_bias_stride_t, p.score_bias_stride_h = aux0.stride(0), aux0.stride(1)


# --------------------------------------------------------------------------------------
# K4-K6  decode_attention_fwd   kernels/ops/attention/decode_attention.py:968-1044
# --------------------------------------------------------------------------------------
def decode_attention_fwd(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse,
                         num_kv_splits, max_kv_splits, sm_scale, k_scale, v_scale, logit_cap=0.0,
                         sinks=None, xai_temperature_len=-1, has_mla=False, use_pdl=False,
                         page_size: int = 1, score_mod=None, aux_tensors=None, kv_layout=None, stages: int = 0,
                         merge_counters=None, k_new=None, v_new=None, request_order=None, split_items=None):
    """Same contract as the reference.  q [bs,Hq,Dk], o [bs,Hq,Dv], kv_indptr int32[bs+1],
    kv_indices int32/int64, attn_logits fp32[bs,Hq,max_kv_splits,Dv], attn_lse fp32[bs,Hq,S].
    ``max_kv_splits == 1`` (or num_kv_splits None) runs the single-pass kernel.  ``stages`` as in
    rx_decode_params (1 = the kv-split partials only).  ``score_mod`` = relative_bias_score_mod with
    ``aux_tensors = [rel_logits [bs, Hq, extent]]`` (decode_attention.py:215-227,539-551)."""
    # has_mla only selects a Triton block shape in the reference; here the MLA kernel is chosen from the
    # tensor shapes (Dk 576 / Dv 512, one kv head, V aliasing K)
    _require_cuda(q, k_buffer, v_buffer, o, kv_indptr, kv_indices)
    if max_kv_splits > 1:
        assert max_kv_splits == attn_logits.shape[2]
        assert q.shape[0] <= attn_logits.shape[0]
    assert q.shape[0] <= kv_indptr.shape[0] - 1
    p = _L.RxDecodeParams()
    _fill_decode_common(p, q, k_buffer, v_buffer, o, attn_logits, attn_lse, num_kv_splits,
                        max_kv_splits, sm_scale, k_scale, v_scale, logit_cap, sinks, page_size,
                        kv_layout)
    if kv_indptr.dtype != torch.int32:
        raise TypeError("kv_indptr must be int32")
    p.kv_indptr = kv_indptr.data_ptr()
    p.kv_indices = kv_indices.data_ptr()
    p.kv_indices_is_i64 = _is64(kv_indices, "kv_indices")
    p.xai_temperature_len = int(xai_temperature_len) if xai_temperature_len and xai_temperature_len > 0 else 0
    _set_score_bias(p, score_mod, aux_tensors, q)
    p.stages = int(stages)
    p.merge_counters = _merge_counters_ptr(merge_counters, q.shape[0], q.shape[1])
    _set_new_kv(p, k_new, v_new, q.shape[0])
    p.request_order = _request_order_ptr(request_order, q.shape[0])
    _set_split_items(p, split_items if p.max_kv_splits > 1 else None)
    _L.check(_L.load().rx_decode_attn(C.byref(p), _stream(q)), "rx_decode_attn")


def _set_rope(p, cos_sin_cache, positions, rotary_dim, is_neox_style, k_pe_tokens, k_new):
    """rx_decode_params.rope_*: fused RoPE of the latent decode (see decode_attention_fwd_grouped_rope)."""
    _require_cuda(cos_sin_cache, positions, k_pe_tokens, k_new)
    if cos_sin_cache.dim() != 2 or cos_sin_cache.stride(-1) != 1:
        raise ValueError("cos_sin_cache must be [max_pos, rotary_dim], contiguous in its rows")
    if cos_sin_cache.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError("cos_sin_cache: float32 or the 16-bit dtype of q")
    if positions.dtype not in (torch.int64, torch.int32) or not positions.is_contiguous():
        raise TypeError("positions must be a contiguous int64 / int32 vector")
    p.rope_cos_sin = cos_sin_cache.data_ptr()
    p.rope_cos_sin_is_f32 = int(cos_sin_cache.dtype == torch.float32)
    p.rope_cos_sin_stride = cos_sin_cache.stride(0)
    p.rope_positions = positions.data_ptr()
    p.rope_positions_is_i64 = int(positions.dtype == torch.int64)
    p.rope_dim = int(rotary_dim)
    p.rope_is_neox = int(bool(is_neox_style))
    if k_pe_tokens is not None:
        kp = k_pe_tokens.view(k_pe_tokens.shape[0], -1)
        if kp.shape[1] != 64 or kp.stride(1) != 1:
            raise ValueError("k_pe_tokens must hold [bs, 64] rows")
        p.rope_k_pe_out = kp.data_ptr()
        p.rope_k_pe_out_stride = kp.stride(0)
    if k_new is not None:
        kn = k_new.view(k_new.shape[0], -1)
        if kn.shape[1] != 576 or kn.stride(1) != 1:
            raise ValueError("k_new must hold [bs, 576] latent rows (512 latent + 64 rope, not rotated)")
        p.k_new = kn.data_ptr()
        p.k_new_stride_t, p.k_new_stride_h = kn.stride(0), 576
    p._keep_rope = (cos_sin_cache, positions, k_pe_tokens, k_new)


def decode_attention_fwd_grouped_rope(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, k_pe_tokens, kv_lora_rank,
                                      rotary_dim, cos_sin_cache, positions, attn_logits, num_kv_splits, sm_scale,
                                      logit_cap=0.0, use_rope=False, is_neox_style=False, page_size: int = 1,
                                      kv_layout=None, k_new=None, attn_lse=None, split_counts=None):
    """kernels/ops/attention/rocm_mla_decode_rope.py:402-439 -- the ROCm MLA decode that rotates q_pe and the newest
    token's k_pe inside the attention kernel (DeepSeek absorbed MLA: q [bs, Hq, 512 + 64], one latent kv head, v = the
    first ``kv_lora_rank`` columns of the same rows).  Same arguments: ``k_buffer`` holds every cached row, the step's
    own row included with its k_pe NOT yet rotated; the rotated k_pe comes back in ``k_pe_tokens`` [bs, 1, 64] and the
    pool is left alone (forward_mla_fused_rope_rocm.py:205-210 stores the row again).  ``attn_logits`` is the reference's
    [bs, Hq, num_kv_splits, kv_lora_rank + 1] scratch: accepted for signature parity -- this library keeps the LSEs in
    ``attn_lse`` (made here if not given) and lays its partials out itself.
    Beyond the reference: ``k_new`` [bs, 576] -- the step's rows NOT stored yet: the launch reads them from k_new,
    rotates, attends and stores the finished rows (RoPE + KV store + attention in one launch); ``page_size`` /
    ``kv_layout`` for paged pools (the reference kernel is page_size 1 only).  16-bit pools, rotary_dim 64."""
    _require_cuda(q, k_buffer, v_buffer, o, kv_indptr, kv_indices)
    if kv_lora_rank != 512 or q.shape[-1] != 576 or o.shape[-1] != 512:
        raise NotImplementedError("decode_attention_fwd_grouped_rope: the latent shape 512 + 64 only")
    bs, hq = q.shape[0], q.shape[1]
    S = max(1, int(num_kv_splits))
    # the caller's scratch is reused when it can hold this library's partials ([bs, Hq, S, 512] fp32, contiguous): a per-layer
    # allocation otherwise (graph-unfriendly -- ADVICE r4); the reference's own [.., kv_lora_rank + 1] layout is large enough
    logits = None
    if S > 1:
        n = bs * hq * S * 512
        if (attn_logits is not None and attn_logits.is_cuda and attn_logits.dtype == torch.float32 and attn_logits.is_contiguous()
                and attn_logits.numel() >= n and attn_logits.data_ptr() % 16 == 0):
            logits = attn_logits.view(-1)[:n].view(bs, hq, S, 512)
        else:
            logits = torch.empty((bs, hq, S, 512), dtype=torch.float32, device=q.device)
    lse = (attn_lse if attn_lse is not None else torch.empty((bs, hq, S), dtype=torch.float32, device=q.device)) if S > 1 else None
    if S > 1 and split_counts is None:
        split_counts = torch.full((bs,), S, dtype=torch.int32, device=q.device)  # the reference cuts every request S ways
    p = _L.RxDecodeParams()
    _fill_decode_common(p, q, k_buffer, v_buffer, o, logits, lse, split_counts, S, sm_scale, 1.0, 1.0, logit_cap, None,
                        page_size, kv_layout)
    if kv_indptr.dtype != torch.int32:
        raise TypeError("kv_indptr must be int32")
    p.kv_indptr, p.kv_indices = kv_indptr.data_ptr(), kv_indices.data_ptr()
    p.kv_indices_is_i64 = _is64(kv_indices, "kv_indices")
    if use_rope:
        if k_pe_tokens is None and k_new is None:
            raise ValueError("fused RoPE returns the rotated k_pe of the step's tokens: pass k_pe_tokens (or k_new)")
        _set_rope(p, cos_sin_cache, positions, rotary_dim, is_neox_style, k_pe_tokens, k_new)
    p._keep_scratch = (logits, lse, split_counts)
    _L.check(_L.load().rx_decode_attn(C.byref(p), _stream(q)), "rx_decode_attn")


def _decode_paged_params(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens,
                               attn_logits, attn_lse, num_kv_splits, max_kv_splits, sm_scale,
                               k_scale=1.0, v_scale=1.0, logit_cap=0.0, sinks=None,
                               page_size: int = 1, kv_layout=None, xai_temperature_len=-1,
                               kv_start=None, extra_o=None, extra_lse=None, stages: int = 0, extra_index=None):
    """MI355X-native entry: the kernel walks req_to_token itself (as the reference's CPU kernel
    decode_attention_cpu does, aot/csrc/cpu/decode.cpp:1586), so no kv_indices are materialised.
    ``kv_start`` int32[bs]: attend tokens [kv_start[b], seq_len_b) only; ``extra_o`` [P,bs,Hq,Dv] +
    ``extra_lse`` fp32[P,bs,Hq]: more partial results for stage 2 to merge (shared-prefix decode)."""
    _require_cuda(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens, kv_start, extra_o, extra_lse)
    if req_to_token.dtype != torch.int32:
        raise TypeError("req_to_token must be int32")
    p = _L.RxDecodeParams()
    _fill_decode_common(p, q, k_buffer, v_buffer, o, attn_logits, attn_lse, num_kv_splits,
                        max_kv_splits, sm_scale, k_scale, v_scale, logit_cap, sinks, page_size,
                        kv_layout)
    p.kv_indices = None
    p.req_to_token = req_to_token.data_ptr()
    p.req_row_stride = req_to_token.stride(0)
    p.req_pool_indices = req_pool_indices.data_ptr()
    p.req_pool_indices_is_i64 = _is64(req_pool_indices, "req_pool_indices")
    p.seq_lens = seq_lens.data_ptr()
    p.seq_lens_is_i64 = _is64(seq_lens, "seq_lens")
    p.xai_temperature_len = int(xai_temperature_len) if xai_temperature_len and xai_temperature_len > 0 else 0
    if kv_start is not None:
        if kv_start.dtype != torch.int32:
            raise TypeError("kv_start must be int32")
        p.kv_start = kv_start.data_ptr()
    if extra_o is not None:
        _set_extra_partials(p, extra_o, extra_lse, q, extra_index)
    p.stages = int(stages)
    return p


class SplitItems:
    """The live (request, split) pairs of a split schedule, compacted on the device (rx_split_items ->
    rx_decode_params.split_items): the decode kernel's grid then holds live work only, longest requests first, instead of
    bs x max_kv_splits split slots whose dead workgroups sit in front of a long request's later splits.

    ``cap`` sizes the grid: the exact pair count when the caller knows the counts on the host (eager forwards), an upper
    bound otherwise (graph replay: the surplus workgroups are at the END of the grid and exit at once)."""

    def __init__(self, max_items: int, device):
        self.items = torch.zeros(max(1, int(max_items)) * 2, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)  # sticky: a guarded build replaced a schedule
        self.cap = 0
        self.wgs_per_cu = 0

    def build(self, num_kv_splits, request_order=None, cap: Optional[int] = None, wgs_per_cu: int = 0,
              guarded: bool = False):
        """num_kv_splits int32[bs] (device), request_order int32[bs] or None; cap defaults to the table's size.
        wgs_per_cu = 3: the schedule was made for 3 x CUs pieces of a MIXED batch (get_num_kv_splits_balanced,
        wg_target_mixed) -- the launch takes the kernel's three-per-CU instance (rx_decode_params.split_items_wgs_per_cu).
        guarded (cap is a BOUND, not a count: graph replay): rx_split_items_guarded -- a schedule with more live pairs
        than cap is replaced on the device by one whole pass per request (num_kv_splits is rewritten) and
        ``self.overflow`` is set, instead of a table with pairs missing."""
        self.wgs_per_cu = int(wgs_per_cu)
        _require_cuda(num_kv_splits, request_order)
        bs = num_kv_splits.shape[0]
        if num_kv_splits.dtype != torch.int32 or (request_order is not None and request_order.dtype != torch.int32):
            raise TypeError("SplitItems.build: num_kv_splits / request_order must be int32")
        if request_order is not None and request_order.numel() < bs:
            raise ValueError("SplitItems.build: request_order shorter than the batch")
        self.cap = self.items.numel() // 2 if cap is None else int(cap)
        if self.cap > self.items.numel() // 2:
            raise ValueError(f"SplitItems.build: cap {self.cap} exceeds the table ({self.items.numel() // 2} pairs)")
        if guarded:
            st = _L.load().rx_split_items_guarded(_ptr(num_kv_splits), _ptr(request_order), bs, _ptr(self.items),
                                                  _ptr(self.count), self.cap, _ptr(self.overflow), _stream(num_kv_splits))
            _L.check(st, "rx_split_items_guarded")
            return self
        st = _L.load().rx_split_items(_ptr(num_kv_splits), _ptr(request_order), bs, _ptr(self.items), _ptr(self.count),
                                      self.cap, _stream(num_kv_splits))
        _L.check(st, "rx_split_items")
        return self


class DecodeUnits:
    """rx_decode_params.unit_desc / unit_first_slots: per-unit descriptors of ONE forward's decode launches, built on the
    device by rx_decode_units and shared by every layer (a unit = a (request, split) pair of a SplitItems table, or a whole
    request of an unsplit step).  They shorten each launch's prologue from five dependent round trips to three: the
    descriptor and the first tiles' slot ids sit at addresses a workgroup knows from its block index."""

    def __init__(self, max_units: int, device):
        self.max_units = max(1, int(max_units))
        self.desc = torch.zeros(self.max_units * 8, dtype=torch.int32, device=device)
        self.first = torch.zeros(self.max_units * 128, dtype=torch.int32, device=device)

    def build(self, req_to_token, req_pool_indices, seq_lens, num_kv_splits=None, max_kv_splits: int = 1, split_items=None,
              request_order=None):
        """From the SAME tensors the decode calls of this forward will pass.  split_items: the SplitItems of a split
        schedule (its cap rows are filled; rows past the live count are never read), or None for an unsplit step."""
        _require_cuda(req_to_token, req_pool_indices, seq_lens, num_kv_splits, request_order)
        if req_to_token.dtype != torch.int32:
            raise TypeError("req_to_token must be int32")
        bs = seq_lens.shape[0]
        cap = bs if split_items is None else int(split_items.cap)
        if cap > self.max_units:
            raise ValueError(f"DecodeUnits.build: {cap} units exceed the tables ({self.max_units})")
        st = _L.load().rx_decode_units(
            req_to_token.data_ptr(), req_to_token.stride(0), req_pool_indices.data_ptr(), _is64(req_pool_indices, "req_pool_indices"),
            seq_lens.data_ptr(), _is64(seq_lens, "seq_lens"), _ptr(num_kv_splits) if max_kv_splits > 1 else None, int(max_kv_splits),
            None if split_items is None else split_items.items.data_ptr(), None if split_items is None else split_items.count.data_ptr(),
            cap, _request_order_ptr(request_order, bs), bs, self.desc.data_ptr(), self.first.data_ptr(), _stream(req_to_token))
        _L.check(st, "rx_decode_units")
        return self


def _set_units(p, units):
    if units is None:
        p.unit_desc = p.unit_first_slots = None
    else:
        p.unit_desc, p.unit_first_slots = units.desc.data_ptr(), units.first.data_ptr()


def _set_split_items(p, split_items):
    if split_items is None:
        p.split_items, p.split_items_count, p.split_items_cap, p.split_items_wgs_per_cu = None, None, 0, 0
    else:
        p.split_items, p.split_items_count = split_items.items.data_ptr(), split_items.count.data_ptr()
        p.split_items_cap, p.split_items_wgs_per_cu = int(split_items.cap), int(split_items.wgs_per_cu)


def decode_attention_fwd_paged(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens,
                               attn_logits, attn_lse, num_kv_splits, max_kv_splits, sm_scale,
                               k_scale=1.0, v_scale=1.0, logit_cap=0.0, sinks=None,
                               page_size: int = 1, kv_layout=None, xai_temperature_len=-1,
                               kv_start=None, extra_o=None, extra_lse=None, stages: int = 0, merge_counters=None,
                               k_new=None, v_new=None, request_order=None, split_items=None,
                               score_mod=None, aux_tensors=None, units=None):
    p = _decode_paged_params(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens, attn_logits, attn_lse,
                             num_kv_splits, max_kv_splits, sm_scale, k_scale, v_scale, logit_cap, sinks, page_size,
                             kv_layout, xai_temperature_len, kv_start, extra_o, extra_lse, stages)
    p.merge_counters = _merge_counters_ptr(merge_counters, q.shape[0], q.shape[1])
    _set_new_kv(p, k_new, v_new, q.shape[0])
    p.request_order = _request_order_ptr(request_order, q.shape[0])
    _set_split_items(p, split_items if p.max_kv_splits > 1 else None)
    _set_score_bias(p, score_mod, aux_tensors, q)
    _set_units(p, units)
    p._keep_units = units
    _L.check(_L.load().rx_decode_attn(C.byref(p), _stream(q)), "rx_decode_attn")


def _set_extra_partials(p, extra_o, extra_lse, q, extra_index=None):
    bs, hq = q.shape[0], q.shape[1]
    if extra_o.dtype != q.dtype or extra_lse.dtype != torch.float32:
        raise TypeError("extra_o must have q's dtype, extra_lse float32")
    rows = bs if extra_index is None else extra_o.shape[1]
    if (extra_o.dim() != 4 or tuple(extra_o.shape[1:3]) != (rows, hq) or not extra_o.is_contiguous()
            or tuple(extra_lse.shape) != tuple(extra_o.shape[:3]) or not extra_lse.is_contiguous()):
        raise ValueError("extra_o must be contiguous [P, bs, Hq, Dv], extra_lse contiguous [P, bs, Hq] "
                         "(with extra_index: [P, rows, Hq, Dv] / [P, rows, Hq])")
    p.extra_o, p.extra_lse, p.num_extra_partials = extra_o.data_ptr(), extra_lse.data_ptr(), extra_o.shape[0]
    if extra_index is not None:
        if extra_index.dtype != torch.int32 or extra_index.numel() < bs or not extra_index.is_cuda:
            raise TypeError("extra_index must be a device int32[bs]")
        p.extra_index, p.extra_rows = extra_index.data_ptr(), rows
    else:
        p.extra_index, p.extra_rows = None, 0


def _fill_decode_common(p, q, k_buffer, v_buffer, o, attn_logits, attn_lse, num_kv_splits,
                        max_kv_splits, sm_scale, k_scale, v_scale, logit_cap, sinks, page_size,
                        kv_layout):
    if q.dim() != 3 or o.dim() != 3:
        raise ValueError("q/o must be [bs, heads, dim]")
    if q.stride(-1) != 1 or o.stride(-1) != 1:
        raise ValueError("q/o must be contiguous in head_dim")
    p.q, p.o = q.data_ptr(), o.data_ptr()
    p.q_stride_t, p.q_stride_h = q.stride(0), q.stride(1)
    p.o_stride_t, p.o_stride_h = o.stride(0), o.stride(1)
    p.kv = kv_layout if kv_layout is not None else _kv_layout(k_buffer, v_buffer, page_size)
    p.num_kv_splits = None if num_kv_splits is None else num_kv_splits.data_ptr()
    p.max_kv_splits = max(1, int(max_kv_splits)) if num_kv_splits is not None else 1
    if p.max_kv_splits > 1:
        if attn_logits.dtype != torch.float32 or attn_lse.dtype != torch.float32:
            raise TypeError("attn_logits / attn_lse must be float32")
        if not attn_logits.is_contiguous() or not attn_lse.is_contiguous():
            raise ValueError("attn_logits / attn_lse must be contiguous")
        if num_kv_splits.dtype != torch.int32:
            raise TypeError("num_kv_splits must be int32")
        p.attn_logits, p.attn_lse = attn_logits.data_ptr(), attn_lse.data_ptr()
    p.bs, p.num_q_heads = q.shape[0], q.shape[1]
    p.num_kv_heads = k_buffer.shape[-2] if kv_layout is None else k_buffer.shape[1]
    p.head_dim, p.v_head_dim = q.shape[-1], o.shape[-1]
    p.sm_scale, p.k_scale, p.v_scale, p.logit_cap = sm_scale, k_scale, v_scale, logit_cap
    if sinks is not None:
        if sinks.dtype != torch.float32:
            sinks = sinks.float()
        p._keep = sinks
        p.sinks = sinks.data_ptr()
    p.dtype = _rx_dtype(q)
    fp8 = _is_fp8_pool(k_buffer)
    if o.dtype != q.dtype or _is_fp8_pool(v_buffer) != fp8 or (not fp8 and (
            k_buffer.dtype != q.dtype or v_buffer.dtype != q.dtype)):
        raise TypeError("q and o must share one 16-bit dtype; k_buffer / v_buffer that dtype or fp8 e4m3fn")


def _set_new_kv(p, k_new, v_new, bs: int) -> None:
    """rx_decode_params.k_new / v_new: the step's KV store fused into the decode kernel ([bs, Hkv, D] rows)."""
    if k_new is None and v_new is None:
        p.k_new = p.v_new = None
        return
    if k_new is None or v_new is None or k_new.dim() != 3 or v_new.dim() != 3 or k_new.shape[0] != bs or v_new.shape[0] != bs:
        raise ValueError("k_new / v_new must both be [bs, Hkv, D]")
    if not (k_new.is_cuda and v_new.is_cuda) or k_new.stride(-1) != 1 or v_new.stride(-1) != 1:
        raise ValueError("k_new / v_new must be GPU tensors, contiguous in head_dim")
    p.k_new, p.v_new = k_new.data_ptr(), v_new.data_ptr()
    p.k_new_stride_t, p.k_new_stride_h = k_new.stride(0), k_new.stride(1)
    p.v_new_stride_t, p.v_new_stride_h = v_new.stride(0), v_new.stride(1)


def _request_order_ptr(request_order, bs: int):
    """rx_decode_params.request_order: int32[bs] permutation (launch order of the requests), or None."""
    if request_order is None:
        return None
    if (not request_order.is_cuda or request_order.dtype != torch.int32 or not request_order.is_contiguous()
            or request_order.numel() < bs):
        raise ValueError("request_order must be a contiguous int32 GPU tensor of bs entries")
    return request_order.data_ptr()


def _merge_counters_ptr(merge_counters, bs: int, num_q_heads: int):
    """rx_decode_params.merge_counters: zeroed int32 device memory of at least bs * Hq words (the kernels leave it
    zero), or None for the separate stage-2 launch."""
    if merge_counters is None:
        return None
    if (not merge_counters.is_cuda or merge_counters.dtype != torch.int32 or not merge_counters.is_contiguous()
            or merge_counters.numel() < bs * num_q_heads):
        raise ValueError("merge_counters must be a contiguous int32 GPU tensor of at least bs * num_q_heads words")
    return merge_counters.data_ptr()


class DecodeLauncher:
    """rx_decode_params pre-filled for one layer: the KV layout, head geometry and scales are set
    once; ``set_metadata`` writes the per-forward fields (shared by all layers of a forward) and
    ``__call__`` only patches q / o.  Keeps the per-layer host cost of a decode step to a few
    microseconds (the generic ``decode_attention_fwd*`` wrappers re-derive everything per call)."""

    def __init__(self, k_buffer, v_buffer, page_size, num_q_heads, num_kv_heads, head_dim, v_head_dim,
                 sm_scale, k_scale=1.0, v_scale=1.0, logit_cap=0.0, kv_layout=None, q_dtype=None):
        _require_cuda(k_buffer, v_buffer)
        self._lib = _L.load()
        p = self.p = _L.RxDecodeParams()
        p.kv = kv_layout if kv_layout is not None else _kv_layout(k_buffer, v_buffer, page_size)
        p.num_q_heads, p.num_kv_heads = num_q_heads, num_kv_heads
        p.head_dim, p.v_head_dim = head_dim, v_head_dim
        p.sm_scale, p.k_scale, p.v_scale, p.logit_cap = sm_scale, k_scale, v_scale, logit_cap
        if _is_fp8_pool(k_buffer):  # q / o dtype cannot be read off an fp8 pool
            if q_dtype not in (torch.bfloat16, torch.float16):
                raise TypeError("DecodeLauncher: an fp8 pool needs q_dtype = bfloat16 / float16")
            p.dtype = _L.RX_BF16 if q_dtype == torch.bfloat16 else _L.RX_F16
        else:
            p.dtype = _rx_dtype(k_buffer)
        p.max_kv_splits = 1
        self._ref = C.byref(p)
        self._keep = ()
        self.version = -1

    def set_metadata(self, version, bs, *, kv_indptr=None, kv_indices=None, req_to_token=None,
                     req_pool_indices=None, seq_lens=None, num_kv_splits=None, max_kv_splits=1,
                     attn_logits=None, attn_lse=None, merge_counters=None, request_order=None,
                     partial_pairs_hint: int = 0, split_items=None, units=None):
        p = self.p
        p.bs = bs
        p.partial_pairs_hint = int(partial_pairs_hint)
        p.merge_counters = _merge_counters_ptr(merge_counters, bs, p.num_q_heads)
        p.request_order = _request_order_ptr(request_order, bs)
        if kv_indices is not None:
            p.kv_indptr, p.kv_indices = kv_indptr.data_ptr(), kv_indices.data_ptr()
            p.kv_indices_is_i64 = _is64(kv_indices, "kv_indices")
        else:
            p.kv_indices = None
            p.req_to_token, p.req_row_stride = req_to_token.data_ptr(), req_to_token.stride(0)
            p.req_pool_indices = req_pool_indices.data_ptr()
            p.req_pool_indices_is_i64 = _is64(req_pool_indices, "req_pool_indices")
            p.seq_lens, p.seq_lens_is_i64 = seq_lens.data_ptr(), _is64(seq_lens, "seq_lens")
        if num_kv_splits is not None and max_kv_splits > 1:
            p.num_kv_splits, p.max_kv_splits = num_kv_splits.data_ptr(), max_kv_splits
            p.attn_logits, p.attn_lse = attn_logits.data_ptr(), attn_lse.data_ptr()
        else:
            p.num_kv_splits, p.max_kv_splits = None, 1
        _set_split_items(p, split_items if p.max_kv_splits > 1 else None)
        _set_units(p, units if kv_indices is None else None)
        self._keep = (kv_indptr, kv_indices, req_to_token, req_pool_indices, seq_lens, num_kv_splits,
                      attn_logits, attn_lse, merge_counters, request_order, split_items, units)
        self.version = version

    def can_fuse_store(self) -> bool:
        """The new token's KV store can ride in the decode launch (rx_decode_params.k_new): D = 64 / 128 on a 16-bit
        pool with at most 16 q heads per kv head."""
        p = self.p
        return (not p.kv.kv_fp8 and p.head_dim == p.v_head_dim and p.head_dim in (64, 128)
                and p.num_q_heads <= 16 * p.num_kv_heads)

    def __call__(self, q3, o3, stream_ptr, sinks=None, k_new=None, v_new=None):
        p = self.p
        _set_new_kv(p, k_new, v_new, q3.shape[0])
        p.q, p.o = q3.data_ptr(), o3.data_ptr()
        p.q_stride_t, p.q_stride_h = q3.stride(0), q3.stride(1)
        p.o_stride_t, p.o_stride_h = o3.stride(0), o3.stride(1)
        p.sinks = None if sinks is None else sinks.data_ptr()
        st = self._lib.rx_decode_attn(self._ref, stream_ptr)
        if st:
            _L.check(st, "rx_decode_attn")


def shared_prefix_plan(req_to_token, req_pool_indices, seq_lens, plan, chunk_indptr, shared_indices, kv_start,
                       suffix_lens, *, min_shared: int = 0, chunk_align: int = 64) -> None:
    """rx_shared_prefix_plan: batch-wide common prefix of the req_to_token rows, found on the device.
    All outputs int32: plan[2], chunk_indptr[num_chunks+1], shared_indices[max_shared], kv_start[bs],
    suffix_lens[bs] (include/radix_hip.h)."""
    _require_cuda(req_to_token, req_pool_indices, seq_lens, plan, chunk_indptr, shared_indices, kv_start, suffix_lens)
    for name, t in (("req_to_token", req_to_token), ("plan", plan), ("chunk_indptr", chunk_indptr),
                    ("shared_indices", shared_indices), ("kv_start", kv_start), ("suffix_lens", suffix_lens)):
        if t.dtype != torch.int32:
            raise TypeError(f"{name} must be int32")
    bs = req_pool_indices.shape[0]
    if plan.numel() < 2 or kv_start.numel() < bs or suffix_lens.numel() < bs or chunk_indptr.numel() < 2:
        raise ValueError("shared_prefix_plan: output too small")
    max_shared = min(shared_indices.numel(), req_to_token.shape[1])
    st = _L.load().rx_shared_prefix_plan(_ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices),
                                         _is64(req_pool_indices, "req_pool_indices"), _ptr(seq_lens),
                                         _is64(seq_lens, "seq_lens"), bs, max_shared, int(min_shared),
                                         chunk_indptr.numel() - 1, int(chunk_align), _ptr(plan), _ptr(chunk_indptr),
                                         _ptr(shared_indices), _ptr(kv_start), _ptr(suffix_lens),
                                         _stream(req_to_token))
    _L.check(st, "rx_shared_prefix_plan")


class CascadeDecode:
    """Shared-prefix (cascade) decode attention for one batch geometry (the reference only has the
    merge building block, kernels/ops/attention/merge_state.py:8-96).

    When the requests of a decode batch carry the same leading slots in their ``req_to_token`` rows (a radix
    hit), the per-request decode kernel reads those K/V rows bs times.  Here they are read once:

    * ``plan`` (once per forward, shared by all layers; no host sync): common prefix length L on the device,
      the prefix cut into ``num_chunks`` pieces, per-request suffix starts / lengths and suffix kv splits;
    * ``__call__`` (per layer): (1) the extend MFMA kernel over the shared rows with all bs decode queries as
      its M dimension -- one pseudo-request per chunk, so the launch still fills the chip; (2) the decode
      kernel over the suffixes ``[L, seq_len_b)``; its stage 2 merges the chunk partials with the kv splits.

    The result equals plain decode attention up to the partials' 16-bit rounding.  Not available with
    sliding windows or the Grok temperature (position dependent per request)."""

    def __init__(self, max_bs: int, num_q_heads: int, num_kv_heads: int, head_dim: int, dtype, device,
                 max_shared: int, cu_count: int = 256, min_shared: int = 1024, max_kv_splits: int = 16,
                 num_chunks: Optional[int] = None, overlap: bool = False, v_head_dim: Optional[int] = None):
        v_head_dim = head_dim if v_head_dim is None else int(v_head_dim)
        self.mla = (head_dim, v_head_dim) == (576, 512) and num_kv_heads == 1   # latent rows: rx::extend_mla_kernel
        if not self.mla and (head_dim not in (64, 96, 128, 256) or v_head_dim != head_dim):
            raise ValueError("CascadeDecode: head_dim 64 / 96 / 128 / 256, or the latent MLA shape 576 / 512 over one kv head "
                             "(the MFMA extend and decode kernels)")
        self.max_bs, self.hq, self.hkv, self.d, self.dv = max_bs, num_q_heads, num_kv_heads, head_dim, v_head_dim
        self.cu_count, self.min_shared, self.max_shared = cu_count, int(min_shared), int(max_shared)
        # chunks of the shared prefix = pseudo-requests of phase 1.  Its workgroups = chunks * ceil(bs / 128) * Hq
        # should cover every CU once, so the count follows the ACTUAL batch (a pool sized for thousands of requests
        # usually runs batches far smaller): chosen per plan(), buffers sized for the worst bs <= max_bs
        self._fixed_chunks = None if num_chunks is None else int(num_chunks)
        self.num_chunks = self._chunks_for(max_bs)
        cands = list(range(1, min(max_bs, 2048) + 1)) + [max_bs]
        rows_max = max(self._chunks_for(b) * b for b in cands)
        S_max = max(self._chunks_for(b) for b in cands)
        self.split_cap = max(2, int(max_kv_splits))
        i32 = dict(dtype=torch.int32, device=device)
        self.plan_buf = torch.zeros(2, **i32)
        self._chunk_indptr_buf = torch.zeros(S_max + 1, **i32)
        self.chunk_indptr = self._chunk_indptr_buf[: self.num_chunks + 1]
        self.shared_indices = torch.zeros(max(1, self.max_shared), **i32)
        self.kv_start = torch.zeros(max_bs, **i32)
        self.suffix_lens = torch.zeros(max_bs, **i32)
        self.num_kv_splits = torch.ones(max_bs, **i32)
        self.q_rep = torch.zeros(rows_max * num_q_heads * head_dim, dtype=dtype, device=device)
        self.o_parts = torch.zeros(rows_max * num_q_heads * v_head_dim, dtype=dtype, device=device)
        self.lse_parts = torch.zeros(rows_max * num_q_heads, dtype=torch.float32, device=device)
        self._qo_indptr_buf = torch.zeros(S_max + 1, **i32)
        self.qo_indptr = self._qo_indptr_buf[: self.num_chunks + 1]
        # suffix partials: bs * S(bs) <= cu / (Hkv * ceil(G / 16)) + bs rows per head (native schedule), >= 2 slots
        group = max(1, num_q_heads // num_kv_heads)
        rows = (max(2 * max_bs, cu_count // (num_kv_heads * ((group + 15) // 16)) + max_bs) + 1) * num_q_heads
        self.attn_logits = torch.empty(rows * v_head_dim, dtype=torch.float32, device=device)
        self.attn_lse = torch.empty(rows, dtype=torch.float32, device=device)
        self.bs = 0
        self._params = {}  # (k ptr, v ptr, bs, scalars) -> filled rx_extend_params / rx_decode_params
        self._lib = _L.load()
        # overlap: phase 1 (MFMA-bound) on a side stream next to phase 2's stage 1 (HBM-bound), joined before
        # stage 2.  Measured at bs=256 / 3584 shared / 512 private: 166 -> 156 us per layer, but the two extra
        # stream waits cost more host time than that on small batches -- off by default.
        self._side = torch.cuda.Stream(device=device) if overlap else None

    def _chunks_for(self, bs: int) -> int:
        if self._fixed_chunks is not None:
            return self._fixed_chunks
        if self.mla:  # rows are (query, head) pairs there: ceil(bs * Hq / 128) workgroups per chunk
            return max(1, min(16, -(-self.cu_count // -(-bs * self.hq // 128))))
        return max(1, min(16, -(-self.cu_count // (-(-bs // 128) * self.hq))))

    def plan(self, req_to_token, req_pool_indices, seq_lens) -> None:
        bs = self.bs = req_pool_indices.shape[0]
        if bs > self.max_bs:
            raise ValueError(f"CascadeDecode: bs {bs} > max_bs {self.max_bs}")
        self._tabs = (req_to_token, req_pool_indices, seq_lens)
        self.num_chunks = self._chunks_for(bs)
        self.chunk_indptr = self._chunk_indptr_buf[: self.num_chunks + 1]
        self.qo_indptr = self._qo_indptr_buf[: self.num_chunks + 1]
        shared_prefix_plan(req_to_token, req_pool_indices, seq_lens, self.plan_buf, self.chunk_indptr,
                           self.shared_indices, self.kv_start[:bs], self.suffix_lens[:bs],
                           min_shared=self.min_shared)
        # suffix kv splits: the native schedule on the suffix lengths.  One split (a batch that fills the chip by
        # itself): the decode kernel's single-pass epilogue folds the chunk partials in -- no fp32 partials, no
        # stage-2 launch.  (The side-stream overlap joins before stage 2, so it keeps >= 2 slots.)
        self.max_kv_splits = native_max_kv_splits(bs, self.hq, self.hkv, self.cu_count, self.split_cap)
        if self._side is not None or self.mla:  # (the MLA decode kernel has no single-pass fold: stage 2 merges)
            self.max_kv_splits = max(2, self.max_kv_splits)
        if self.max_kv_splits > 1:
            get_num_kv_splits_native(self.num_kv_splits[:bs], self.suffix_lens[:bs], self.hq, self.hkv,
                                     self.max_kv_splits, self.cu_count)
        S = self.num_chunks
        torch.arange(0, (S + 1) * bs, bs, out=self.qo_indptr)
        if bs * self.hq * self.max_kv_splits > self.attn_lse.numel():
            raise RuntimeError("CascadeDecode: partials scratch too small")  # sized for every bs <= max_bs

    def shared_len(self) -> int:
        """Host copy of L (synchronises; diagnostics / tests only)."""
        return int(self.plan_buf[0].item())

    def __call__(self, q, k_buffer, v_buffer, o, sm_scale, k_scale=1.0, v_scale=1.0, logit_cap=0.0, sinks=None,
                 page_size: int = 1, kv_layout=None, k_new=None, v_new=None) -> None:
        """k_new / v_new: the step's KV store rides in the suffix launch (rx_decode_params.k_new): the newest token of
        every request is the last of its suffix."""
        bs, S = self.bs, self.num_chunks
        if q.shape != (bs, self.hq, self.d):
            raise ValueError(f"CascadeDecode: q {tuple(q.shape)} != {(bs, self.hq, self.d)}")
        req_to_token, req_pool_indices, seq_lens = self._tabs
        # phase 1: every chunk of the shared prefix against all bs queries (M = bs per head)
        q_rep = self.q_rep[: S * bs * self.hq * self.d].view(S, bs, self.hq, self.d)
        o_parts = self.o_parts[: S * bs * self.hq * self.dv].view(S, bs, self.hq, self.dv)
        lse_parts = self.lse_parts[: S * bs * self.hq].view(S, bs, self.hq)
        attn_logits = self.attn_logits[: bs * self.hq * self.max_kv_splits * self.dv].view(
            bs, self.hq, self.max_kv_splits, self.dv)
        attn_lse = self.attn_lse[: bs * self.hq * self.max_kv_splits].view(bs, self.hq, self.max_kv_splits)
        qf, of = q_rep.view(S * bs, self.hq, self.d), o_parts.view(S * bs, self.hq, self.dv)

        # the two parameter blocks depend on the layer's buffers, bs and a few scalars only: filled once, then
        # just q / o / sinks are patched (per-layer host cost: one copy + two ctypes calls)
        key = (k_buffer.data_ptr(), v_buffer.data_ptr(), bs, self.max_kv_splits, float(sm_scale), float(k_scale),
               float(v_scale), float(logit_cap), page_size, req_to_token.data_ptr())
        ent = self._params.get(key)
        if ent is None:
            if len(self._params) > 4096:
                self._params.clear()
            pe = _extend_params(qf, qf, qf[..., : self.dv], of, k_buffer, v_buffer, self.qo_indptr, self.chunk_indptr,
                                self.shared_indices, None, False, None, bs, k_scale, 1.0, sm_scale=sm_scale,
                                logit_cap=logit_cap, lse_extend=lse_parts.view(S * bs, self.hq), skip_extend=True,
                                page_size=page_size, kv_layout=kv_layout, _num_kv_heads=self.hkv, avg_kv_len_hint=0)
            pd = _decode_paged_params(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens, attn_logits,
                                      attn_lse, self.num_kv_splits[:bs] if self.max_kv_splits > 1 else None,
                                      self.max_kv_splits, sm_scale, k_scale,
                                      v_scale, logit_cap, None, page_size, kv_layout, kv_start=self.kv_start[:bs],
                                      extra_o=o_parts, extra_lse=lse_parts)
            ent = self._params[key] = (pe, C.byref(pe), pd, C.byref(pd), (k_buffer, v_buffer, req_to_token))
        pe, pe_ref, pd, pd_ref, _ = ent
        # per-forward tensors of the caller (fresh every step in eager serving): patched, not part of the key
        pd.req_pool_indices, pd.req_pool_indices_is_i64 = req_pool_indices.data_ptr(), _is64(req_pool_indices, "req_pool_indices")
        pd.seq_lens, pd.seq_lens_is_i64 = seq_lens.data_ptr(), _is64(seq_lens, "seq_lens")
        if q.stride(-1) != 1 or o.stride(-1) != 1 or q.dtype != self.q_rep.dtype or o.dtype != q.dtype:
            raise ValueError("CascadeDecode: q / o must be contiguous in head_dim and of the planned dtype")
        pd.q, pd.o = q.data_ptr(), o.data_ptr()
        pd.q_stride_t, pd.q_stride_h, pd.o_stride_t, pd.o_stride_h = q.stride(0), q.stride(1), o.stride(0), o.stride(1)
        if sinks is not None and sinks.dtype != torch.float32:
            sinks = sinks.float()
        pd.sinks = None if sinks is None else sinks.data_ptr()
        self._sinks_keep = sinks
        _set_new_kv(pd, k_new, v_new, bs)
        lib = self._lib

        def phase1():
            q_rep.copy_(q.unsqueeze(0).expand(S, -1, -1, -1))
            st = lib.rx_extend_attn(pe_ref, _stream(q))
            if st:
                _L.check(st, "rx_extend_attn")

        def phase2(stages):  # suffixes [L, seq_len_b); stage 2 merges everything (sinks join once, there)
            pd.stages = stages
            st = lib.rx_decode_attn(pd_ref, _stream(q))
            if st:
                _L.check(st, "rx_decode_attn")

        if self._side is None:
            phase1()
            phase2(0)
            return
        cur = torch.cuda.current_stream(q.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            phase1()
        phase2(1)
        cur.wait_stream(self._side)
        phase2(2)


class CascadeGroups:
    """Shared-prefix decode with SEVERAL prefixes in one batch: one per radix-tree node the scheduler found requests
    under (RadixCache.match_prefix groups, srt/mem_cache/radix_cache.py:352-430; plan them with
    mem_cache.radix_cache.plan_shared_prefix_groups).  CascadeDecode is the one-group case with a device-side plan.

    * ``plan(req_to_token, req_pool_indices, seq_lens, groups)`` once per forward: ``groups`` = [(member batch rows,
      shared token count)], disjoint.  Host work is O(groups * chunks); the slot lists are gathered on the device.
    * ``__call__`` per layer: (1) ONE launch of the extend MFMA kernel over pseudo-requests (chunk c, group g): the
      queries of the group's members (gathered, M rows in group order) against chunk c of the group's prefix;
      (2) the decode kernel over every request's own suffix [L_g(b), seq_len_b) (the whole row for requests in no
      group); its stage 2 / single-pass epilogue merges the chunk partials of the request's row
      (rx_decode_params.extra_index).

    Same result as plain decode attention up to the partials' 16-bit rounding (merge_state.py:8-96 is the reference's
    building block)."""

    def __init__(self, max_bs: int, num_q_heads: int, num_kv_heads: int, head_dim: int, dtype, device,
                 max_shared_total: int, cu_count: int = 256, max_groups: int = 32, max_chunks: int = 16,
                 max_kv_splits: int = 16):
        if head_dim not in (64, 96, 128, 256):
            raise ValueError("CascadeGroups: head_dim 64 / 96 / 128 / 256 (the MFMA extend and decode kernels)")
        self.max_bs, self.hq, self.hkv, self.d = int(max_bs), num_q_heads, num_kv_heads, head_dim
        self.cu_count, self.max_groups, self.max_chunks = int(cu_count), int(max_groups), int(max_chunks)
        self.split_cap = max(2, int(max_kv_splits))
        self.device = device
        i32 = dict(dtype=torch.int32, device=device)
        P = self.max_groups * self.max_chunks
        self.max_shared_total = int(max_shared_total)
        self.shared_indices = torch.zeros(max(1, self.max_shared_total), **i32)
        # one packed int32 table per plan: [kv_indptr P+1 | qo_indptr P+1 | src_row P | src_col P | kv_start bs |
        # extra_index bs | gather C*bs]
        self._tab_len = 2 * (P + 1) + 2 * P + 2 * self.max_bs + self.max_chunks * self.max_bs  # (int32 words only)
        self._tab_host = torch.zeros(self._tab_len, dtype=torch.int32).pin_memory() if torch.cuda.is_available() \
            else torch.zeros(self._tab_len, dtype=torch.int32)
        self._tab = torch.zeros(self._tab_len, **i32)
        self.suffix_lens = torch.zeros(self.max_bs, **i32)
        self.num_kv_splits = torch.ones(self.max_bs, **i32)
        # rows of the gathered queries / chunk partials: C * M with C = layout()'s chunk count, which shrinks as the member
        # count M grows (C <= ceil(CUs / (ceil(M / 128) * Hq))) -- the maximum over reachable layouts, not
        # max_chunks * max_bs (16x oversized, ~1 GB at a 4096-request pool)
        rows = self.rows_max = self.rows_bound(self.max_bs, num_q_heads, self.cu_count, self.max_chunks)
        self.q_rep = torch.zeros(rows * num_q_heads * head_dim, dtype=dtype, device=device)
        self.o_parts = torch.zeros(rows * num_q_heads * head_dim, dtype=dtype, device=device)
        self.lse_parts = torch.zeros(rows * num_q_heads, dtype=torch.float32, device=device)
        group = max(1, num_q_heads // num_kv_heads)
        prow = (max(2 * self.max_bs, cu_count // (num_kv_heads * ((group + 15) // 16)) + self.max_bs) + 1) * num_q_heads
        self.attn_logits = torch.empty(prow * head_dim, dtype=torch.float32, device=device)
        self.attn_lse = torch.empty(prow, dtype=torch.float32, device=device)
        self._lib = _L.load()
        self.bs = self.members = self.num_chunks = self.num_groups = 0

    @staticmethod
    def rows_bound(max_bs: int, hq: int, cu_count: int, max_chunks: int) -> int:
        """max over member counts M <= max_bs of layout()'s C * M: C <= min(max_chunks, ceil(CUs / (ceil(M/128) * Hq)))
        (layout's workgroup count sum(ceil(size_g / 128)) * Hq is at least ceil(M / 128) * Hq)."""
        return max(min(max_chunks, max(1, -(-cu_count // (-(-m // 128) * hq)))) * m for m in range(1, max_bs + 1))

    @staticmethod
    def layout(groups, bs: int, hq: int, cu_count: int, max_chunks: int, chunk_align: int = 64):
        """Host half of the plan (pure numpy, no device): returns a dict of int32 arrays -- the pseudo-request tables
        of phase 1 and the per-request tables of phase 2.  Pseudo-request p = c * G + g; the shared slot list is laid
        out in the same order, so both indptr arrays are monotone."""
        G = len(groups)
        sizes = np.asarray([len(m) for m, _ in groups], dtype=np.int64)
        lens = np.asarray([int(L) for _, L in groups], dtype=np.int64)
        members = np.concatenate([np.asarray(m, dtype=np.int64) for m, _ in groups]) if G else np.zeros(0, np.int64)
        if len(np.unique(members)) != len(members) or (members < 0).any() or (members >= bs).any():
            raise ValueError("CascadeGroups: groups must be disjoint sets of batch rows")
        M = int(sizes.sum())
        wgs = int((-(-sizes // 128)).sum()) * hq  # phase-1 workgroups per chunk
        C = int(max(1, min(max_chunks, -(-cu_count // max(1, wgs)))))
        per = -(-(-(-lens // C)) // chunk_align) * chunk_align          # chunk length of each group, aligned
        starts = np.concatenate([[0], np.cumsum(sizes)])[:-1]
        kv_indptr = np.zeros(C * G + 1, np.int64)
        qo_indptr = np.zeros(C * G + 1, np.int64)
        src_col = np.zeros(C * G, np.int64)
        for c in range(C):
            lo = np.minimum(c * per, lens)
            hi = np.minimum((c + 1) * per, lens)
            kv_indptr[c * G + 1: (c + 1) * G + 1] = hi - lo
            src_col[c * G: (c + 1) * G] = lo
            qo_indptr[c * G: (c + 1) * G] = c * M + starts
        qo_indptr[C * G] = C * M
        kv_indptr = np.cumsum(kv_indptr)
        kv_start = np.zeros(bs, np.int64)
        extra_index = np.full(bs, -1, np.int64)
        if G:
            kv_start[members] = np.repeat(lens, sizes)
            extra_index[members] = np.arange(M)
        gather = (np.tile(members, C) if M else np.zeros(0, np.int64))
        first = np.asarray([int(m[0]) for m, _ in groups], dtype=np.int64) if G else np.zeros(0, np.int64)
        return dict(C=C, G=G, M=M, kv_indptr=kv_indptr, qo_indptr=qo_indptr, src_col=src_col,
                    src_member=np.tile(first, C), kv_start=kv_start, extra_index=extra_index, gather=gather,
                    max_group=int(sizes.max()) if G else 0, total_shared=int(kv_indptr[-1]))

    def plan(self, req_to_token, req_pool_indices, seq_lens, groups) -> None:
        bs = self.bs = req_pool_indices.shape[0]
        if bs > self.max_bs or len(groups) > self.max_groups:
            raise ValueError(f"CascadeGroups: bs {bs} / {len(groups)} groups beyond max_bs {self.max_bs} / max_groups {self.max_groups}")
        lay = self.layout(groups, bs, self.hq, self.cu_count, self.max_chunks)
        C, G, M = lay["C"], lay["G"], lay["M"]
        if lay["total_shared"] > self.max_shared_total:
            raise ValueError(f"CascadeGroups: {lay['total_shared']} shared tokens > max_shared_total {self.max_shared_total}")
        if C * M > self.rows_max:
            raise ValueError(f"CascadeGroups: {C} chunks x {M} member rows beyond the {self.rows_max} rows allocated")
        self.num_chunks, self.num_groups, self.members, self.max_group = C, G, M, lay["max_group"]
        self._tabs = (req_to_token, req_pool_indices, seq_lens)
        P = C * G
        parts = [lay["kv_indptr"], lay["qo_indptr"], lay["src_member"], lay["src_col"], lay["kv_start"],
                 lay["extra_index"], lay["gather"]]
        offs = np.concatenate([[0], np.cumsum([len(x) for x in parts])])
        host = self._tab_host
        host[: offs[-1]] = torch.from_numpy(np.concatenate(parts).astype(np.int32))
        # (a blocking copy: the pinned staging buffer is rewritten by the next plan(), which may run before the GPU has
        # reached this one)
        self._tab[: offs[-1]].copy_(host[: offs[-1]], non_blocking=False)
        view = lambda i: self._tab[int(offs[i]): int(offs[i + 1])]
        self.kv_indptr, self.qo_indptr, src_member, src_col = view(0), view(1), view(2), view(3)
        self.kv_start, self.extra_index, self.gather = view(4), view(5), view(6).long()
        # the shared slot lists, gathered from the first member's req_to_token row (every member holds the same slots)
        T = lay["total_shared"]
        if T:
            pos = torch.arange(T, dtype=torch.int32, device=self.device)
            pr = torch.searchsorted(self.kv_indptr[1:].contiguous(), pos, right=True)
            col = src_col[pr].long() + (pos - self.kv_indptr[pr]).long()
            row = req_pool_indices.long()[src_member[pr].long()]
            self.shared_indices[:T] = req_to_token[row, col]
        torch.sub(seq_lens.to(torch.int32), self.kv_start, out=self.suffix_lens[:bs])
        self.max_kv_splits = native_max_kv_splits(bs, self.hq, self.hkv, self.cu_count, self.split_cap)
        if self.max_kv_splits > 1:
            get_num_kv_splits_native(self.num_kv_splits[:bs], self.suffix_lens[:bs], self.hq, self.hkv,
                                     self.max_kv_splits, self.cu_count)

    def __call__(self, q, k_buffer, v_buffer, o, sm_scale, k_scale=1.0, v_scale=1.0, logit_cap=0.0, sinks=None,
                 page_size: int = 1, kv_layout=None, k_new=None, v_new=None) -> None:
        bs, nc, M = self.bs, self.num_chunks, self.members
        if q.shape != (bs, self.hq, self.d):
            raise ValueError(f"CascadeGroups: q {tuple(q.shape)} != {(bs, self.hq, self.d)}")
        req_to_token, req_pool_indices, seq_lens = self._tabs
        S = self.max_kv_splits
        attn_logits = self.attn_logits[: bs * self.hq * S * self.d].view(bs, self.hq, S, self.d)
        attn_lse = self.attn_lse[: bs * self.hq * S].view(bs, self.hq, S)
        extra_o = extra_lse = None
        if M:
            rows = nc * M
            q_rep = self.q_rep[: rows * self.hq * self.d].view(rows, self.hq, self.d)
            o_parts = self.o_parts[: rows * self.hq * self.d].view(rows, self.hq, self.d)
            lse_parts = self.lse_parts[: rows * self.hq].view(rows, self.hq)
            torch.index_select(q, 0, self.gather, out=q_rep)
            pe = _extend_params(q_rep, q_rep, q_rep, o_parts, k_buffer, v_buffer, self.qo_indptr, self.kv_indptr,
                                self.shared_indices, None, False, None, self.max_group, k_scale, 1.0, sm_scale=sm_scale,
                                logit_cap=logit_cap, lse_extend=lse_parts, skip_extend=True, page_size=page_size,
                                kv_layout=kv_layout, _num_kv_heads=self.hkv, avg_kv_len_hint=0)
            _L.check(self._lib.rx_extend_attn(C.byref(pe), _stream(q)), "rx_extend_attn")
            extra_o = o_parts.view(nc, M, self.hq, self.d)
            extra_lse = lse_parts.view(nc, M, self.hq)
        pd = _decode_paged_params(q, k_buffer, v_buffer, o, req_to_token, req_pool_indices, seq_lens, attn_logits, attn_lse,
                                  self.num_kv_splits[:bs] if S > 1 else None, S, sm_scale, k_scale, v_scale, logit_cap,
                                  sinks, page_size, kv_layout, kv_start=self.kv_start, extra_o=extra_o,
                                  extra_lse=extra_lse, extra_index=self.extra_index if M else None)
        _set_new_kv(pd, k_new, v_new, bs)
        _L.check(self._lib.rx_decode_attn(C.byref(pd), _stream(q)), "rx_decode_attn")


class StoreLayoutLauncher:
    """rx_store_kv_layout with the pool-side arguments of one layer pre-computed (HND pools)."""

    def __init__(self, layout, num_kv_heads, head_dim, v_head_dim, size_limit, err_flag,
                 reserved_skip_index=0):
        self._lib = _L.load()
        self.layout, self._ref = layout, C.byref(layout)
        self.h, self.dk, self.dv = num_kv_heads, head_dim, v_head_dim
        self.size_limit, self.skip = size_limit, reserved_skip_index
        self.err = None if err_flag is None else err_flag.data_ptr()
        self._err_keep = err_flag

    def __call__(self, k2, v2, loc, stream_ptr):
        st = self._lib.rx_store_kv_layout(k2.data_ptr(), v2.data_ptr(), self._ref, loc.data_ptr(),
                                          k2.shape[0], self.h, self.dk, self.dv, k2.stride(0),
                                          v2.stride(0), 1 if loc.dtype == torch.int64 else 0,
                                          self.size_limit, self.skip, self.err, stream_ptr)
        if st:
            _L.check(st, "rx_store_kv_layout")


class StoreLauncher:
    """rx_store_kv with the cache-side arguments of one layer pre-computed."""

    def __init__(self, k_cache2d, v_cache2d, size_limit, err_flag, reserved_skip_index=0):
        _require_cuda(k_cache2d, v_cache2d)
        self._lib = _L.load()
        self.kc, self.vc = k_cache2d, v_cache2d
        es = k_cache2d.element_size()
        self._tail = None
        self.k_row_bytes = k_cache2d.shape[1] * es
        self.v_row_bytes = v_cache2d.shape[1] * es
        self.kcs, self.vcs = k_cache2d.stride(0) * es, v_cache2d.stride(0) * es
        self.es = es
        self.size_limit, self.skip = size_limit, reserved_skip_index
        self.err = None if err_flag is None else err_flag.data_ptr()
        self._err_keep = err_flag

    def __call__(self, k2, v2, loc, stream_ptr):
        st = self._lib.rx_store_kv(k2.data_ptr(), v2.data_ptr(), self.kc.data_ptr(), self.vc.data_ptr(),
                                   loc.data_ptr(), k2.shape[0], self.k_row_bytes, self.v_row_bytes,
                                   k2.stride(0) * self.es, v2.stride(0) * self.es, self.kcs, self.vcs,
                                   1 if loc.dtype == torch.int64 else 0, self.size_limit, self.skip,
                                   self.err, stream_ptr)
        if st:
            _L.check(st, "rx_store_kv")


# --------------------------------------------------------------------------------------
# K7  extend_attention_fwd      kernels/ops/attention/extend_attention.py:664-812
# --------------------------------------------------------------------------------------
def _extend_params(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
                         kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr,
                         max_len_extend, k_scale, v_scale, sm_scale=None, logit_cap=0.0,
                         skip_prefix_custom_mask=True, sliding_window_size=-1, sinks=None,
                         window_kv_offsets=None, xai_temperature_len=-1, lse_extend=None,
                         skip_prefix=False, skip_extend=False, page_size: int = 1,
                         score_mod=None, aux_tensors=None, kv_layout=None, unified_prefix_lens=None,
                         _num_kv_heads=None, avg_kv_len_hint=None, q_pack: int = 1):
    _require_cuda(q_extend, k_extend, v_extend, o_extend, qo_indptr, kv_indptr, custom_mask, mask_indptr,
                  window_kv_offsets, unified_prefix_lens)
    unified = unified_prefix_lens is not None
    for name, t in (("q", q_extend), ("k", k_extend), ("v", v_extend), ("o", o_extend)):
        if t is None and unified and name in ("k", "v"):
            continue
        if t.dim() != 3 or t.stride(-1) != 1:
            raise ValueError(f"{name}_extend must be [T, heads, dim] and contiguous in dim")
    if kv_indptr.dtype != torch.int32:
        raise TypeError("kv_indptr must be int32")
    p = _L.RxExtendParams()
    p.q, p.o = q_extend.data_ptr(), o_extend.data_ptr()
    p.q_stride_t, p.q_stride_h = q_extend.stride(0), q_extend.stride(1)
    if k_extend is not None:
        p.k_extend, p.v_extend = k_extend.data_ptr(), v_extend.data_ptr()
        p.k_stride_t, p.k_stride_h = k_extend.stride(0), k_extend.stride(1)
        p.v_stride_t, p.v_stride_h = v_extend.stride(0), v_extend.stride(1)
    p.o_stride_t, p.o_stride_h = o_extend.stride(0), o_extend.stride(1)
    if k_buffer is not None:
        p.kv = kv_layout if kv_layout is not None else _kv_layout(k_buffer, v_buffer, page_size)
    else:
        p.kv.page_size = 1
        skip_prefix = True
    p.qo_indptr = qo_indptr.data_ptr()
    p.qo_indptr_is_i64 = _is64(qo_indptr, "qo_indptr")
    p.kv_indptr = kv_indptr.data_ptr()
    if kv_indices is not None and kv_indices.numel() > 0:
        p.kv_indices = kv_indices.data_ptr()
        p.kv_indices_is_i64 = _is64(kv_indices, "kv_indices")
    else:
        skip_prefix = True  # an empty index list: no request has a cached prefix (its data_ptr is null)
    if lse_extend is not None:
        if lse_extend.dtype != torch.float32:
            raise TypeError("lse_extend must be float32")
        p.lse = lse_extend.data_ptr()
        p.lse_stride_t, p.lse_stride_h = lse_extend.stride(0), lse_extend.stride(1)
    p.bs = qo_indptr.shape[0] - 1
    p.max_extend_len = int(max_len_extend)
    if avg_kv_len_hint is not None:
        p.avg_kv_len_hint = int(avg_kv_len_hint)
    elif kv_indices is not None and p.bs > 0:
        p.avg_kv_len_hint = min(int(kv_indices.numel() // p.bs), 2 ** 31 - 1)
    if unified:
        hnd = kv_layout is not None
        p.num_kv_heads = k_buffer.shape[1] if hnd else k_buffer.shape[-2]
        p.v_head_dim = v_buffer.shape[-1]
        upl = unified_prefix_lens if unified_prefix_lens.dtype == torch.int32 else unified_prefix_lens.to(torch.int32)
        p._keep_unified = upl = upl.contiguous()
        p.unified_prefix_lens = upl.data_ptr()
    else:
        p.num_kv_heads, p.v_head_dim = k_extend.shape[1], v_extend.shape[-1]
        if _num_kv_heads is not None:  # skip_extend callers without new K/V (k_extend / v_extend are not read)
            p.num_kv_heads = int(_num_kv_heads)
    p.num_q_heads, p.head_dim = q_extend.shape[1], q_extend.shape[-1]
    p.sm_scale = sm_scale or 1.0 / (q_extend.shape[-1] ** 0.5)
    p.k_scale, p.v_scale, p.logit_cap = k_scale, v_scale, logit_cap
    p.is_causal, p.skip_prefix, p.skip_extend = int(bool(is_causal)), int(skip_prefix), int(skip_extend)
    p.sliding_window_size = int(sliding_window_size) if sliding_window_size and sliding_window_size > 0 else 0
    if sinks is not None:
        if sinks.dtype != torch.float32:
            sinks = sinks.float()
        p._keep = sinks
        p.sinks = sinks.data_ptr()
    p.dtype = _rx_dtype(q_extend)
    keep = []
    if custom_mask is not None:  # speculative tree mask (bool / uint8 bytes) + int64 offsets
        if mask_indptr is None:
            raise ValueError("custom_mask needs mask_indptr")
        cm = custom_mask if custom_mask.dtype == torch.uint8 else custom_mask.to(torch.uint8)
        mi = mask_indptr if mask_indptr.dtype == torch.int64 else mask_indptr.to(torch.int64)
        cm, mi = cm.contiguous(), mi.contiguous()
        keep += [cm, mi]
        p.custom_mask, p.mask_indptr = cm.data_ptr(), mi.data_ptr()
        p.skip_prefix_custom_mask = int(bool(skip_prefix_custom_mask))
        if window_kv_offsets is not None:
            wo = window_kv_offsets if window_kv_offsets.dtype == torch.int32 else window_kv_offsets.to(torch.int32)
            wo = wo.contiguous()
            keep.append(wo)
            p.window_kv_offsets = wo.data_ptr()
    p.xai_temperature_len = int(xai_temperature_len) if xai_temperature_len and xai_temperature_len > 0 else 0
    p._keep_mask = keep
    p.q_pack = int(q_pack)
    _set_score_bias(p, score_mod, aux_tensors, q_extend)  # extend_attention.py:463-476,594-607,1093-1104
    return p


def extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
                         kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr,
                         max_len_extend, k_scale, v_scale, sm_scale=None, logit_cap=0.0,
                         skip_prefix_custom_mask=True, sliding_window_size=-1, sinks=None,
                         window_kv_offsets=None, xai_temperature_len=-1, lse_extend=None,
                         skip_prefix=False, skip_extend=False, page_size: int = 1,
                         score_mod=None, aux_tensors=None, kv_layout=None, unified_prefix_lens=None,
                         _num_kv_heads=None, avg_kv_len_hint=None, q_pack: int = 1):
    p = _extend_params(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                       custom_mask, is_causal, mask_indptr, max_len_extend, k_scale, v_scale, sm_scale=sm_scale,
                       logit_cap=logit_cap, skip_prefix_custom_mask=skip_prefix_custom_mask,
                       sliding_window_size=sliding_window_size, sinks=sinks, window_kv_offsets=window_kv_offsets,
                       xai_temperature_len=xai_temperature_len, lse_extend=lse_extend, skip_prefix=skip_prefix,
                       skip_extend=skip_extend, page_size=page_size, score_mod=score_mod, aux_tensors=aux_tensors,
                       kv_layout=kv_layout, unified_prefix_lens=unified_prefix_lens, _num_kv_heads=_num_kv_heads,
                       avg_kv_len_hint=avg_kv_len_hint, q_pack=q_pack)
    _L.check(_L.load().rx_extend_attn(C.byref(p), _stream(q_extend)), "rx_extend_attn")


def extend_attention_fwd_gqa_packed(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr,
                                    kv_indptr, kv_indices, custom_mask, is_causal, mask_indptr, max_len_extend,
                                    k_scale, v_scale, **kw):
    """extend_attention_fwd for requests with FEW new tokens (speculative verify, short chunks) under GQA: with
    ``q_pack = G`` (include/radix_hip.h) the kernel runs one workgroup head per KV head whose query rows are (token,
    q head of the group) pairs, so a request's 4-16 new tokens x G heads fill the 32-row query blocks and its K/V
    tiles are staged once per kv head instead of once per q head.  Same arguments, same (bit-identical) result;
    head_dim 128 only, otherwise the ordinary launch."""
    hq, d = q_extend.shape[1], q_extend.shape[2]
    g = hq // k_extend.shape[1]
    if g <= 1 or d != 128 or kw.get("unified_prefix_lens") is not None:
        g = 1
    return extend_attention_fwd(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr,
                                kv_indices, custom_mask, is_causal, mask_indptr, max_len_extend, k_scale, v_scale,
                                q_pack=g, **kw)


def merge_chunks(o_chunks, lse_chunks, o_last, lse_last, out):
    """rx_merge_chunks: o_chunks [groups, S, R, H, D] + lse [groups, S, R, H] (+ o_last [groups, R, H, D], lse_last)
    -> out [groups, R, H, D]."""
    _require_cuda(o_chunks, lse_chunks, o_last, lse_last, out)
    G_, S, R, H, D = o_chunks.shape
    for t in (o_chunks, lse_chunks, out) + ((o_last, lse_last) if o_last is not None else ()):
        if not t.is_contiguous():
            raise ValueError("merge_chunks: tensors must be contiguous")
    st = _L.load().rx_merge_chunks(_ptr(o_chunks), _ptr(lse_chunks), S, _ptr(o_last), _ptr(lse_last), _ptr(out), None,
                                   G_, R, H, D, _rx_dtype(o_chunks), _stream(o_chunks))
    _L.check(st, "rx_merge_chunks")


class VerifySplitKV:
    """Speculative verify (TARGET_VERIFY) for SMALL batches: every request has ``nd`` new tokens under a tree mask
    over a long cached sequence.  One workgroup per (request, kv head) leaves the chip idle (1 request x 32k cached
    tokens: 650 us per layer), so -- as the reference's verify_splitkv does (kernels/ops/attention/verify_splitkv.py)
    -- the cached part is cut into chunks that run as pseudo-requests of ONE GQA-packed extend launch (no mask: every
    draft token sees the whole cache), the draft tokens' own block (tree mask) is a second, tiny launch, and
    rx_merge_chunks joins the partials.

    ``plan`` (once per forward, shared by all layers, no host sync): chunk boundaries inside every request's cached
    list (multiples of 64 tokens) and the row offsets; ``__call__`` (per layer): two re-layout copies, the two
    launches and the merge, with parameter blocks filled once per (layer buffers, batch geometry).
    head_dim 128 (GQA-packed launches), or the latent MLA shape 576 / 512 over one kv head (rx::extend_mla_kernel packs
    the heads itself; a tree mask's own-block launch runs the generic kernel there); no sliding window, Grok
    temperature or sinks (use extend_attention_fwd_gqa_packed)."""

    def __init__(self, num_q_heads: int, num_kv_heads: int, dtype, device, cu_count: int = 256, max_chunks: int = 32,
                 head_dim: int = 128, v_head_dim: Optional[int] = None):
        self.hq, self.hkv, self.g, self.d = num_q_heads, num_kv_heads, max(1, num_q_heads // num_kv_heads), int(head_dim)
        self.dv = int(v_head_dim) if v_head_dim is not None else self.d
        if (self.d, self.dv) != (128, 128) and not ((self.d, self.dv) == (576, 512) and num_kv_heads == 1):
            raise ValueError(f"VerifySplitKV: head dims {self.d} / {self.dv} with {num_kv_heads} kv heads")
        self.pack = self.g if self.d == 128 else 0
        self.dtype, self.device, self.cu_count, self.max_chunks = dtype, device, cu_count, max_chunks
        self._geo = None
        self._params = {}
        self._lib = _L.load()

    def num_chunks(self, bs: int, nd: int = 1) -> int:
        """Chunks per request: enough workgroups (bs * Hkv * query blocks * chunks) to cover every CU."""
        mblocks = -(-(max(1, int(nd)) * self.g) // 128)
        return max(1, min(self.max_chunks, -(-self.cu_count // max(1, bs * self.hkv * mblocks))))

    def plan(self, qo_indptr, kv_indptr, kv_indices, custom_mask, mask_indptr, nd: int) -> None:
        bs = qo_indptr.shape[0] - 1
        S, R, dev = self.num_chunks(bs, nd), int(nd), self.device   # R = new tokens per request
        if self._geo != (bs, S, R):
            self._geo = (bs, S, R)
            n = bs * S * R
            self.q_rep = torch.empty(n, self.hq, self.d, dtype=self.dtype, device=dev)
            self.o_c = torch.empty(n, self.hq, self.dv, dtype=self.dtype, device=dev)
            self.lse_c = torch.empty(n, self.hq, dtype=torch.float32, device=dev)
            self.o_l = torch.empty(bs * R, self.hq, self.dv, dtype=self.dtype, device=dev)
            self.lse_l = torch.empty(bs * R, self.hq, dtype=torch.float32, device=dev)
            self.qo_c = torch.arange(0, (bs * S + 1) * R, R, dtype=torch.int32, device=dev)
            self.qo_g = torch.arange(0, (bs + 1) * R, R, dtype=torch.int32, device=dev)
            self.chunk_indptr = torch.empty(bs * S + 1, dtype=torch.int32, device=dev)
            self._params.clear()
        if kv_indptr.dtype != torch.int32:
            raise TypeError("kv_indptr must be int32")
        _L.check(self._lib.rx_chunk_indptr(_ptr(kv_indptr), bs, S, 64, _ptr(self.chunk_indptr), _stream(kv_indptr)),
                 "rx_chunk_indptr")
        if custom_mask is not None:  # the forms the C ABI takes: uint8 bytes, int64 offsets
            custom_mask = (custom_mask if custom_mask.dtype == torch.uint8 else custom_mask.to(torch.uint8)).contiguous()
            mask_indptr = (mask_indptr if mask_indptr.dtype == torch.int64 else mask_indptr.to(torch.int64)).contiguous()
        self._tabs = (kv_indptr, kv_indices, custom_mask, mask_indptr)
        self.nd = int(nd)

    def __call__(self, q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, k_scale, v_scale, sm_scale=None,
                 logit_cap=0.0, page_size: int = 1, kv_layout=None) -> None:
        bs, S, R = self._geo
        nd, g, hkv, hq, d = self.nd, self.g, self.hkv, self.hq, self.d
        kv_indptr, kv_indices, custom_mask, mask_indptr = self._tabs
        if q_extend.shape != (bs * nd, hq, d) or not q_extend.is_contiguous() or not o_extend.is_contiguous():
            raise ValueError(f"VerifySplitKV: q / o must be contiguous {(bs * nd, hq, d)}")
        # parameter blocks per (layer buffers, scalars); everything that changes per forward or per layer -- the
        # activations q / k / v, the index lists, the mask -- is patched in
        key = (k_buffer.data_ptr(), v_buffer.data_ptr(), float(sm_scale or 0.0), float(k_scale), float(v_scale),
               float(logit_cap), page_size, custom_mask is None)
        ent = self._params.get(key)
        if ent is None:
            if len(self._params) > 1024:
                self._params.clear()
            pc = _extend_params(self.q_rep, self.q_rep, self.q_rep[..., :self.dv], self.o_c, k_buffer, v_buffer, self.qo_c,
                                self.chunk_indptr, kv_indices, None, False, None, R, k_scale, v_scale,
                                sm_scale=sm_scale, logit_cap=logit_cap, lse_extend=self.lse_c, skip_extend=True,
                                page_size=page_size, kv_layout=kv_layout, _num_kv_heads=hkv, q_pack=self.pack,
                                avg_kv_len_hint=0)
            pl = _extend_params(q_extend, k_extend, v_extend, self.o_l, k_buffer, v_buffer, self.qo_g, kv_indptr,
                                kv_indices, custom_mask, True, mask_indptr, R, k_scale, v_scale, sm_scale=sm_scale,
                                logit_cap=logit_cap, lse_extend=self.lse_l, skip_prefix=True, page_size=page_size,
                                kv_layout=kv_layout, q_pack=self.pack, avg_kv_len_hint=0)
            ent = self._params[key] = (pc, C.byref(pc), pl, C.byref(pl), (k_buffer, v_buffer))
        pc, pc_ref, pl, pl_ref, _ = ent
        i64 = _is64(kv_indices, "kv_indices")
        pc.kv_indices, pc.kv_indices_is_i64 = kv_indices.data_ptr(), i64
        pl.kv_indices, pl.kv_indices_is_i64, pl.kv_indptr = kv_indices.data_ptr(), i64, kv_indptr.data_ptr()
        pl.q, pl.q_stride_t, pl.q_stride_h = q_extend.data_ptr(), q_extend.stride(0), q_extend.stride(1)
        pl.k_extend, pl.k_stride_t, pl.k_stride_h = k_extend.data_ptr(), k_extend.stride(0), k_extend.stride(1)
        pl.v_extend, pl.v_stride_t, pl.v_stride_h = v_extend.data_ptr(), v_extend.stride(0), v_extend.stride(1)
        if custom_mask is not None:
            pl.custom_mask, pl.mask_indptr = custom_mask.data_ptr(), mask_indptr.data_ptr()
        stream = _stream(q_extend)
        self.q_rep.view(bs, S, R, hq, d).copy_(q_extend.view(bs, 1, R, hq, d).expand(bs, S, R, hq, d))
        lib = self._lib
        for ref in (pc_ref, pl_ref):
            st = lib.rx_extend_attn(ref, stream)
            if st:
                _L.check(st, "rx_extend_attn")
        st = lib.rx_merge_chunks(_ptr(self.o_c), _ptr(self.lse_c), S, _ptr(self.o_l), _ptr(self.lse_l), _ptr(o_extend),
                                 None, bs, R, hq, self.dv, _rx_dtype(q_extend), stream)
        if st:
            _L.check(st, "rx_merge_chunks")


def verify_attention_splitkv(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr,
                             kv_indices, custom_mask, mask_indptr, num_draft_tokens: int, num_chunks: int, k_scale,
                             v_scale, sm_scale=None, logit_cap=0.0, page_size: int = 1, kv_layout=None):
    """One-shot form of VerifySplitKV (plan + call) with an explicit chunk count; see the class."""
    vs = VerifySplitKV(q_extend.shape[1], k_extend.shape[1], q_extend.dtype, q_extend.device, max_chunks=num_chunks,
                       cu_count=1 << 30, head_dim=q_extend.shape[2], v_head_dim=o_extend.shape[2])
    vs.plan(qo_indptr, kv_indptr, kv_indices, custom_mask, mask_indptr, num_draft_tokens)
    vs(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, k_scale, v_scale, sm_scale=sm_scale,
       logit_cap=logit_cap, page_size=page_size, kv_layout=kv_layout)


def extend_attention_fwd_unified(q, o, k_buffer, v_buffer, k_scale, v_scale, qo_indptr, kv_indptr, kv_indices,
                                 prefix_lens, max_len_extend, custom_mask=None, mask_indptr=None, sm_scale=None,
                                 logit_cap=0.0, is_causal=True, sliding_window_size=-1, sinks=None,
                                 window_start_pos=None, xai_temperature_len=-1, page_size: int = 1,
                                 score_mod=None, aux_tensors=None, kv_layout=None):
    """K8: the one-stage extend of deterministic inference (extend_attention.py:1160-1300).  kv_indptr /
    kv_indices list prefix AND new tokens (their K/V are already in the pool); prefix_lens int32[bs].
    window_start_pos only shifts both sides of the window test and is accepted for signature parity."""
    extend_attention_fwd(q, None, None, o, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, custom_mask,
                         is_causal, mask_indptr, max_len_extend, k_scale, v_scale, sm_scale=sm_scale,
                         logit_cap=logit_cap, skip_prefix_custom_mask=False,
                         sliding_window_size=sliding_window_size, sinks=sinks,
                         xai_temperature_len=xai_temperature_len, page_size=page_size, score_mod=score_mod,
                         aux_tensors=aux_tensors, kv_layout=kv_layout, unified_prefix_lens=prefix_lens)


# --------------------------------------------------------------------------------------
# fused QK-norm + RoPE (+ KV store)     fused_qk_norm_rope, kernels/ops/attention/fused_qknorm_rope.py:127-186
# --------------------------------------------------------------------------------------
def fused_qk_norm_rope(qkv: torch.Tensor, num_heads_q: int, num_heads_k: int, num_heads_v: int, head_dim: int, eps: float,
                       q_weight: torch.Tensor, k_weight: torch.Tensor, base: float, is_neox: bool,
                       position_ids: torch.Tensor, factor: float = 1.0, low: float = 0.0, high: float = 0.0,
                       attention_factor: float = 1.0, rotary_dim: Optional[int] = None, *, cos_sin_cache=None,
                       layout: Optional["_L.RxKvLayout"] = None, loc=None, size_limit: int = 0, k_scale: float = 1.0,
                       v_scale: float = 1.0, reserved_skip_index: int = 0, err_flag=None) -> None:
    """Per-head RMSNorm of q and k + RoPE, IN PLACE on qkv [num_tokens, (nq + nk + nv) * head_dim] -- the reference's name,
    argument order and meaning (fused_qknorm_rope.py:127-186; kernel jit/csrc/elementwise/fused_qknorm_rope.cuh): frequencies
    on the fly from ``base`` (YaRN blend when factor != 1), the rotated part times attention_factor, v untouched.
    Beyond the reference: fp16 as well as bf16, int32 or int64 positions, any even head_dim <= 512, frequencies from a
    ``cos_sin_cache`` (fp32 [max_pos, rotary_dim]) instead, and -- with ``layout`` + ``loc`` -- the finished k rows and the
    v rows written to the paged pool in the same launch (as rope_store_kv)."""
    _require_cuda(qkv, q_weight, k_weight, position_ids, cos_sin_cache, loc)
    nq, nk, nv, d = int(num_heads_q), int(num_heads_k), int(num_heads_v), int(head_dim)
    if qkv.dim() != 2 or qkv.shape[1] != (nq + nk + nv) * d or qkv.stride(1) != 1:
        raise ValueError(f"fused_qk_norm_rope: qkv must be [num_tokens, {(nq + nk + nv) * d}] with unit inner stride, "
                         f"got {tuple(qkv.shape)} / strides {qkv.stride()}")
    if qkv.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError("fused_qk_norm_rope: qkv must be bfloat16 or float16")
    for nm, w in (("q_weight", q_weight), ("k_weight", k_weight)):
        if w.dtype != qkv.dtype or w.numel() != d or not w.is_contiguous():
            raise ValueError(f"fused_qk_norm_rope: {nm} must be a contiguous [{d}] tensor of qkv's dtype")
    if position_ids.dtype not in (torch.int32, torch.int64) or position_ids.numel() != qkv.shape[0]:
        raise ValueError("fused_qk_norm_rope: position_ids must be int32 / int64 [num_tokens]")
    rot = d if rotary_dim is None else int(rotary_dim)
    n = qkv.shape[0]
    st = qkv.stride(0)
    qp = qkv.data_ptr()
    es = qkv.element_size()
    kp, vp = qp + nq * d * es, qp + (nq + nk) * d * es
    cs_ptr, cs_stride = 0, 0
    if cos_sin_cache is not None:
        if cos_sin_cache.dtype != torch.float32 or cos_sin_cache.stride(-1) != 1 or cos_sin_cache.shape[-1] != rot:
            raise TypeError("fused_qk_norm_rope: cos_sin_cache must be float32 [max_pos, rotary_dim]")
        cs_ptr, cs_stride = cos_sin_cache.data_ptr(), cos_sin_cache.stride(0)
    lay_ref, locp, l64 = None, None, 0
    if layout is not None:
        if loc is None or nv != nk:
            raise ValueError("fused_qk_norm_rope: the pool store needs loc and as many v heads as k heads")
        loc_c = loc.contiguous()  # (kept alive in a local until the launch is queued: the kernel reads it asynchronously)
        lay_ref, locp, l64 = C.byref(layout), _ptr(loc_c), _is64(loc, "loc")
    pos = position_ids.contiguous()
    stt = _L.load().rx_qknorm_rope_store_kv(
        qp, kp, vp, st, d, st, d, st, d, n, nq, nk, d, d if layout is not None else 0, rot, _ptr(q_weight), _ptr(k_weight),
        float(eps), _ptr(pos), int(pos.dtype == torch.int64), float(base), float(factor), float(low), float(high),
        float(attention_factor), cs_ptr, cs_stride, int(bool(is_neox)), lay_ref, locp, l64, int(size_limit),
        int(reserved_skip_index), float(k_scale), float(v_scale), _rx_dtype(qkv), _ptr(err_flag), _stream(qkv))
    _L.check(stt, "rx_qknorm_rope_store_kv")


# --------------------------------------------------------------------------------------
# fused RoPE + KV store      RotaryEmbedding.forward + set_kv_buffer; kernels/ops/kvcache/rope_cache.py
# --------------------------------------------------------------------------------------
def rope_store_kv(q, k, v, positions, cos_sin_cache, is_neox: bool, *, rotary_dim: Optional[int] = None,
                  layout: Optional["_L.RxKvLayout"] = None, loc=None, size_limit: int = 0,
                  k_scale: float = 1.0, v_scale: float = 1.0, reserved_skip_index: int = 0, err_flag=None):
    """Rotate q [n,Hq,D] and k [n,Hkv,D] in place; with ``layout`` + ``loc`` also write the rotated k and
    v [n,Hkv,Dv] into the pool (16-bit or fp8) in the same launch.  cos_sin_cache: fp32 [max_pos, rot]."""
    _require_cuda(q, k, v, positions, cos_sin_cache, loc)
    if q.dim() != 3 or k.dim() != 3 or q.stride(-1) != 1 or k.stride(-1) != 1:
        raise ValueError("rope_store_kv: q / k must be [n, heads, dim], contiguous in dim")
    if cos_sin_cache.dtype != torch.float32 or cos_sin_cache.stride(-1) != 1:
        raise TypeError("rope_store_kv: cos_sin_cache must be float32 [max_pos, rotary_dim]")
    pos = positions if positions.dtype == torch.int64 else positions.to(torch.int64)
    rot = int(rotary_dim or cos_sin_cache.shape[-1])
    lay_ref, locp, l64 = None, None, 0
    dv = 0
    vs_t = vs_h = 0
    if layout is not None:
        if v is None or loc is None:
            raise ValueError("rope_store_kv: the pool store needs v and loc")
        if v.dim() != 3 or v.stride(-1) != 1:
            raise ValueError("rope_store_kv: v must be [n, Hkv, Dv], contiguous in Dv")
        loc_c = loc.contiguous()  # (kept alive in a local until the launch is queued)
        lay_ref, locp, l64 = C.byref(layout), _ptr(loc_c), _is64(loc, "loc")
        dv, vs_t, vs_h = v.shape[-1], v.stride(0), v.stride(1)
    st = _L.load().rx_rope_store_kv(
        _ptr(q), _ptr(k), _ptr(v), q.stride(0), q.stride(1), k.stride(0), k.stride(1), vs_t, vs_h, q.shape[0],
        q.shape[1], k.shape[1], q.shape[2], dv, rot, _ptr(pos), _ptr(cos_sin_cache), cos_sin_cache.stride(0),
        int(bool(is_neox)), lay_ref, locp, l64, size_limit, reserved_skip_index, float(k_scale), float(v_scale),
        _rx_dtype(q), _ptr(err_flag), _stream(q))
    _L.check(st, "rx_rope_store_kv")


# --------------------------------------------------------------------------------------
# merge_state                kernels/ops/attention/merge_state.py:66-96
# --------------------------------------------------------------------------------------
def merge_state(prefix_output: torch.Tensor, prefix_lse: torch.Tensor, suffix_output: torch.Tensor,
                suffix_lse: torch.Tensor, output: Optional[torch.Tensor] = None,
                output_lse: Optional[torch.Tensor] = None):
    """Same contract as merge_state_triton: returns (output, output_lse); buffers are created when not
    given.  [T, H, D] 16-bit outputs, fp32 [T, H] LSEs."""
    _require_cuda(prefix_output, prefix_lse, suffix_output, suffix_lse, output, output_lse)
    if prefix_output.shape != suffix_output.shape or prefix_output.dim() != 3:
        raise ValueError("merge_state: outputs must both be [num_tokens, num_heads, head_size]")
    if prefix_output.dtype != suffix_output.dtype:
        raise TypeError("merge_state: output dtypes differ")
    for n, t in (("prefix_lse", prefix_lse), ("suffix_lse", suffix_lse)):
        if t.dtype != torch.float32 or tuple(t.shape) != tuple(prefix_output.shape[:2]):
            raise TypeError(f"merge_state: {n} must be float32 [num_tokens, num_heads]")
    a, b = prefix_output.contiguous(), suffix_output.contiguous()
    la, lb = prefix_lse.contiguous(), suffix_lse.contiguous()
    if output is None:
        output = torch.empty_like(a)
    if output_lse is None:
        output_lse = torch.empty_like(la)
    if not output.is_contiguous() or not output_lse.is_contiguous():
        raise ValueError("merge_state: output buffers must be contiguous")
    T, H, D = a.shape
    st = _L.load().rx_merge_state(_ptr(a), _ptr(la), _ptr(b), _ptr(lb), _ptr(output), _ptr(output_lse), T, H, D,
                                  _rx_dtype(a), _stream(a))
    _L.check(st, "rx_merge_state")
    return output, output_lse


# --------------------------------------------------------------------------------------
# K9  paged allocation kernels   kernels/ops/memory/allocator.py:16-135
# --------------------------------------------------------------------------------------
def alloc_extend(prefix_lens, seq_lens, last_loc, free_pages, out_indices, page_size: int):
    _require_cuda(prefix_lens, seq_lens, last_loc, free_pages, out_indices)
    for n, t in (("prefix_lens", prefix_lens), ("seq_lens", seq_lens), ("last_loc", last_loc),
                 ("free_pages", free_pages), ("out_indices", out_indices)):
        if t.dtype != torch.int64:
            raise TypeError(f"{n} must be int64")
    st = _L.load().rx_alloc_extend(_ptr(prefix_lens), _ptr(seq_lens), _ptr(last_loc),
                                   _ptr(free_pages), _ptr(out_indices), prefix_lens.shape[0],
                                   page_size, _stream(seq_lens))
    _L.check(st, "rx_alloc_extend")


def alloc_decode(seq_lens, last_loc, free_pages, out_indices, page_size: int):
    _require_cuda(seq_lens, last_loc, free_pages, out_indices)
    for n, t in (("seq_lens", seq_lens), ("last_loc", last_loc), ("free_pages", free_pages),
                 ("out_indices", out_indices)):
        if t.dtype != torch.int64:
            raise TypeError(f"{n} must be int64")
    st = _L.load().rx_alloc_decode(_ptr(seq_lens), _ptr(last_loc), _ptr(free_pages),
                                   _ptr(out_indices), seq_lens.shape[0], page_size,
                                   _stream(seq_lens))
    _L.check(st, "rx_alloc_decode")


# --------------------------------------------------------------------------------------
# K11 write_req_to_token_pool    called srt/mem_cache/allocation.py:75-84
# --------------------------------------------------------------------------------------
def write_req_to_token(req_to_token, req_pool_indices, prefix_ptrs, pre_lens, seq_lens,
                       extend_lens, out_cache_loc):
    _require_cuda(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc)
    st = _L.load().rx_write_req_to_token(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices), _ptr(prefix_ptrs),
        _ptr(pre_lens), _ptr(seq_lens), _ptr(extend_lens), _ptr(out_cache_loc),
        req_pool_indices.shape[0], _stream(req_to_token))
    _L.check(st, "rx_write_req_to_token")


# --------------------------------------------------------------------------------------
# K10 move_kv_cache              kernels/ops/kvcache/cache_move.py:60-133
# --------------------------------------------------------------------------------------
def move_kv(data_ptrs: torch.Tensor, row_bytes: torch.Tensor, tgt_loc: torch.Tensor,
            src_loc: torch.Tensor):
    _require_cuda(data_ptrs, row_bytes, tgt_loc, src_loc)
    if data_ptrs.dtype not in (torch.uint64, torch.int64) or row_bytes.dtype != torch.int64:
        raise TypeError("data_ptrs must be (u)int64 and row_bytes int64")
    if tgt_loc.dtype != torch.int64 or src_loc.dtype != torch.int64:
        raise TypeError("tgt_loc / src_loc must be int64")
    st = _L.load().rx_move_kv(_ptr(data_ptrs), _ptr(row_bytes), data_ptrs.shape[0], _ptr(tgt_loc),
                              _ptr(src_loc), tgt_loc.shape[0], _stream(tgt_loc))
    _L.check(st, "rx_move_kv")


def move_kv_layout(data_ptrs: torch.Tensor, geom: torch.Tensor, page_size: int, num_heads: int,
                   tgt_loc: torch.Tensor, src_loc: torch.Tensor):
    """rx_move_kv_layout: move_kv_cache on a paged (HND) pool; geom int64[num_bufs, 4] =
    {page_stride, head_stride, tok_stride, piece_bytes} in bytes per buffer."""
    _require_cuda(data_ptrs, geom, tgt_loc, src_loc)
    if data_ptrs.dtype not in (torch.uint64, torch.int64) or geom.dtype != torch.int64:
        raise TypeError("data_ptrs must be (u)int64 and geom int64")
    if tuple(geom.shape) != (data_ptrs.shape[0], 4) or not geom.is_contiguous():
        raise ValueError("geom must be contiguous [num_bufs, 4]")
    if tgt_loc.dtype != torch.int64 or src_loc.dtype != torch.int64:
        raise TypeError("tgt_loc / src_loc must be int64")
    st = _L.load().rx_move_kv_layout(_ptr(data_ptrs), _ptr(geom), data_ptrs.shape[0], int(page_size), int(num_heads),
                                     _ptr(tgt_loc), _ptr(src_loc), tgt_loc.shape[0], _stream(tgt_loc))
    _L.check(st, "rx_move_kv_layout")


# --------------------------------------------------------------------------------------
# decode context parallel (DCP)   srt/layers/dcp/{layout,comm}.py, kernels/ops/attention/dcp_kernels.py
# --------------------------------------------------------------------------------------
def dcp_kv_indices(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor, lens: torch.Tensor,
                   kv_indptr: torch.Tensor, kv_indices: Optional[torch.Tensor], dcp_size: int, dcp_rank: int,
                   kv_start: Optional[torch.Tensor] = None, dcp_lens: Optional[torch.Tensor] = None):
    """TritonAttnBackend._dcp_kv_indices (triton_backend.py:356-384) in one call: this rank's share of every request
    (owner rule position % dcp_size == dcp_rank) as kv_indptr[:bs+1] (int32), its LOCAL slots (virtual slot //
    dcp_size) in kv_indices when given, and the per-request counts in dcp_lens (int32[bs]) when given."""
    _require_cuda(req_to_token, req_pool_indices, lens, kv_indptr, kv_indices, kv_start, dcp_lens)
    bs = lens.shape[0]
    if req_to_token.dtype != torch.int32 or kv_indptr.dtype != torch.int32:
        raise TypeError("req_to_token and kv_indptr must be int32")
    if kv_indptr.numel() < bs + 1:
        raise ValueError("kv_indptr too small")
    if dcp_lens is not None and (dcp_lens.dtype != torch.int32 or dcp_lens.numel() < bs):
        raise TypeError("dcp_lens must be int32[bs]")
    if kv_start is not None and kv_start.dtype != torch.int32:
        kv_start = kv_start.to(torch.int32)
    st = _L.load().rx_dcp_kv_indices(
        _ptr(req_to_token), req_to_token.stride(0), _ptr(req_pool_indices),
        _is64(req_pool_indices, "req_pool_indices"), _ptr(lens), _is64(lens, "lens"), _ptr(kv_start), int(dcp_size),
        int(dcp_rank), _ptr(kv_indptr), _ptr(kv_indices),
        0 if kv_indices is None else _is64(kv_indices, "kv_indices"), _ptr(dcp_lens), bs, _stream(req_to_token))
    _L.check(st, "rx_dcp_kv_indices")
    return kv_indptr[: bs + 1]


def dcp_store_loc(out_cache_loc: torch.Tensor, positions: torch.Tensor, dcp_size: int, dcp_rank: int,
                  skip_index: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Local write slots of the new tokens: out_cache_loc // dcp_size for the tokens this rank owns
    (positions % dcp_size == dcp_rank), ``skip_index`` (left unwritten by the store kernels) for the rest."""
    _require_cuda(out_cache_loc, positions, out)
    n = out_cache_loc.numel()
    if positions.numel() != n:
        raise ValueError("positions and out_cache_loc differ in length")
    if out is None:
        out = torch.empty(n, dtype=torch.int64, device=out_cache_loc.device)
    elif out.dtype != torch.int64 or out.numel() < n:
        raise TypeError("out must be int64[n]")
    st = _L.load().rx_dcp_store_loc(_ptr(out_cache_loc), _is64(out_cache_loc, "out_cache_loc"), _ptr(positions),
                                    _is64(positions, "positions"), n, int(dcp_size), int(dcp_rank), int(skip_index),
                                    _ptr(out), _stream(out_cache_loc))
    _L.check(st, "rx_dcp_store_loc")
    return out[:n]


def dcp_local_merge(attn_logits: torch.Tensor, attn_lse: torch.Tensor, o32: Optional[torch.Tensor] = None,
                    lse: Optional[torch.Tensor] = None, v_scale: float = 1.0):
    """kv-split partials [bs, H, S, Dv] / [bs, H, S] (dead splits -inf) -> (fp32 [bs, H, Dv], natural-log LSE [bs, H])."""
    _require_cuda(attn_logits, attn_lse, o32, lse)
    if attn_logits.dtype != torch.float32 or attn_lse.dtype != torch.float32 or attn_logits.dim() != 4:
        raise TypeError("dcp_local_merge: fp32 [bs, H, S, Dv] partials and [bs, H, S] LSEs")
    if not attn_logits.is_contiguous() or not attn_lse.is_contiguous():
        raise ValueError("dcp_local_merge: partials must be contiguous")
    bs, H, S, Dv = attn_logits.shape
    if o32 is None:
        o32 = torch.empty(bs, H, Dv, dtype=torch.float32, device=attn_logits.device)
    if lse is None:
        lse = torch.empty(bs, H, dtype=torch.float32, device=attn_logits.device)
    st = _L.load().rx_dcp_local_merge(_ptr(attn_logits), _ptr(attn_lse), bs * H, S, Dv, float(v_scale), _ptr(o32), _ptr(lse),
                                      _stream(attn_logits))
    _L.check(st, "rx_dcp_local_merge")
    return o32, lse


def dcp_widen(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """16-bit partial -> fp32 (same shape)."""
    _require_cuda(x, out)
    if not x.is_contiguous():
        raise ValueError("dcp_widen: contiguous input")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _L.check(_L.load().rx_dcp_widen(_ptr(x), x.numel(), _rx_dtype(x), _ptr(out), _stream(x)), "rx_dcp_widen")
    return out


def dcp_scale(o32: torch.Tensor, lses_all: torch.Tensor, dcp_rank: int, global_lse: Optional[torch.Tensor] = None):
    """In place: o32 [T, H, Dv] *= exp(lses_all[rank] - logsumexp over ranks); lses_all fp32 [dcp, T, H] (all-gathered)."""
    _require_cuda(o32, lses_all, global_lse)
    if o32.dtype != torch.float32 or lses_all.dtype != torch.float32 or not o32.is_contiguous() or not lses_all.is_contiguous():
        raise TypeError("dcp_scale: contiguous fp32 tensors")
    dcp, rows = lses_all.shape[0], o32.shape[0] * o32.shape[1]
    if lses_all.numel() != dcp * rows:
        raise ValueError("dcp_scale: lses_all must be [dcp, T, H]")
    st = _L.load().rx_dcp_scale(_ptr(o32), _ptr(lses_all), rows, dcp, int(dcp_rank), o32.shape[2], _ptr(global_lse),
                                _stream(o32))
    _L.check(st, "rx_dcp_scale")
    return o32


def dcp_finish(o32: torch.Tensor, out: torch.Tensor, head_start: int, global_lse: Optional[torch.Tensor] = None,
               cur_o: Optional[torch.Tensor] = None, cur_lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """This rank's heads of the all-reduced fp32 [T, H_all, Dv] -> out [T, H_loc, Dv] (16-bit), joined with the
    extend path's own-chunk partial (cur_o 16-bit [T, H_loc, Dv], cur_lse fp32 [T, H_loc]) when given."""
    _require_cuda(o32, out, global_lse, cur_o, cur_lse)
    T, Hall, Dv = o32.shape
    Hloc = out.shape[1]
    if tuple(out.shape) != (T, Hloc, Dv) or not out.is_contiguous() or not o32.is_contiguous():
        raise ValueError("dcp_finish: out must be contiguous [T, H_loc, Dv]")
    if cur_o is not None and (tuple(cur_o.shape) != tuple(out.shape) or cur_o.dtype != out.dtype or not cur_o.is_contiguous()
                              or not cur_lse.is_contiguous() or not global_lse.is_contiguous()):
        raise ValueError("dcp_finish: cur_o must match out; LSEs contiguous")
    st = _L.load().rx_dcp_finish(_ptr(o32), _ptr(global_lse), _ptr(cur_o), _ptr(cur_lse), _ptr(out), T, Hall,
                                 int(head_start), Hloc, Dv, _rx_dtype(out), _stream(o32))
    _L.check(st, "rx_dcp_finish")
    return out
