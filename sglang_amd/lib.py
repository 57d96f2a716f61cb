"""ctypes binding of libradix_hip.so (include/radix_hip.h).

The product path has NO fallback: if the shared object is missing and cannot be built, or a
call returns a negative status, this module raises.  Nothing under ``oracle/`` is imported.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_int32, c_int64, c_float = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float

RX_ABI_VERSION = 16  # include/radix_hip.h
RX_BF16, RX_F16 = 0, 1
RX_DEVERR_SLOT_OOB = 1


class RxKvLayout(C.Structure):
    _fields_ = [
        ("k_buf", c_void_p), ("v_buf", c_void_p), ("page_size", c_int32),
        ("k_page_stride", c_int64), ("k_tok_stride", c_int64), ("k_head_stride", c_int64),
        ("v_page_stride", c_int64), ("v_tok_stride", c_int64), ("v_head_stride", c_int64),
        ("kv_fp8", c_int32),
    ]


class RxPoolDesc(C.Structure):
    _fields_ = [("free_ring", c_void_p), ("release_ring", c_void_p), ("capacity", c_int64), ("flags", c_void_p),
                ("num_ids", c_int64), ("tile_scratch", c_void_p), ("state", c_void_p)]


class RxDecodeParams(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("o", c_void_p),
        ("q_stride_t", c_int64), ("q_stride_h", c_int64), ("o_stride_t", c_int64), ("o_stride_h", c_int64),
        ("kv", RxKvLayout),
        ("kv_indptr", c_void_p), ("kv_indices", c_void_p), ("kv_indices_is_i64", c_int32),
        ("req_to_token", c_void_p), ("req_row_stride", c_int64),
        ("req_pool_indices", c_void_p), ("req_pool_indices_is_i64", c_int32),
        ("seq_lens", c_void_p), ("seq_lens_is_i64", c_int32),
        ("num_kv_splits", c_void_p), ("max_kv_splits", c_int32),
        ("attn_logits", c_void_p), ("attn_lse", c_void_p),
        ("bs", c_int32), ("num_q_heads", c_int32), ("num_kv_heads", c_int32),
        ("head_dim", c_int32), ("v_head_dim", c_int32),
        ("sm_scale", c_float), ("k_scale", c_float), ("v_scale", c_float), ("logit_cap", c_float),
        ("sinks", c_void_p), ("dtype", c_int32),
        ("xai_temperature_len", c_int32),
        ("kv_start", c_void_p), ("extra_o", c_void_p), ("extra_lse", c_void_p), ("num_extra_partials", c_int32),
        ("extra_index", c_void_p), ("extra_rows", c_int32),
        ("stages", c_int32), ("merge_counters", c_void_p),
        ("k_new", c_void_p), ("v_new", c_void_p),
        ("k_new_stride_t", c_int64), ("k_new_stride_h", c_int64), ("v_new_stride_t", c_int64), ("v_new_stride_h", c_int64),
        ("request_order", c_void_p), ("partial_pairs_hint", c_int32),
        ("split_items", c_void_p), ("split_items_count", c_void_p), ("split_items_cap", c_int32),
        ("split_items_wgs_per_cu", c_int32),
        ("rope_cos_sin", c_void_p), ("rope_cos_sin_is_f32", c_int32), ("rope_cos_sin_stride", c_int64),
        ("rope_positions", c_void_p), ("rope_positions_is_i64", c_int32), ("rope_dim", c_int32),
        ("rope_is_neox", c_int32), ("rope_k_pe_out", c_void_p), ("rope_k_pe_out_stride", c_int64),
        ("score_bias", c_void_p), ("score_bias_is_f32", c_int32), ("score_bias_len", c_int32),
        ("score_bias_stride_t", c_int64), ("score_bias_stride_h", c_int64),
        ("unit_desc", c_void_p), ("unit_first_slots", c_void_p),
    ]


class RxExtendParams(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k_extend", c_void_p), ("v_extend", c_void_p), ("o", c_void_p),
        ("q_stride_t", c_int64), ("q_stride_h", c_int64), ("k_stride_t", c_int64), ("k_stride_h", c_int64),
        ("v_stride_t", c_int64), ("v_stride_h", c_int64), ("o_stride_t", c_int64), ("o_stride_h", c_int64),
        ("kv", RxKvLayout),
        ("qo_indptr", c_void_p), ("qo_indptr_is_i64", c_int32),
        ("kv_indptr", c_void_p), ("kv_indices", c_void_p), ("kv_indices_is_i64", c_int32),
        ("lse", c_void_p), ("lse_stride_t", c_int64), ("lse_stride_h", c_int64),
        ("bs", c_int32), ("max_extend_len", c_int32), ("num_q_heads", c_int32), ("num_kv_heads", c_int32),
        ("head_dim", c_int32), ("v_head_dim", c_int32),
        ("sm_scale", c_float), ("k_scale", c_float), ("v_scale", c_float), ("logit_cap", c_float),
        ("is_causal", c_int32), ("skip_prefix", c_int32), ("skip_extend", c_int32),
        ("sliding_window_size", c_int32),
        ("sinks", c_void_p), ("dtype", c_int32),
        ("custom_mask", c_void_p), ("mask_indptr", c_void_p), ("skip_prefix_custom_mask", c_int32),
        ("window_kv_offsets", c_void_p), ("xai_temperature_len", c_int32),
        ("unified_prefix_lens", c_void_p), ("avg_kv_len_hint", c_int32), ("q_pack", c_int32),
        ("score_bias", c_void_p), ("score_bias_is_f32", c_int32), ("score_bias_len", c_int32),
        ("score_bias_stride_t", c_int64), ("score_bias_stride_h", c_int64),
    ]


# symbol -> (restype, argtypes); every prototype of include/radix_hip.h
PROTOTYPES = {
    "rx_version": (c_int, []),
    "rx_abi_sizeof": (c_int64, [c_int]),
    "rx_last_error": (C.c_char_p, []),
    "rx_last_dispatch": (C.c_char_p, []),
    "rx_set_option": (c_int, [C.c_char_p, c_int]),
    "rx_get_option": (c_int, [C.c_char_p, C.POINTER(c_int)]),
    "rx_store_kv": (c_int, [c_void_p] * 5 + [c_int64] * 7 + [c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "rx_store_kv_layout": (c_int, [c_void_p, c_void_p, C.POINTER(RxKvLayout), c_void_p, c_int64, c_int, c_int,
                                   c_int, c_int64, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "rx_ar_region_bytes": (c_int64, [c_int64]),
    "rx_ar_alloc_region": (c_int, [c_int64, C.POINTER(c_void_p)]),
    "rx_ar_free_region": (c_int, [c_void_p]),
    "rx_ipc_get_handle": (c_int, [c_void_p, c_void_p]),
    "rx_ipc_open_handle": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "rx_ipc_close_handle": (c_int, [c_void_p]),
    "rx_ar_init": (c_int, [C.POINTER(c_void_p), c_int, c_int, C.POINTER(c_void_p), c_int64, c_void_p]),
    "rx_allreduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rx_allreduce_det": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rx_allreduce_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                     c_float, c_int, c_void_p]),
    "rx_ar_destroy": (c_int, [c_void_p]),
    "rx_qr_region_bytes": (c_int64, []),
    "rx_qr_init": (c_int, [C.POINTER(c_void_p), c_int, c_int, C.POINTER(c_void_p), c_void_p]),
    "rx_quick_allreduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "rx_qr_destroy": (c_int, [c_void_p]),
    "rx_rcp_f16_table": (c_int, [c_void_p, c_void_p]),
    "rx_rope_store_kv": (c_int, [c_void_p] * 3 + [c_int64] * 7 + [c_int] * 5 + [c_void_p, c_void_p, c_int64, c_int,
                                 C.POINTER(RxKvLayout), c_void_p, c_int, c_int64, c_int64, c_float, c_float, c_int,
                                 c_void_p, c_void_p]),
    "rx_qknorm_rope_store_kv": (c_int, [c_void_p] * 3 + [c_int64] * 7 + [c_int] * 5 + [c_void_p, c_void_p, c_float,
                                        c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p, c_int64,
                                        c_int, C.POINTER(RxKvLayout), c_void_p, c_int, c_int64, c_int64, c_float,
                                        c_float, c_int, c_void_p, c_void_p]),
    "rx_chunk_indptr": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rx_merge_chunks": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                c_int, c_int, c_void_p]),
    "rx_shared_prefix_plan": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_int, c_int32, c_int32,
                                      c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rx_num_kv_splits_native": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rx_split_items": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "rx_debug_counters": (c_int, [c_void_p, c_int]),
    "rx_clock_probe": (c_int, [c_void_p, c_int32, c_void_p]),
    "rx_draft_decode_kv_indices": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                           c_int, c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p]),
    "rx_split_items_guarded": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rx_num_kv_splits_balanced": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rx_merge_state": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "rx_get_mla_kv": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_void_p,
                              c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "rx_store_kv_fp8": (c_int, [c_void_p, c_void_p, C.POINTER(RxKvLayout), c_void_p, c_int64, c_int, c_int, c_int,
                                c_int64, c_int64, c_int, C.c_float, C.c_float, c_int, c_int64, c_int64,
                                c_void_p, c_void_p]),
    "rx_fused_fp8_qkv_kv_cache": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.POINTER(RxKvLayout), c_void_p, c_int,
                                          c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                          c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "rx_build_kv_indices": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rx_build_unified_kv_indices": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                            c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rx_num_kv_splits": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p]),
    "rx_decode_attn": (c_int, [C.POINTER(RxDecodeParams), c_void_p]),
    "rx_decode_units": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rx_extend_attn": (c_int, [C.POINTER(RxExtendParams), c_void_p]),
    "rx_alloc_extend": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "rx_alloc_decode": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "rx_write_req_to_token": (c_int, [c_void_p, c_int64] + [c_void_p] * 6 + [c_int, c_void_p]),
    "rx_move_kv": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    # device-resident allocator free list
    "rx_pool_tile_scratch_len": (c_int64, [c_int64]),
    "rx_pool_state_words": (c_int, []),
    "rx_pool_reset": (c_int, [C.POINTER(RxPoolDesc), c_int64, c_int64, c_void_p]),
    "rx_pool_load": (c_int, [C.POINTER(RxPoolDesc), c_int, c_void_p, c_int64, c_void_p]),
    "rx_pool_snapshot": (c_int, [C.POINTER(RxPoolDesc), c_int, c_void_p, c_int64, c_void_p]),
    "rx_pool_alloc": (c_int, [C.POINTER(RxPoolDesc), c_int64, c_int, c_void_p, c_void_p]),
    "rx_pool_alloc_extend": (c_int, [C.POINTER(RxPoolDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int64, c_void_p]),
    "rx_pool_alloc_decode": (c_int, [C.POINTER(RxPoolDesc), c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                     c_void_p]),
    "rx_pool_alloc_decode_rows": (c_int, [C.POINTER(RxPoolDesc), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                          c_int, c_int64, c_void_p]),
    "rx_pool_alloc_extend_rows": (c_int, [C.POINTER(RxPoolDesc), c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int64,
                                          c_void_p]),
    "rx_pool_append": (c_int, [C.POINTER(RxPoolDesc), c_int, c_void_p, c_int64, c_void_p]),
    "rx_pool_prepend_strided": (c_int, [C.POINTER(RxPoolDesc), c_int, c_void_p, c_int64, c_int, c_int64, c_int64,
                                        c_int, c_void_p]),
    "rx_pool_mark": (c_int, [C.POINTER(RxPoolDesc), c_void_p, c_int64, c_int, c_void_p]),
    "rx_pool_flush_marks": (c_int, [C.POINTER(RxPoolDesc), c_int, c_void_p]),
    "rx_pool_merge_sort": (c_int, [C.POINTER(RxPoolDesc), c_void_p]),
    # decode context parallel
    "rx_dcp_kv_indices": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                  c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "rx_dcp_store_loc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "rx_dcp_local_merge": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "rx_dcp_widen": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "rx_dcp_scale": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rx_dcp_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                              c_int, c_void_p]),
    "rx_move_kv_layout": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    # host-side radix tree
    "rx_radix_create": (c_void_p, [c_int, c_int]),
    "rx_radix_destroy": (None, [c_void_p]),
    "rx_radix_reset": (None, [c_void_p]),
    "rx_radix_root": (c_int64, [c_void_p]),
    "rx_radix_match_prefix": (c_int64, [c_void_p, c_void_p, c_int64, C.c_char_p, c_void_p, c_int64,
                                        C.POINTER(c_int64)]),
    "rx_radix_insert": (c_int64, [c_void_p, c_void_p, c_void_p, c_int64, C.c_char_p, c_int64, c_int,
                                  C.POINTER(c_int64)]),
    "rx_radix_inc_lock_ref": (c_int64, [c_void_p, c_int64]),
    "rx_radix_dec_lock_ref": (c_int64, [c_void_p, c_int64]),
    "rx_radix_evict": (c_int64, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                 C.POINTER(c_int64)]),
    "rx_radix_evictable_size": (c_int64, [c_void_p]),
    "rx_radix_protected_size": (c_int64, [c_void_p]),
    "rx_radix_total_size": (c_int64, [c_void_p]),
    "rx_radix_num_nodes": (c_int64, [c_void_p]),
    "rx_radix_node_info": (c_int, [c_void_p, c_int64, c_void_p]),
    "rx_radix_cache_req": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, C.c_char_p, c_int, c_int, c_int64, c_int64,
                                   c_void_p, c_int64, c_void_p]),
}


class RadixHipError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load (building first if the .so is absent/stale and hipcc exists).  Raises on failure."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if build_if_missing and _build.needs_build():
            try:
                _build.build()
            except Exception as e:  # stale-but-present lib: usable only if the ABI checks below pass
                if not os.path.exists(path):
                    raise RadixHipError(f"libradix_hip.so missing and build failed: {e}") from e
                import warnings
                warnings.warn(f"libradix_hip.so is older than its sources and the rebuild failed ({e}); "
                              f"loading the stale library", RuntimeWarning)
        if not os.path.exists(path):
            raise RadixHipError(f"{path} not found: run `python -m sglang_amd.build`")
        lib = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the header and the .so disagree
            fn.restype = res
            fn.argtypes = args
        if lib.rx_version() != RX_ABI_VERSION:
            raise RadixHipError(f"ABI version mismatch: library {lib.rx_version()}, binding {RX_ABI_VERSION} "
                                f"(rebuild: python -m sglang_amd.build --force)")
        for which, st in enumerate((RxKvLayout, RxDecodeParams, RxExtendParams)):
            if lib.rx_abi_sizeof(which) != C.sizeof(st):
                raise RadixHipError(f"{path}: sizeof({st.__name__}) is {lib.rx_abi_sizeof(which)} in the library, "
                                    f"{C.sizeof(st)} in the binding -- stale build (python -m sglang_amd.build --force)")
        _lib = lib
        return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().rx_last_error()
        raise RadixHipError(f"{what or 'libradix_hip'} failed with status {status}: "
                            f"{msg.decode() if msg else ''}")


def last_dispatch() -> str:
    """Name of the kernel instance this thread's last rx_extend_attn / rx_decode_attn launched (rx_last_dispatch)."""
    msg = load().rx_last_dispatch()
    return msg.decode() if msg else ""


def debug_counters(reset: bool = False):
    """rx_debug_counters: (softmax blocks, redone blocks) of the counting extend instance (option ext32_count_redo)."""
    import ctypes as _C

    out = (_C.c_uint64 * 2)()
    check(load().rx_debug_counters(out, int(bool(reset))), "rx_debug_counters")
    return int(out[0]), int(out[1])


def clock_probe(out2_dev, spin_us: int, stream) -> None:
    """rx_clock_probe: one wave on `stream` that watches both clocks for spin_us; out2_dev = 2 x uint64 device tensor
    (shader cycles, 100-MHz ticks)."""
    check(load().rx_clock_probe(out2_dev.data_ptr(), int(spin_us), stream), "rx_clock_probe")


_roctx = None


def range_push(name: str) -> None:
    """A roctx range from the host side (e.g. one per layer around RadixAttention.forward) when option `roctx` is on;
    the library's own entry points add theirs inside it (RX_RANGE, csrc/rx_common.h)."""
    global _roctx
    if not get_option("roctx"):
        return
    if _roctx is None:
        try:
            _roctx = C.CDLL("libroctx64.so")
        except OSError:
            _roctx = False
    if _roctx:
        _roctx.roctxRangePushA(name.encode())


def range_pop() -> None:
    if _roctx and get_option("roctx"):
        _roctx.roctxRangePop()


def set_option(name: str, value: int) -> int:
    """rx_set_option: a process-wide dispatch switch (include/radix_hip.h lists the names).  Returns the old value."""
    old = c_int(0)
    check(load().rx_get_option(name.encode(), C.byref(old)), "rx_get_option")
    check(load().rx_set_option(name.encode(), int(value)), "rx_set_option")
    return int(old.value)


def get_option(name: str) -> int:
    v = c_int(0)
    check(load().rx_get_option(name.encode(), C.byref(v)), "rx_get_option")
    return int(v.value)


class option:
    """``with lib.option("ext32_autopack", 0): ...`` -- a dispatch switch for the duration of a block (tests)."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, int(value)

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False
