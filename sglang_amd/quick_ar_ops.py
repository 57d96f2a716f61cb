"""Op-level surface of the quick all-reduce, with the names, argument order and meaning of the reference's
``custom_all_reduce_ops`` (srt/distributed/device_communicators/custom_all_reduce_ops.py:131-163 -> sgl_kernel ops
init_custom_qr / qr_get_handle / qr_open_handles / qr_all_reduce / qr_destroy / qr_max_size,
kernels/aot/csrc/allreduce/quick_all_reduce.cu:10-89) over the C ABI (rx_qr_* / rx_quick_allreduce, include/radix_hip.h).

The reference's ``QuickAllReduce`` drives exactly these six functions (quick_all_reduce.py:203-260): with
``ops = sglang_amd.quick_ar_ops`` its ``init_quick_all_reduce`` / ``create_shared_buffer`` / ``quick_all_reduce`` /
``close`` run unchanged.  ``sglang_amd.parallel.QuickAllReduce`` is built on the same six calls.

Differences that do not show at this surface: a handle's buffer is a FIXED 64-MiB region whatever ``qr_max_size`` says
(slots are per workgroup, not per tile -- csrc/rx_quick_allreduce.hip), so ``qr_max_size`` only bounds the message the
caller will send; the 64-byte IPC handle travels as a uint8 CPU tensor as in the reference."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import lib as _L

IS_QUICK_AR_AVAILABLE = True   # (custom_all_reduce_ops.py:18: the ROCm build of the reference sets it; the ops below fail loudly without the library)

_MAX_DEFAULT = 1 << 31         # qr_max_size(), quick_all_reduce.cu:86-89: 2 GiB


class _State:
    __slots__ = ("rank", "world", "max_size", "region", "opened", "ctx", "err_flag", "device")


_states = {}
_next = [1]


def init_custom_qr(rank: int, world_size: int, qr_max_size: Optional[int] = None) -> int:
    """A new communicator for ``rank`` of ``world_size`` (2, 4 or 8) on the CURRENT device: allocates and zeroes its shared
    region.  Returns an opaque integer handle (the reference returns the C++ object's address)."""
    if world_size not in (2, 4, 8):
        # quick_all_reduce.cu:10-14
        raise ValueError(f"init_custom_qr: world size {world_size} is not supported (2, 4 or 8)")
    if not 0 <= rank < world_size:
        raise ValueError("init_custom_qr: invalid rank passed in")
    lib = _L.load()
    st = _State()
    st.rank, st.world = int(rank), int(world_size)
    st.max_size = int(qr_max_size) if qr_max_size and qr_max_size > 0 else _MAX_DEFAULT
    st.device = torch.device("cuda", torch.cuda.current_device())
    st.region = C.c_void_p()
    _L.check(lib.rx_ar_alloc_region(lib.rx_qr_region_bytes(), C.byref(st.region)), "rx_ar_alloc_region")
    st.opened, st.ctx = [], None
    st.err_flag = torch.zeros(1, dtype=torch.int32, device=st.device)
    fa = _next[0]
    _next[0] += 1
    _states[fa] = st
    return fa


def qr_get_handle(fa: int) -> torch.Tensor:
    """The communicator's IPC handle: 64 bytes as a uint8 CPU tensor (quick_all_reduce.cu:28-35)."""
    st = _states[fa]
    buf = C.create_string_buffer(64)
    _L.check(_L.load().rx_ipc_get_handle(st.region, buf), "rx_ipc_get_handle")
    return torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()


def qr_open_handles(fa: int, handles: List[torch.Tensor]) -> None:
    """Map every rank's buffer (``handles[r]`` = rank r's qr_get_handle, own entry included as in the reference,
    quick_all_reduce.cu:37-49) and arm the communicator."""
    st = _states[fa]
    if len(handles) != st.world:
        raise ValueError(f"qr_open_handles: {len(handles)} handles for {st.world} ranks")
    lib = _L.load()
    ptrs = (C.c_void_p * st.world)()
    for r, h in enumerate(handles):
        if r == st.rank:
            ptrs[r] = st.region.value
            continue
        raw = bytes(h.cpu().contiguous().view(torch.uint8).numpy().tobytes()) if isinstance(h, torch.Tensor) else bytes(h)
        p = C.c_void_p()
        _L.check(lib.rx_ipc_open_handle(C.create_string_buffer(raw, 64), C.byref(p)), "rx_ipc_open_handle")
        ptrs[r] = p.value
        st.opened.append(p)
    ctx = C.c_void_p()
    _L.check(lib.rx_qr_init(C.byref(ctx), st.rank, st.world, ptrs, C.c_void_p(st.err_flag.data_ptr())), "rx_qr_init")
    st.ctx = ctx


def qr_all_reduce(fa: int, inp: torch.Tensor, out: torch.Tensor, quant_level: int, cast_bf2half: bool = False) -> None:
    """out = all-reduce sum of inp over the communicator's ranks at ``quant_level`` (0 FP, 1 INT8, 2 INT6, 3 INT4), on the
    current stream (quick_all_reduce.cu:51-84: same dtype and element count, fp16 or bf16, at most qr_max_size bytes)."""
    st = _states[fa]
    if st.ctx is None:
        raise RuntimeError("qr_all_reduce: qr_open_handles has not been called")
    if inp.dtype != out.dtype or inp.numel() != out.numel():
        raise ValueError("qr_all_reduce: inp and out differ in dtype or element count")
    if inp.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("quick allreduce only supports float16 and bfloat16")
    if inp.numel() * inp.element_size() > st.max_size:
        raise ValueError(f"qr_all_reduce: {inp.numel() * inp.element_size()} bytes exceed the communicator's {st.max_size}")
    cp = C.c_void_p
    _L.check(_L.load().rx_quick_allreduce(st.ctx, cp(inp.data_ptr()), cp(out.data_ptr()), inp.numel(),
                                          _L.RX_BF16 if inp.dtype == torch.bfloat16 else _L.RX_F16, int(quant_level),
                                          int(bool(cast_bf2half)), cp(torch.cuda.current_stream(inp.device).cuda_stream)),
              "rx_quick_allreduce")


def qr_check_errors(fa: int) -> int:
    """(no reference counterpart) the communicator's device error word (RX_DEVERR_AR_TIMEOUT = 2), cleared by the read."""
    st = _states[fa]
    v = int(st.err_flag.item())
    if v:
        st.err_flag.zero_()
    return v


def qr_close_peers(fa: int) -> None:
    """(no reference counterpart) destroy the context and unmap the peers' buffers, keeping the own buffer allocated: a
    caller that tears a group down puts a barrier between this and qr_destroy, so that nobody frees a buffer a peer still has
    mapped (sglang_amd.parallel.QuickAllReduce.close)."""
    st = _states[fa]
    lib = _L.load()
    if st.ctx is not None:
        lib.rx_qr_destroy(st.ctx)
        st.ctx = None
    for p in st.opened:
        lib.rx_ipc_close_handle(p)
    st.opened = []


def qr_destroy(fa: int) -> None:
    """Unmap the peers and free the communicator (quick_all_reduce.cu:20-26).  A no-op for 0 / unknown handles, as the
    reference's null check."""
    st = _states.pop(fa, None)
    if st is None:
        return
    _states[fa] = st
    qr_close_peers(fa)
    _states.pop(fa, None)
    _L.load().rx_ar_free_region(st.region)


def qr_max_size() -> int:
    return _MAX_DEFAULT
