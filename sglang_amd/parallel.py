"""Tensor-parallel plumbing of the attention path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The attention itself shards by heads with NO exchange: every rank owns Hq/tp query
heads and max(1, Hkv/tp) kv heads (kv heads are replicated when Hkv < tp, models/llama.py:158-171)
and reads the same page table.  The single exchange is the sum all-reduce of the row-parallel
o_proj output [tokens, hidden] (srt/layers/linear.py:1606-1627 ->
srt/distributed/communication_op.py:18-20 -> parallel_state.py:622-732).

MI355X: the all-reduce is issued on a side HIP stream so that it overlaps the next layer's
attention / GEMM on the main stream; events fence both directions.  At 2 MiB per message
(bs=256 x 4096 x bf16) RCCL's latency-optimised tree/direct algorithms are the right regime; ring
bandwidth does not matter.
"""
from __future__ import annotations

import enum
import warnings
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class HeadShard:
    tp_size: int
    tp_rank: int
    num_q_heads: int  # per rank
    num_kv_heads: int  # per rank
    q_head_start: int  # first global q head owned
    kv_head_start: int  # first global kv head owned (replicated groups share it)
    kv_replicas: int  # ranks sharing one kv head (tp / Hkv when Hkv < tp, else 1)


def shard_heads(total_q_heads: int, total_kv_heads: int, tp_size: int, tp_rank: int) -> HeadShard:
    """LlamaAttention.__init__ (srt/models/llama.py:158-171): q heads split evenly; kv heads split
    when Hkv >= tp (must divide), replicated when Hkv < tp (tp must be a multiple of Hkv)."""
    if total_q_heads % tp_size != 0:
        raise ValueError(f"q heads {total_q_heads} not divisible by tp {tp_size}")
    nq = total_q_heads // tp_size
    if total_kv_heads >= tp_size:
        if total_kv_heads % tp_size != 0:
            raise ValueError(f"kv heads {total_kv_heads} not divisible by tp {tp_size}")
        nkv, rep = total_kv_heads // tp_size, 1
        kv_start = tp_rank * nkv
    else:
        if tp_size % total_kv_heads != 0:
            raise ValueError(f"tp {tp_size} not a multiple of kv heads {total_kv_heads}")
        nkv, rep = 1, tp_size // total_kv_heads
        kv_start = tp_rank // rep
    return HeadShard(tp_size, tp_rank, nq, nkv, tp_rank * nq, kv_start, rep)


class CustomAllReduce:
    """Peer-to-peer two-shot all-reduce over IPC-mapped buffers (csrc/rx_allreduce.hip; the role of
    the reference's custom all-reduce behind GroupCoordinator.all_reduce, parallel_state.py:622-732) and its
    fused all-reduce + residual add + RMSNorm form (fused_allreduce_rmsnorm, :748-878).
    One instance per rank; the 64-byte IPC handles travel through the torch.distributed group.

    A context's kernels count their calls in device memory, so captured launches replay correctly under HIP
    graphs, but the calls of ONE context must be ordered (one stream, or one chain of graph dependencies).  This
    object therefore owns ``lanes`` contexts with a region each: lane 0 for the caller's main stream, lane 1 for
    the side (communication) stream of TPGroup.all_reduce_async.

    Opt-in (``RX_CUSTOM_AR=1``): RCCL stays the default because the kernels have only been exercised with several
    processes on ONE GPU (the gpurun box), not across xGMI."""

    def __init__(self, group: Optional[dist.ProcessGroup], device: torch.device, max_bytes: int = 8 << 20,
                 lanes: int = 2):
        import ctypes as C

        from . import lib as _L

        self._L, self._C = _L, C
        lib = self._lib = _L.load()
        self.group, self.device = group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.max_bytes = (int(max_bytes) + 255) // 256 * 256
        region_bytes = lib.rx_ar_region_bytes(self.max_bytes)
        self._own, self._opened, self._ctxs = [], [], []
        with torch.cuda.device(device):
            self.err_flag = torch.zeros(1, dtype=torch.int32, device=device)
            for _ in range(lanes):
                own = C.c_void_p()
                _L.check(lib.rx_ar_alloc_region(region_bytes, C.byref(own)), "rx_ar_alloc_region")
                self._own.append(own)
                handle = C.create_string_buffer(64)
                _L.check(lib.rx_ipc_get_handle(own, handle), "rx_ipc_get_handle")
                handles = [None] * self.world
                dist.all_gather_object(handles, bytes(handle.raw), group=group)
                ptrs = (C.c_void_p * self.world)()
                for r, h in enumerate(handles):
                    if r == self.rank:
                        ptrs[r] = own.value
                    else:
                        p = C.c_void_p()
                        _L.check(lib.rx_ipc_open_handle(C.create_string_buffer(h, 64), C.byref(p)), "rx_ipc_open_handle")
                        ptrs[r] = p.value
                        self._opened.append(p)
                ctx = C.c_void_p()
                _L.check(lib.rx_ar_init(C.byref(ctx), self.rank, self.world, ptrs, self.max_bytes,
                                        C.c_void_p(self.err_flag.data_ptr())), "rx_ar_init")
                self._ctxs.append(ctx)
        self._ctx = self._ctxs[0]
        dist.barrier(group=group)  # every region is mapped everywhere before the first call

    def shape_ok(self, x: torch.Tensor) -> bool:
        """The rank-independent half of ``supports``: dtype and element count only (every rank of a group sees the
        same answer for the same logical tensor)."""
        return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.numel() > 0
                and x.numel() % 8 == 0 and x.numel() * 2 <= self.max_bytes)

    def supports(self, x: torch.Tensor) -> bool:
        return self.shape_ok(x) and x.is_contiguous() and x.data_ptr() % 16 == 0

    def _dt(self, x):
        return self._L.RX_BF16 if x.dtype == torch.bfloat16 else self._L.RX_F16

    def all_reduce(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, lane: int = 0) -> torch.Tensor:
        out = x if out is None else out
        st = self._lib.rx_allreduce(self._ctxs[lane], self._C.c_void_p(x.data_ptr()), self._C.c_void_p(out.data_ptr()),
                                    x.numel(), self._dt(x),
                                    self._C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        self._L.check(st, "rx_allreduce")
        return out

    def all_reduce_det(self, x: torch.Tensor, lane: int = 0) -> torch.Tensor:
        """Deterministic in-place sum (rx_allreduce_det: the one-shot form, every rank sums every element in fp32 in rank
        order and rounds once -- the reference's AMD path under --enable-deterministic-inference,
        custom_all_reduce.py:294-301).  ANY size: a message above the context's max_bytes goes through the staging
        region in max_bytes pieces (custom_all_reduce.py:277-278: the deterministic implementation takes every size);
        the arithmetic is per element, so the cut points do not show in the result.  16-bit, contiguous, 16-byte
        aligned, a multiple of 8 elements -- ``det_ok(x)``."""
        if not self.det_ok(x):
            raise ValueError("all_reduce_det: needs a contiguous, 16-byte aligned bf16 / fp16 tensor of a multiple of 8 elements")
        flat = x.view(-1)
        step = self.max_bytes // 2
        cp = self._C.c_void_p
        stream = cp(torch.cuda.current_stream(x.device).cuda_stream)
        for lo in range(0, flat.numel(), step):
            piece = flat[lo: lo + step]
            st = self._lib.rx_allreduce_det(self._ctxs[lane], cp(piece.data_ptr()), cp(piece.data_ptr()), piece.numel(),
                                            self._dt(x), stream)
            self._L.check(st, "rx_allreduce_det")
        return x

    def det_shape_ok(self, x: torch.Tensor) -> bool:
        """Rank-independent: dtype and element count only."""
        return x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.numel() > 0 and x.numel() % 8 == 0

    def det_ok(self, x: torch.Tensor) -> bool:
        return self.det_shape_ok(x) and x.is_contiguous() and x.data_ptr() % 16 == 0

    def supports_fused_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor) -> bool:
        return (x.dim() == 2 and self.supports(x) and residual.shape == x.shape and residual.dtype == x.dtype
                and residual.is_contiguous() and weight.dtype == x.dtype and weight.is_contiguous()
                and weight.numel() == x.shape[1] and x.shape[1] % 8 == 0 and x.shape[1] <= 16384
                and residual.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)

    def fused_allreduce_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                                lane: int = 0):
        """(out, residual_out): residual_out = all_reduce(x) + residual, out = rmsnorm(residual_out) * weight
        (parallel_state.py:748-878), one kernel.  residual_out is written IN PLACE over ``residual``."""
        if not self.supports_fused_rmsnorm(x, residual, weight):
            raise ValueError("fused_allreduce_rmsnorm: unsupported shapes / dtypes (see supports_fused_rmsnorm)")
        out = torch.empty_like(x)
        cp = self._C.c_void_p
        st = self._lib.rx_allreduce_rmsnorm(self._ctxs[lane], cp(x.data_ptr()), cp(residual.data_ptr()),
                                            cp(weight.data_ptr()), cp(out.data_ptr()), cp(residual.data_ptr()),
                                            x.shape[0], x.shape[1], float(eps), self._dt(x),
                                            cp(torch.cuda.current_stream(x.device).cuda_stream))
        self._L.check(st, "rx_allreduce_rmsnorm")
        return out, residual

    # ---- the reference class's own method names (CustomAllreduce, custom_all_reduce.py:40-340), for a caller written against it
    disabled = False
    _SUPPORTED_WORLD_SIZES = [2, 3, 4, 5, 6, 7, 8]   # (the reference: 2, 4, 6, 8; the kernels here take any 2 .. 8)

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        """custom_all_reduce.py:260-283: a multiple of 16 bytes, dense, at most max_bytes (16-bit dtypes only here)."""
        dense = inp.is_contiguous() or (inp.untyped_storage().nbytes() - inp.storage_offset() * inp.element_size()
                                        == inp.numel() * inp.element_size())
        return bool(dense and self.shape_ok(inp) and inp.data_ptr() % 16 == 0)

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:   # noqa: A002  (the reference's parameter name)
        """custom_all_reduce.py:309-329: the OUT-OF-PLACE sum, or None when this communicator does not take the tensor (the
        caller then falls through to the next one).  Graph capture needs no special casing here: nothing is registered, the
        kernels count their calls on the device."""
        if not self.should_custom_ar(input):
            return None
        out = torch.empty_like(input)
        return self.all_reduce(input, out=out)   # (element-wise: a dense permuted layout sums as it lies; empty_like keeps the strides)

    def capture(self):
        """custom_all_reduce.py:182-194 registers the graph's buffers at the end of the capture; the staging regions of this
        implementation are fixed, so the context manager has nothing to do."""
        import contextlib

        return contextlib.nullcontext()

    def check_errors(self) -> int:
        v = int(self.err_flag.item())
        if v:
            self.err_flag.zero_()
        return v

    def close(self):
        if getattr(self, "_ctxs", None):
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)  # nobody still reads a region that is about to go away
            for ctx in self._ctxs:
                self._lib.rx_ar_destroy(ctx)
            for p in self._opened:
                self._lib.rx_ipc_close_handle(p)
            dist.barrier(group=self.group)  # ... and nobody frees a region a peer still has mapped (see QuickAllReduce.close)
            for own in self._own:
                self._lib.rx_ar_free_region(own)
            self._ctxs, self._ctx = [], None


CustomAllreduce = CustomAllReduce   # the reference's spelling (custom_all_reduce.py:40)


class QuickReduceRegime(enum.Enum):
    """quick_all_reduce.py:41-46; the first four are rx_qr_level (include/radix_hip.h)."""
    FP = 0
    INT8 = 1
    INT6 = 2
    INT4 = 3
    NONE = 4


MB = 1024 * 1024


class QuickAllReduce:
    """The quick all-reduce for LARGE 16-bit messages (csrc/rx_quick_allreduce.hip): two-shot, pushed, with a block-scaled
    INT8 / INT6 / INT4 (or plain 16-bit) wire format.  Mirrors the reference's QuickAllReduce (srt/distributed/
    device_communicators/quick_all_reduce.py:49-267): same environment switches --
      ROCM_QUICK_REDUCE_QUANTIZATION = FP | INT8 | INT6 | INT4 | NONE (default NONE: the communicator stays disabled),
      ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16 (default 1: bf16 tensors travel and are summed as fp16),
      ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB (default: 2 GiB) --
    same world sizes (2, 4, 8), dtypes and size gate (``should_quick_allreduce``), out-of-place ``quick_all_reduce``, and
    the same arithmetic (the result of a level is a function of the inputs, not of the schedule; the tests hold it to the
    CPU restatement under oracle/, bit for bit).  The minimum sizes are the reference's table: it was measured for ITS kernel on
    MI300 and is kept as the contract of the environment switch, not as a tuned MI355X threshold -- this pool has one GPU
    per box, so neither this kernel nor the table has been timed across xGMI.

    One context = one fixed 64-MiB region per rank whatever the message size; calls of one context must be stream ordered
    (HIP-graph capture is fine: the tile counters live in device memory), so the object owns ``lanes`` contexts: lane 0 for
    the caller's stream, lane 1 for TPGroup.all_reduce_async's communication stream."""

    _SUPPORTED_WORLD_SIZES = [2, 4, 8]
    _SUPPORTED_DTYPES = [torch.float16, torch.bfloat16]
    # [FP, INT8, INT6, INT4] (quick_all_reduce.py:55-66)
    _QR_MIN_SIZE = {
        (torch.float16, 2): [1 * MB, 2 * MB, 2 * MB, 1 * MB],
        (torch.float16, 4): [1 * MB, 16 * MB, 4 * MB, 2 * MB],
        (torch.float16, 8): [16 * MB, 4 * MB, 4 * MB, 2 * MB],
        (torch.bfloat16, 2): [2 * MB, 8 * MB, 8 * MB, 8 * MB],
        (torch.bfloat16, 4): [8 * MB, 64 * MB, 64 * MB, 16 * MB],
        (torch.bfloat16, 8): [16 * MB, 2048 * MB, 2048 * MB, 2048 * MB],
    }

    def __init__(self, group: Optional[dist.ProcessGroup], device, regime: Optional[str] = None,
                 cast_bf16_to_fp16: Optional[bool] = None, max_size_mb: Optional[int] = None, lanes: int = 2):
        import os

        self.disabled = True
        self.group = group
        self.device = torch.device(f"cuda:{device}") if isinstance(device, int) else torch.device(device)
        self._ptr, self._ptrs = None, []
        if not dist.is_initialized():
            return
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)
        if self.world_size == 1:
            return
        if self.world_size not in self._SUPPORTED_WORLD_SIZES:
            warnings.warn(f"quick all-reduce is disabled: world size {self.world_size} not in {self._SUPPORTED_WORLD_SIZES}")
            return
        regime = os.environ.get("ROCM_QUICK_REDUCE_QUANTIZATION", "NONE") if regime is None else regime
        if regime not in QuickReduceRegime.__members__:
            warnings.warn(f"quick all-reduce: invalid quantization level {regime!r}; supported: {list(QuickReduceRegime.__members__)}")
            return
        if regime == "NONE":
            return
        self.qr_quant_level = QuickReduceRegime[regime]
        self.use_fp16_kernels = (int(os.environ.get("ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16", 1)) if cast_bf16_to_fp16 is None
                                 else int(bool(cast_bf16_to_fp16)))
        mb = int(os.environ.get("ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB", 0)) if max_size_mb is None else int(max_size_mb)
        self.qr_max_size = mb * MB if mb > 0 else 1 << 31   # (qr_max_size(), quick_all_reduce.cu:86-89)
        if self.device.type != "cuda":
            return   # (CPU groups of the gloo tests: the gate above is all that runs there)
        from . import quick_ar_ops as ops   # the reference's op names (custom_all_reduce_ops.py:131-163) over the C ABI

        self._ops = ops
        self._ptrs = []
        with torch.cuda.device(self.device):
            # one communicator (= one 64-MiB region) per LANE: a communicator's launches must be ordered, and TPGroup reduces
            # on two streams -- lane 0 the caller's, lane 1 the communication stream of all_reduce_async (as CustomAllReduce).
            # init_quick_all_reduce + create_shared_buffer, quick_all_reduce.py:176-220:
            for _ in range(max(1, int(lanes))):
                ptr = ops.init_custom_qr(self.rank, self.world_size, self.qr_max_size)
                handles = [None] * self.world_size
                dist.all_gather_object(handles, ops.qr_get_handle(ptr), group=group)
                ops.qr_open_handles(ptr, handles)
                self._ptrs.append(ptr)
            self._ptr = self._ptrs[0]
        dist.barrier(group=group)   # every region is mapped everywhere before the first call
        self.disabled = False

    def size_ok(self, dtype: torch.dtype, nbytes: int) -> bool:
        """The rank-independent gate (quick_all_reduce.py:222-244): dtype, a multiple of 16 bytes, and the level's size
        window for (dtype as it travels, world size)."""
        if self.disabled or dtype not in self._SUPPORTED_DTYPES or nbytes % 16 != 0 or nbytes == 0:
            return False
        travel = torch.float16 if self.use_fp16_kernels else dtype
        return self._QR_MIN_SIZE[(travel, self.world_size)][self.qr_quant_level.value] <= nbytes <= self.qr_max_size

    def should_quick_allreduce(self, inp: torch.Tensor) -> bool:
        if self.disabled or not inp.is_cuda:
            return False
        dense = inp.is_contiguous() or (inp.untyped_storage().nbytes() - inp.storage_offset() * inp.element_size()
                                        == inp.numel() * inp.element_size())
        return dense and self.size_ok(inp.dtype, inp.numel() * inp.element_size())

    def quick_all_reduce(self, inp: torch.Tensor, *, out: Optional[torch.Tensor] = None, lane: int = 0) -> torch.Tensor:
        """Out of place (``out`` may be ``inp``).  ``lane``: which of the object's contexts (one per stream of launches)."""
        if self.disabled:
            raise RuntimeError("quick all-reduce is disabled (ROCM_QUICK_REDUCE_QUANTIZATION, world size, device)")
        if inp.dtype not in self._SUPPORTED_DTYPES or inp.numel() % 8 != 0 or inp.data_ptr() % 16 != 0:
            raise ValueError("quick_all_reduce: needs a 16-byte aligned fp16 / bf16 tensor of a multiple of 8 elements")
        if out is None:
            out = torch.empty_like(inp)
        self._ops.qr_all_reduce(self._ptrs[min(lane, len(self._ptrs) - 1)], inp, out, self.qr_quant_level.value,
                                bool(self.use_fp16_kernels))
        return out

    def check_errors(self) -> int:
        v = 0
        for ptr in getattr(self, "_ptrs", []):
            v |= self._ops.qr_check_errors(ptr)
        return v

    def close(self):
        """Collective.  A context is meant to live as long as its process group (GroupCoordinator builds qr_comm once): with
        eight processes on this driver stack (dmabuf IPC), contexts created AFTER an earlier one had been closed and freed
        either failed in hipIpcGetMemHandle or never saw their peers' flags -- the freed region's mapping appears to be
        reused on the importing side.  Create once, close at exit."""
        if getattr(self, "_ptrs", None):
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
            for ptr in self._ptrs:
                self._ops.qr_close_peers(ptr)
            # every mapping is gone before any owner frees: with eight processes a region freed while a peer still had it
            # open made the NEXT hipIpcGetMemHandle of that owner fail ("invalid argument", dmabuf IPC; round 6)
            dist.barrier(group=self.group)
            for ptr in self._ptrs:
                self._ops.qr_destroy(ptr)
            self._ptrs, self._ptr = [], None
            self.disabled = True


class TPGroup:
    """Thin coordinator over one torch.distributed process group (GroupCoordinator's all_reduce
    entry, parallel_state.py:622-732)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, custom_ar: Optional[CustomAllReduce] = None,
                 deterministic: bool = False, quick_ar: Optional[QuickAllReduce] = None):
        if not dist.is_initialized():
            self.rank, self.world_size, self.group = 0, 1, None
        else:
            self.group = group
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        self._comm_stream = None
        self.custom_ar = custom_ar if self.world_size > 1 else None
        # C3 (GroupCoordinator.qr_comm, parallel_state.py:443-472): taken behind the peer-to-peer kernel's size window and
        # ahead of the backend, for the messages its gate admits (parallel_state.py:886-900)
        self.quick_ar = quick_ar if (self.world_size > 1 and quick_ar is not None and not quick_ar.disabled) else None
        import os

        self._strict = os.environ.get("RX_CUSTOM_AR_STRICT", "0") not in ("", "0")
        # Deterministic inference with TP > 1: the o_proj / fused-RMSNorm reduce leaves RCCL (whose
        # summation order is not pinned) for the fixed rank-order kernels -- rx_allreduce_det for every size, and the fused
        # all-reduce + RMSNorm, which sums in the same fixed order.  The reference does the same on AMD
        # (custom_all_reduce.py:277-278,294-301,415-421: SGLANG_USE_1STAGE_ALLREDUCE, else
        # SGLANG_ENABLE_DETERMINISTIC_INFERENCE).  Needs the peer-to-peer context: a deterministic group without one raises
        # on its first GPU reduce rather than silently keeping RCCL.
        self.deterministic = bool(deterministic) and self.world_size > 1
        self._warned_det = False

    @staticmethod
    def deterministic_collectives_enabled(server_args=None) -> bool:
        """GroupCoordinator._deterministic_collectives_enabled (parallel_state.py:1204-1208) + the server flag:
        SGLANG_USE_1STAGE_ALLREDUCE when set, else SGLANG_ENABLE_DETERMINISTIC_INFERENCE / --enable-deterministic-inference."""
        import os

        def env_bool(name):
            v = os.environ.get(name)
            return None if v is None else v.strip().lower() in ("1", "true", "yes", "on")

        one_stage = env_bool("SGLANG_USE_1STAGE_ALLREDUCE")
        if one_stage is not None:
            return one_stage
        return bool(getattr(server_args, "enable_deterministic_inference", False)) or bool(env_bool("SGLANG_ENABLE_DETERMINISTIC_INFERENCE"))

    @classmethod
    def from_server_args(cls, group: Optional[dist.ProcessGroup], server_args, device: torch.device,
                         max_bytes: int = 8 << 20) -> "TPGroup":
        """The group a model runner builds: under deterministic inference (see deterministic_collectives_enabled) with more
        than one rank the peer-to-peer context is created and every GPU reduce takes the fixed-order kernels; otherwise the
        defaults (RCCL, the two-shot kernel as opt-in through RX_CUSTOM_AR=1)."""
        import os

        det = cls.deterministic_collectives_enabled(server_args)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        ar = qr = None
        if world > 1 and torch.device(device).type == "cuda" and (det or os.environ.get("RX_CUSTOM_AR", "0") not in ("", "0")):
            ar = CustomAllReduce(group, torch.device(device), max_bytes=max_bytes)
        # the quick all-reduce is opt-in through the reference's own switch (ROCM_QUICK_REDUCE_QUANTIZATION != NONE); a lossy
        # wire format has no place under deterministic inference, whose reduce is the fixed-order 16-bit kernel
        if world in QuickAllReduce._SUPPORTED_WORLD_SIZES and torch.device(device).type == "cuda" and not det \
                and os.environ.get("ROCM_QUICK_REDUCE_QUANTIZATION", "NONE") != "NONE":
            qr = QuickAllReduce(group, torch.device(device))
        return cls(group, custom_ar=ar, deterministic=det, quick_ar=qr)

    def _reduce(self, x: torch.Tensor, lane: int = 0) -> None:
        # Per tensor, like GroupCoordinator.all_reduce's should_custom_ar test (parallel_state.py:672-700): the
        # peer-to-peer kernel takes what it can (decode-sized 16-bit activations), everything else -- prefill-sized
        # activations above max_bytes, fp32, odd element counts -- goes to RCCL.  The choice depends on dtype and
        # element count ONLY, never on addresses or strides, so every rank makes the same one (a rank-dependent
        # choice would deadlock: half the group in the kernel's flag exchange, half in RCCL); a tensor the shape
        # rule sends to the kernel but whose memory it cannot take (a strided or misaligned view) is reduced
        # through a contiguous copy.  RX_CUSTOM_AR_STRICT=1 raises instead of falling back (debugging: "which of my
        # tensors are missing the fast path").
        ar = self.custom_ar
        if self.deterministic and x.is_cuda:
            if ar is None:
                raise RuntimeError("deterministic inference with TP > 1 needs the peer-to-peer all-reduce context "
                                   "(TPGroup.from_server_args builds it); refusing to fall back to RCCL")
            if ar.det_shape_ok(x):   # dtype and element count only: the same answer on every rank
                if ar.det_ok(x):
                    ar.all_reduce_det(x, lane=lane)
                else:
                    tmp = x.contiguous().clone() if x.is_contiguous() else x.contiguous()
                    ar.all_reduce_det(tmp, lane=lane)
                    x.copy_(tmp)
                return
            if self._strict:
                raise ValueError(f"deterministic all-reduce: {x.dtype} x {x.numel()} is not a 16-bit tensor of a multiple of 8 elements")
            if not self._warned_det:
                import warnings

                warnings.warn(f"deterministic all-reduce covers 16-bit tensors of a multiple of 8 elements; {x.dtype} x "
                              f"{x.numel()} is reduced by the group's backend (order not pinned)", RuntimeWarning)
                self._warned_det = True
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            return
        if ar is not None and ar.shape_ok(x):
            if ar.supports(x):
                ar.all_reduce(x, lane=lane)
            else:
                tmp = x.contiguous().clone() if x.is_contiguous() else x.contiguous()
                ar.all_reduce(tmp, lane=lane)
                x.copy_(tmp)
            return
        qr = self.quick_ar
        # (rank-independent like the rule above: dtype and byte count only; a strided or misaligned view goes through a copy)
        if qr is not None and x.is_cuda and qr.size_ok(x.dtype, x.numel() * x.element_size()):
            if x.is_contiguous() and x.data_ptr() % 16 == 0:
                qr.quick_all_reduce(x, out=x, lane=lane)
            else:
                tmp = x.contiguous().clone() if x.is_contiguous() else x.contiguous()
                qr.quick_all_reduce(tmp, out=tmp, lane=lane)
                x.copy_(tmp)
            return
        if ar is not None and self._strict:
            raise ValueError("custom all-reduce selected (RX_CUSTOM_AR_STRICT) but the tensor is not 16-bit, a "
                             f"multiple of 8 elements and <= {ar.max_bytes} bytes: {x.dtype} x {x.numel()}")
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)

    def fused_allreduce_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
        """tensor_model_parallel_fused_allreduce_rmsnorm (communication_op.py -> parallel_state.py:748-878):
        (out, residual_out) or None when no fused path applies (the caller then runs all_reduce + add + norm)."""
        if self.world_size == 1 or self.custom_ar is None:
            return None
        if not self.custom_ar.supports_fused_rmsnorm(x, residual, weight):
            return None
        return self.custom_ar.fused_allreduce_rmsnorm(x, residual, weight, eps)

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """In-place sum over the group; bypassed for world size 1 (parallel_state.py:640-642)."""
        if self.world_size == 1:
            return x
        self._reduce(x)
        return x

    # ---- side-stream overlap (GPU only) ------------------------------------------------------
    def all_reduce_async(self, x: torch.Tensor):
        """Enqueue the all-reduce of ``x`` on the communication stream behind everything already
        on the current stream; returns a handle whose ``wait()`` makes the current stream wait
        for the result.  CPU tensors (gloo tests) reduce synchronously."""
        if self.world_size == 1:
            return _Done(x)
        if not x.is_cuda:
            self.all_reduce(x)
            return _Done(x)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=x.device)
            # a ring of reusable events: creating two per call costs ~20 us of host time, a third of a
            # TP=8 shard's per-layer GPU time.  A stream-wait refers to the record that precedes it, so an
            # event may be re-recorded once its waiter has been enqueued; the ring is far longer than the
            # one all-reduce a caller keeps pending.
            self._events = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(16)]
            self._ev_i = 0
        main = torch.cuda.current_stream(x.device)
        ready, done = self._events[self._ev_i]
        self._ev_i = (self._ev_i + 1) % len(self._events)
        ready.record(main)
        self._comm_stream.wait_event(ready)
        with torch.cuda.stream(self._comm_stream):
            self._reduce(x, lane=1)
            if not torch.cuda.is_current_stream_capturing():  # a captured graph owns its memory pool
                x.record_stream(self._comm_stream)
            done.record(self._comm_stream)
        return _Pending(x, done)


class _Done:
    def __init__(self, x):
        self.tensor = x

    def wait(self):
        return self.tensor


class _Pending:
    def __init__(self, x, event):
        self.tensor, self._event = x, event

    def wait(self):
        torch.cuda.current_stream(self.tensor.device).wait_event(self._event)
        return self.tensor


def tensor_model_parallel_all_reduce(x: torch.Tensor, group: Optional[TPGroup] = None) -> torch.Tensor:
    """srt/distributed/communication_op.py:18-20."""
    return (group or TPGroup()).all_reduce(x)


class RowParallelOProj:
    """o_proj of the attention block as a row-parallel linear (linear.py:1606-1627): rank r holds
    rows [r*Hq_local*D, (r+1)*Hq_local*D) of W_o [Hq*D, hidden]; forward = local GEMM + sum
    all-reduce.  The GEMM is a plain library GEMM (torch.matmul -> hipBLASLt)."""

    def __init__(self, full_weight: torch.Tensor, shard: HeadShard, head_dim: int, group: TPGroup):
        lo = shard.q_head_start * head_dim
        hi = lo + shard.num_q_heads * head_dim
        # stored [hidden, Hq_local*D] (the nn.Linear layout): hipBLASLt's NT kernel runs the TP=1 shape
        # [256, 4096] x [4096, 4096] in 19-20 us against 24-25 us for the row-major weight
        self.weight_t = full_weight[lo:hi].t().contiguous()
        self.group = group

    @property
    def weight(self) -> torch.Tensor:
        return self.weight_t.t()

    def forward(self, attn_out: torch.Tensor, overlap: bool = False):
        y = torch.nn.functional.linear(attn_out, self.weight_t)
        if overlap:
            return self.group.all_reduce_async(y)
        return _Done(self.group.all_reduce(y))
