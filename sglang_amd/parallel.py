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


class TPGroup:
    """Thin coordinator over one torch.distributed process group (GroupCoordinator's all_reduce
    entry, parallel_state.py:622-732)."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            self.rank, self.world_size, self.group = 0, 1, None
        else:
            self.group = group
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        self._comm_stream = None

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        """In-place sum over the group; bypassed for world size 1 (parallel_state.py:640-642)."""
        if self.world_size == 1:
            return x
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
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
        main = torch.cuda.current_stream(x.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self._comm_stream.wait_event(ready)
        with torch.cuda.stream(self._comm_stream):
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            x.record_stream(self._comm_stream)
            done = torch.cuda.Event()
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
        self.weight = full_weight[lo:hi].contiguous()
        self.group = group

    def forward(self, attn_out: torch.Tensor, overlap: bool = False):
        y = torch.matmul(attn_out, self.weight)
        if overlap:
            return self.group.all_reduce_async(y)
        return _Done(self.group.all_reduce(y))
