"""Decode context parallel (DCP) for the HIP backend -- SURVEY 8e "alternative shardings".

The KV cache of ONE request is spread over the ``size`` ranks of a group: the token at position p lives on rank
``p % size`` at local slot ``virtual_slot // size`` (srt/layers/dcp/layout.py).  Attention then runs in three steps per
layer (TritonAttnBackend.forward_decode, triton_backend.py:1797-1839; _forward_extend_dcp, :1439-1569):

1. all-gather the q heads of the group (every rank needs every head against ITS tokens),
2. the ordinary kernels over the rank's local kv_indices, producing a partial result and its LSE,
3. ``cp_lse_ag_out_rs_mha`` (srt/layers/dcp/comm.py:82-108): all-gather the LSEs, scale the partial by
   exp(lse - logsumexp), sum over the ranks, keep this rank's heads.

The arithmetic of steps 2-3 is HIP (csrc/rx_dcp.hip, fp32); this module owns the exchanges, which are plain
``torch.distributed`` collectives over the group (RCCL on a GPU group; through the host for a gloo group, which is how
the tests run two ranks on one GPU)."""
from typing import Optional

import torch
import torch.distributed as dist

from sglang_amd import ops


def get_dcp_lens(lens: torch.Tensor, dcp_size: int, dcp_rank: int, start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Tokens of [start, start + lens) that ``dcp_rank`` owns (layout.py:23-41), for host-side planning code (the kernels'
    copy is in rx_dcp_kv_indices): positions p with p % dcp_size == dcp_rank below an end e number
    floor((e - 1 - rank) / size) + 1, so the count in a range is a difference of two floor divisions."""
    if dcp_size == 1:
        return lens
    first = 0 if start is None else start
    below_end = torch.div(first + lens - 1 - dcp_rank, dcp_size, rounding_mode="floor")
    below_start = torch.div(first - 1 - dcp_rank + torch.zeros_like(lens), dcp_size, rounding_mode="floor")
    return below_end - below_start


class DcpGroup:
    """The three exchanges of DCP over one process group."""

    def __init__(self, size: int, rank: int, group: Optional["dist.ProcessGroup"] = None):
        if size < 1 or not 0 <= rank < size:
            raise ValueError(f"dcp rank {rank} of {size}")
        self.size, self.rank, self.group = int(size), int(rank), group
        if size > 1:
            if not dist.is_initialized():
                raise RuntimeError("DCP needs an initialised torch.distributed process group")
            if dist.get_world_size(group) != size or dist.get_rank(group) != rank:
                raise ValueError("dcp size / rank do not match the process group")
            self._device_collectives = dist.get_backend(group) == "nccl"

    def _all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """[size, *x.shape], rank-major."""
        out = torch.empty((self.size,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        if self._device_collectives:
            dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
        else:  # gloo: through the host
            parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(self.size)]
            dist.all_gather(parts, x.detach().cpu().contiguous(), group=self.group)
            out.copy_(torch.stack(parts))
        return out

    def all_gather_heads(self, q_local: torch.Tensor) -> torch.Tensor:
        """[T, H_loc, D] -> [T, size * H_loc, D]; rank r's heads at [r * H_loc, (r + 1) * H_loc) (group.all_gather
        along dim 1, triton_backend.py:1807)."""
        if self.size == 1:
            return q_local
        T, H, D = q_local.shape
        return self._all_gather(q_local).permute(1, 0, 2, 3).reshape(T, self.size * H, D)

    def all_gather_lse(self, lse: torch.Tensor) -> torch.Tensor:
        """[T, H] -> [size, T, H] (comm.py:71-79)."""
        return self._all_gather(lse)

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self.size > 1:
            if self._device_collectives:
                dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            else:
                h = x.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                x.copy_(h)
        return x

    # ------------------------------------------------------------------ step 3
    def merge_partials(self, o32_all: torch.Tensor, lse_all: torch.Tensor, out: torch.Tensor,
                       cur_o: Optional[torch.Tensor] = None, cur_lse: Optional[torch.Tensor] = None) -> torch.Tensor:
        """cp_lse_ag_out_rs_mha: ``o32_all`` fp32 [T, H_all, Dv] / ``lse_all`` [T, H_all] are this rank's partial over
        ALL heads of the group; ``out`` [T, H_loc, Dv] receives this rank's heads of the joined result.  With
        ``cur_o`` / ``cur_lse`` (the extend path's own-chunk partial over the local heads) that one is joined in as
        well (triton_backend.py:1560-1569)."""
        h_loc = out.shape[1]
        lses = self.all_gather_lse(lse_all.contiguous())
        glse = None
        if cur_o is not None:
            glse = torch.empty_like(lse_all)
        ops.dcp_scale(o32_all, lses, self.rank, glse)
        self.all_reduce(o32_all)
        return ops.dcp_finish(o32_all, out, self.rank * h_loc, glse, cur_o, cur_lse)
