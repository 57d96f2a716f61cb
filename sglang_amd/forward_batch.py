"""Input record of the path: the fields of the reference's ForwardBatch that the attention
backend and the KV pools read (srt/model_executor/forward_batch_info.py:412-638) and the
ForwardMode predicates it dispatches on (:98-196).  Same names, same meaning."""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum, auto
from typing import List, Optional

import torch


class ForwardMode(IntEnum):
    EXTEND = auto()
    DECODE = auto()
    MIXED = auto()
    IDLE = auto()
    TARGET_VERIFY = auto()     # speculative decoding: verify the draft tree (extend under a custom mask)
    DRAFT_EXTEND_V2 = auto()   # speculative decoding: draft model extends over the accepted tokens

    def is_prefill(self, include_draft_extend_v2: bool = False):
        return self.is_extend(include_draft_extend_v2=include_draft_extend_v2)

    def is_extend(self, include_draft_extend_v2: bool = False):
        return (self in (ForwardMode.EXTEND, ForwardMode.MIXED, ForwardMode.TARGET_VERIFY)
                or (include_draft_extend_v2 and self == ForwardMode.DRAFT_EXTEND_V2))

    def is_target_verify(self):
        return self == ForwardMode.TARGET_VERIFY

    def is_draft_extend_v2(self):
        return self == ForwardMode.DRAFT_EXTEND_V2

    def is_extend_without_speculative(self):
        return self.is_extend() and not self.is_target_verify()

    def is_cuda_graph(self):
        return self in (ForwardMode.DECODE, ForwardMode.TARGET_VERIFY, ForwardMode.IDLE)

    def is_decode(self):
        return self == ForwardMode.DECODE

    def is_mixed(self):
        return self == ForwardMode.MIXED

    def is_idle(self):
        return self == ForwardMode.IDLE

    def is_decode_or_idle(self):
        return self in (ForwardMode.DECODE, ForwardMode.IDLE)


@dataclass
class ForwardBatch:
    forward_mode: ForwardMode
    batch_size: int
    req_pool_indices: torch.Tensor  # int32/int64 [bs]
    seq_lens: torch.Tensor  # int32/int64 [bs]
    out_cache_loc: Optional[torch.Tensor]  # int64 [num_tokens]
    seq_lens_sum: Optional[int] = None
    seq_lens_cpu: Optional[torch.Tensor] = None
    positions: Optional[torch.Tensor] = None
    # extend-only
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[torch.Tensor] = None  # int32 [bs]
    extend_prefix_lens: Optional[torch.Tensor] = None  # int32 [bs]
    extend_start_loc: Optional[torch.Tensor] = None
    extend_seq_lens_cpu: Optional[List[int]] = None
    extend_prefix_lens_cpu: Optional[List[int]] = None
    encoder_lens: Optional[torch.Tensor] = None
    spec_info: object = None
    # decode-only, optional: the radix-tree node each request's cached prefix ends in (req.last_node, kept by the
    # scheduler from RadixCache.match_prefix, radix_cache.py:352-430) -- or the shared-prefix groups themselves,
    # [(member batch rows, shared token count)] -- for the backend's cascade over several prefixes (ops.CascadeGroups)
    radix_last_nodes: Optional[list] = None
    cascade_groups: Optional[list] = None

    @classmethod
    def for_decode(cls, req_pool_indices, seq_lens, out_cache_loc, seq_lens_cpu=None):
        """ForwardBatch.init_new for a decode batch (:705-970; positions = seq_lens-1 :902-904)."""
        if seq_lens_cpu is None:
            seq_lens_cpu = seq_lens.cpu()
        return cls(forward_mode=ForwardMode.DECODE, batch_size=len(seq_lens),
                   req_pool_indices=req_pool_indices, seq_lens=seq_lens,
                   out_cache_loc=out_cache_loc, seq_lens_sum=int(seq_lens_cpu.sum()),
                   seq_lens_cpu=seq_lens_cpu, positions=torch.clamp(seq_lens - 1, min=0).to(torch.int64))

    @classmethod
    def for_extend(cls, req_pool_indices, seq_lens, out_cache_loc, extend_prefix_lens_cpu,
                   extend_seq_lens_cpu):
        """ForwardBatch.init_new for an extend batch (:905-931)."""
        dev = seq_lens.device
        pre = torch.tensor(extend_prefix_lens_cpu, dtype=torch.int32, device=dev)
        ext = torch.tensor(extend_seq_lens_cpu, dtype=torch.int32, device=dev)
        start = torch.zeros_like(ext)
        start[1:] = torch.cumsum(ext[:-1], dim=0)
        seq_cpu = seq_lens.cpu()
        # compute_position (forward_batch_info.py:923, 1689-...): prefix_len + 0 .. extend_len - 1 per request
        positions = torch.cat([torch.arange(p_, p_ + e_, dtype=torch.int64)
                               for p_, e_ in zip(extend_prefix_lens_cpu, extend_seq_lens_cpu)]
                              or [torch.zeros(0, dtype=torch.int64)]).to(dev)
        return cls(forward_mode=ForwardMode.EXTEND, batch_size=len(seq_lens), positions=positions,
                   req_pool_indices=req_pool_indices, seq_lens=seq_lens,
                   out_cache_loc=out_cache_loc, seq_lens_sum=int(seq_cpu.sum()),
                   seq_lens_cpu=seq_cpu, extend_num_tokens=int(sum(extend_seq_lens_cpu)),
                   extend_seq_lens=ext, extend_prefix_lens=pre, extend_start_loc=start,
                   extend_seq_lens_cpu=list(extend_seq_lens_cpu),
                   extend_prefix_lens_cpu=list(extend_prefix_lens_cpu))
