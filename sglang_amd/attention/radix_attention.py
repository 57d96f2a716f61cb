"""RadixAttention layer record (srt/layers/radix_attention.py:91-287): the per-layer scalars the
backend reads, and ``forward`` = reshape + dispatch to the attention backend (:279-287)."""
from __future__ import annotations

from enum import Enum
from typing import Optional


class AttentionType(Enum):
    DECODER = "decoder"
    DECODER_BIDIRECTIONAL = "decoder_bidirectional"
    ENCODER_ONLY = "encoder_only"


class RadixAttention:
    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int,
                 layer_id: int, logit_cap: float = 0.0, v_head_dim: int = -1,
                 sliding_window_size: int = -1, is_cross_attention: bool = False,
                 logit_capping_method: str = "tanh",
                 attn_type: AttentionType = AttentionType.DECODER):
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.head_dim = head_dim
        self.qk_head_dim = head_dim
        self.v_head_dim = v_head_dim if v_head_dim != -1 else head_dim
        self.scaling = scaling
        self.layer_id = layer_id
        self.logit_cap = logit_cap
        self.sliding_window_size = sliding_window_size or -1
        self.is_cross_attention = is_cross_attention
        self.k_scale = None
        self.v_scale = None
        self.k_scale_float: Optional[float] = None
        self.v_scale_float: Optional[float] = None
        self.attn_type = attn_type
        self.logit_capping_method = logit_capping_method
        self.xai_temperature_len = -1

    def forward(self, q, k, v, forward_batch, attn_backend, save_kv_cache: bool = True, **kwargs):
        if k is not None:
            assert v is not None
            k = k.reshape(-1, self.tp_k_head_num, self.qk_head_dim)
            v = v.reshape(-1, self.tp_v_head_num, self.v_head_dim)
        return attn_backend.forward(q, k, v, self, forward_batch, save_kv_cache, **kwargs)

    __call__ = forward
