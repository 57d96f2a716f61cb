#!/usr/bin/env python3
"""Sliding-window / capped extend at head dim 256 (Gemma-class layers): ms per call of the config-3 chunk shape
(32 requests x (3584 cached + 512 new), 16 q / 8 kv heads; env BS P E HQ HKV D W).  RX_OPT_EXTEND_D256=0 times
rx_extend_nd.hip instead; D=64 HQ=64 HKV=8 P=0 E=8192 W=128 BS=4 is a gpt-oss-like sliding-window layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
E_ = lambda k, v: int(os.environ.get(k, v))  # noqa: E731
bs, P, E, hq, hkv, d, W = E_("BS", 32), E_("P", 3584), E_("E", 512), E_("HQ", 16), E_("HKV", 8), E_("D", 256), E_("W", 1024)
g = torch.Generator(device=dev).manual_seed(2)
pool = bs * (P + E) + 64
kb = torch.randn(pool, hkv, d, device=dev, generator=g).to(torch.bfloat16)
vb = torch.randn(pool, hkv, d, device=dev, generator=g).to(torch.bfloat16)
perm = (torch.randperm(pool - 1, device=dev, generator=g)[: bs * (P + E)] + 1).view(bs, P + E)
kv_indices = perm[:, :P].reshape(-1).contiguous()
kv_indptr = (torch.arange(bs + 1, device=dev, dtype=torch.int32) * P).contiguous()
qo = (torch.arange(bs + 1, device=dev, dtype=torch.int64) * E).contiguous()
q = torch.randn(bs * E, hq, d, device=dev, generator=g).to(torch.bfloat16)
ke, ve = kb[perm[:, P:].reshape(-1)].contiguous(), vb[perm[:, P:].reshape(-1)].contiguous()
o = torch.empty_like(q)
for name, kw in (("plain", {}), (f"window {W}", dict(sliding_window_size=W)), ("logit cap 50", dict(logit_cap=50.0)),
                 (f"window {W} + cap 50", dict(sliding_window_size=W, logit_cap=50.0))):
    def call():
        ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kv_indptr, kv_indices, None, True, None, E, 1.0, 1.0,
                                 sm_scale=d ** -0.5, **kw)
    call()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        call()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{name:22s} {ev[0].elapsed_time(ev[1]) / 5:.3f} ms")
