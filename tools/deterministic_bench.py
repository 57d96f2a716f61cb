#!/usr/bin/env python3
"""Dev: what deterministic inference costs on the config-3 extend chunk and on the decode step's layer.
  extend: ops.extend_attention_fwd (two-stage, pipelined tiles) vs ops.extend_attention_fwd_unified (one tile body over the
          unified kv list) on 32 req x (3584 prefix + 512 new), Hq 32 / Hkv 8, D 128, bf16, page 16 HND shuffled;
          + the same with score_mod = relative_bias_score_mod (extent 1024)
  decode: bs 256 x 4096, the native schedule vs ceil(len / 256) = 16 splits per request (+ stage 2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import lib as rxlib  # noqa: E402
from sglang_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
HQ, HKV, D, ps = 32, 8, 128, 16
P, E, chunk = 3584, 512, 32
g = torch.Generator(device=dev).manual_seed(1)
n_pages = (P + chunk * E) // ps + 2
kb = torch.randn((n_pages, HKV, ps, D), device=dev, generator=g).to(torch.bfloat16)
vb = torch.randn((n_pages, HKV, ps, D), device=dev, generator=g).to(torch.bfloat16)
lay = ops.kv_layout_hnd(kb, vb)
T = chunk * E
q = torch.randn(T, HQ, D, device=dev, generator=g).to(torch.bfloat16)
perm = torch.randperm(n_pages - 1, device=dev, generator=g) + 1
slots = (perm[:, None] * ps + torch.arange(ps, device=dev)[None, :]).reshape(-1)
pre = slots[:P].to(torch.int64)
new = slots[P: P + T].to(torch.int64)
# the new tokens' K / V as the pool holds them (so both forms see the same values)
kf, vf = kb.permute(0, 2, 1, 3).reshape(-1, HKV, D), vb.permute(0, 2, 1, 3).reshape(-1, HKV, D)
ke, ve = kf[new].contiguous(), vf[new].contiguous()
kvi = pre.repeat(chunk)
kvp = (torch.arange(chunk + 1, device=dev) * P).to(torch.int32)
qo = (torch.arange(chunk + 1, device=dev) * E).to(torch.int64)
start = (torch.arange(chunk, device=dev) * E).to(torch.int32)
elens = torch.full((chunk,), E, dtype=torch.int32, device=dev)
u_indptr, u_idx, plens = ops.build_unified_kv_indices(kvp, kvi, start, elens, new, chunk, max_tokens_per_request=P + E)
aux = torch.randn(T, HQ, 1024, device=dev, generator=g).to(torch.bfloat16)
o1, o2 = torch.empty_like(q), torch.empty_like(q)
flops = 4.0 * HQ * D * chunk * (E * P + E * (E + 1) / 2)


def two_stage(**kw):
    ops.extend_attention_fwd(q, ke, ve, o1, kb, vb, qo, kvp, kvi, None, True, None, E, 1.0, 1.0, sm_scale=D ** -0.5,
                             page_size=ps, kv_layout=lay, **kw)


def unified(**kw):
    ops.extend_attention_fwd_unified(q, o2, kb, vb, 1.0, 1.0, qo, u_indptr, u_idx, plens, E, sm_scale=D ** -0.5,
                                     page_size=ps, kv_layout=lay, **kw)


def timed(fn, n=20):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


bias = dict(score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])
for name, fn in (("two-stage", two_stage), ("unified (deterministic)", unified),
                 ("two-stage + rel. bias 1024", lambda: two_stage(**bias)), ("unified + rel. bias 1024", lambda: unified(**bias))):
    ms = timed(fn)
    print(f"extend {name:28s} {ms:.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s   {rxlib.last_dispatch()}")
print("max |two-stage - unified| =", (o1.float() - o2.float()).abs().max().item())
