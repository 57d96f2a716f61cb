#!/usr/bin/env python3
"""Dev micro-bench: config-5-shaped MLA decode (Hq=16 per GPU at TP=8, Hkv=1, Dk=576, Dv=512,
bs=64, ctx=8192, bf16 latent rows) through rx_decode_attn."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

bs, ctx, hq, dk, dv, ps = int(os.environ.get("BS", "64")), int(os.environ.get("CTX", "8192")), int(os.environ.get("HQ", "16")), 576, 512, int(os.environ.get("PS", "1"))
dev = "cuda"
pool = bs * ctx + ps
kv = torch.empty(pool, 1, dk, dtype=torch.bfloat16, device=dev).normal_()
FP8 = bool(os.environ.get("FP8"))
if FP8:
    kv = kv.to(torch.float8_e4m3fn)
q = torch.randn(bs, hq, dk, device=dev).to(torch.bfloat16)
o = torch.empty(bs, hq, dv, dtype=torch.bfloat16, device=dev)
perm = torch.randperm(bs * ctx // ps, device=dev) + 1 if ps > 1 else torch.randperm(bs * ctx, device=dev) + 1
if ps > 1:
    slots = (perm.view(bs, -1, 1) * ps + torch.arange(ps, device=dev)).view(bs, -1)[:, :ctx]
else:
    slots = perm.view(bs, ctx)
r2t = torch.zeros(bs + 1, ctx, dtype=torch.int32, device=dev)
r2t[1:] = slots.int()
rpi = torch.arange(1, bs + 1, device=dev)
lens = torch.full((bs,), ctx, dtype=torch.int64, device=dev)
S = int(os.environ.get("S", "8"))
nsplit = torch.zeros(bs, dtype=torch.int32, device=dev)
ops.get_num_kv_splits(nsplit, lens.int(), hq, 1, S, 256)
print("splits", nsplit[:4].tolist())
al = torch.empty(bs, hq, S, dv, dtype=torch.float32, device=dev)
lse = torch.empty(bs, hq, S, dtype=torch.float32, device=dev)


cnt = torch.zeros(bs * hq, dtype=torch.int32, device=dev) if os.environ.get("MC") else None  # in-kernel stage 2


def run():
    ops.decode_attention_fwd_paged(q, kv, kv[..., :dv], o, r2t, rpi, lens, al, lse, nsplit, S, dk ** -0.5,
                                   page_size=ps, merge_counters=cnt)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byt = bs * ctx * dk * (1 if FP8 else 2)
print(f"MLA decode bs={bs} ctx={ctx} Hq={hq}: {ms*1e3:.1f} us  {byt/ms/1e6:.0f} GB/s ({byt/ms/1e6/8000:.1%} of 8 TB/s)")

