#!/usr/bin/env python3
"""Phase-time breakdown of the extend32 fast loop from a diagnostic build:
  RX_LIB_NAME=libradix_hip_stamp.so RX_CFLAGS=-DRX_EXT32_STAMP=1 python sglang_amd/build.py
  RX_LIB_NAME=libradix_hip_stamp.so python tools/ext_stamps.py
Stamps (shader cycles, summed over a wave's tiles): 0 barrier, 1 QK^T(b0,b1)+softmax(b0),
2 PV(b0)+softmax(b1), 3 PV(b1)+staging, 4 boundary tiles + loop exit, 5 loop overhead."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops

dev = torch.device("cuda:0")
HQ, HKV, D, P, E, chunk = 32, 8, 128, 3584, 512, 32
g = torch.Generator(device=dev).manual_seed(1)
pool = P + chunk * E + 16
kb = torch.randn(pool, HKV, D, device=dev, generator=g).to(torch.bfloat16)
vb = torch.randn(pool, HKV, D, device=dev, generator=g).to(torch.bfloat16)
T = chunk * E
q = torch.randn(T, HQ, D, device=dev, generator=g).to(torch.bfloat16)
k_ext = torch.randn(T, HKV, D, device=dev, generator=g).to(torch.bfloat16)
v_ext = torch.randn(T, HKV, D, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty_like(q)
kv_indices = torch.arange(16, 16 + P, device=dev, dtype=torch.int64).repeat(chunk)
kv_indptr = (torch.arange(chunk + 1, device=dev) * P).to(torch.int32)
qo_indptr = (torch.arange(chunk + 1, device=dev) * E).to(torch.int64)
for _ in range(3):
    ops.extend_attention_fwd(q, k_ext, v_ext, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, True,
                             None, E, 1.0, 1.0, sm_scale=D ** -0.5, page_size=1)
torch.cuda.synchronize()
QPW = int(os.environ.get("QPW", "64"))
rows = o.view(chunk, E // QPW, QPW, HQ, D)[:, :, 0]          # first row of every wave: [req, wave, head, D]
st = rows.contiguous().view(torch.int32)[..., :6].to(torch.float64)  # [req, wave, head, 6]
names = ["barrier", "QK+SM0", "PV0+SM1", "PV1+stage", "boundary+exit", "loop ovh"]
tot = st.sum(-1, keepdim=True)
print("mean cycles per wave:", {n: round(v) for n, v in zip(names, st.mean((0, 1, 2)).tolist())})
print("share:", {n: round(v, 3) for n, v in zip(names, (st / tot).mean((0, 1, 2)).tolist())})
nfast = 56 + 0  # prefix tiles are all fast; extend tiles fast until the diagonal
print("per fast tile (approx, /56):", {n: round(v / 56) for n, v in zip(names[:4], st.mean((0, 1, 2)).tolist()[:4])})
