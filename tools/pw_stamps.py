#!/usr/bin/env python3
"""Where a tile of rx::extend_pw_kernel's generated loop spends its cycles, from a diagnostic build:
  RX_LIB_NAME=libradix_hip_pwstamp.so RX_CFLAGS=-DRX_PW_STAMP=1 python -m sglang_amd.build
  RX_LIB_NAME=libradix_hip_pwstamp.so RX_EXT_PW=2 python tools/pw_stamps.py
Stamps (shader cycles summed over a wave's pipelined tiles): 0 = tile top (DMA wait, barrier, table step), 1..4 = the
groups G1..G4 of the generated body (16 MFMAs each: 512 cycles of matrix pipe), 5 = everything else (boundary tiles,
drains, prologue, epilogue); word 6 = pipelined tiles of the wave."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

os.environ.setdefault("RX_EXT_PW", "2")
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.extend_attention_fwd(q, k_ext, v_ext, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, True,
                             None, E, 1.0, 1.0, sm_scale=D ** -0.5, page_size=1)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
flops = 4.0 * HQ * D * chunk * (E * P + E * (E + 1) / 2)
print(f"{ms:.3f} ms per chunk = {flops / ms / 1e9:.0f} TFLOP/s (stamp build: its fences forbid overlaps the real kernel has)")
rows = o.view(chunk, E // 64, 64, HQ, D)[:, :, 0]          # first row of every wave: [req, wave-of-request, head, D]
raw_i = rows.contiguous().view(torch.int32)[..., :14]
raw = raw_i[..., :9].to(torch.float64)  # [req, wave, head, 9]
cyc = raw_i[..., 9].to(torch.float64)
rt = (raw_i[..., 10].to(torch.int64) & 0xffffffff).to(torch.float64)
print("in-kernel clock: %.3f GHz (median over waves); wave duration mean %.1f us, max %.1f us" % (
    (cyc / (rt * 10.0)).median().item(), (rt * 0.01).mean().item(), (rt * 0.01).max().item()))
start = (raw_i[..., 11].to(torch.int64) & 0xffffffff)
hw = raw_i[..., 12].to(torch.int64) & 0xffffffff
xcc = raw_i[..., 13].to(torch.int64) & 0xf
# CU identity: XCC id + SE id (bits 13..15) + CU id (bits 8..11) of HW_ID on gfx9
cu = (xcc << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15)
w0 = (torch.arange(E // 64, device=dev) % 4 == 0)           # one wave per workgroup
cu0, st0, du0 = cu[:, w0].reshape(-1), start[:, w0].reshape(-1), rt[:, w0].reshape(-1)
t_first, t_last = st0.min().item(), (st0.to(torch.float64) + du0).max().item()
busy = {}
for c, d in zip(cu0.tolist(), du0.tolist()):
    busy[c] = busy.get(c, 0.0) + d
vals = torch.tensor(list(busy.values()))
print("kernel span %.1f us; %d CUs seen; per-CU sum of workgroup durations: mean %.1f us (%.0f %% of the span), min %.1f, max %.1f; workgroups per CU %.1f" % (
    (t_last - t_first) * 0.01, len(busy), vals.mean().item() * 0.01, 100 * vals.mean().item() / (t_last - t_first),
    vals.min().item() * 0.01, vals.max().item() * 0.01, cu0.numel() / len(busy)))
# words: 0 scalar code between tiles, 1..4 G1..G4, 5 rest, 6 vmcnt wait, 7 barrier wait, 8 pipelined tiles
st = torch.stack([raw[..., 0], raw[..., 6], raw[..., 7], raw[..., 1], raw[..., 2], raw[..., 3], raw[..., 4], raw[..., 5]], -1)
names = ["scalar", "dma-wait", "barrier", "G1", "G2", "G3", "G4", "rest"]
tiles = raw[..., 8]
per = st[..., :7] / tiles[..., None]
print("pipelined tiles per wave: mean %.1f" % tiles.mean().item())
print("cycles per pipelined tile:", {n: round(v) for n, v in zip(names[:7], per.mean((0, 1, 2)).tolist())},
      "sum", round(per.sum(-1).mean().item()), "(matrix pipe: 2048)")
print("rest per wave (boundary tiles, drains, prologue, epilogue):", round(st[..., 7].mean().item()))
tot = st.sum(-1)
print("wave total cycles mean %.0f max %.0f" % (tot.mean().item(), tot.max().item()))
for wv in range(E // 64):
    print("  wave", wv, {n: round(v) for n, v in zip(names[:7], per[:, wv].mean((0, 1)).tolist())}, "rest",
          round(st[:, wv, :, 7].mean().item()), "tiles", round(tiles[:, wv].mean().item(), 1))
