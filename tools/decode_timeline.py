#!/usr/bin/env python3
"""Where a SHORT decode launch's fixed cost goes: per-workgroup clock stamps of ONE launch of rx::decode_mfma_kernel
(dev build with -DRX_DEC_TIMELINE; built here on first use as sglang_amd/libradix_hip_tl.so).

    SHAPES=128x4096 HQ=8 HKV=1 SPLITS=1,2,4 python tools/decode_timeline.py

Prints, per (shape, splits): launch wall (events, graph of 20), and from the stamps of the LAST of 5 eager launches:
start skew (first -> last workgroup entry), prologue (entry -> first tile landed) p50 / p95, loop time p5 / p50 / p95 / max,
epilogue, the span first entry -> last exit, and the time between the p50 and the last loop end (the tail), per XCC loop-time
medians.  Stamps are 10-ns ticks of s_memrealtime."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RX_LIB_NAME", "libradix_hip_tl.so")
os.environ.setdefault("RX_CFLAGS", "-DRX_DEC_TIMELINE")
os.environ.setdefault("RX_VARIANT_SOURCES", "rx_decode.hip")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sglang_amd import lib as rxlib  # noqa: E402
from sglang_amd import ops  # noqa: E402

dev = "cuda"
HQ, HKV, D, PS = int(os.environ.get("HQ", "8")), int(os.environ.get("HKV", "1")), 128, int(os.environ.get("PS", "16"))


def run(bs, ctx, S, mc):
    pages = ctx // PS
    rng = np.random.default_rng(0)
    perm = rng.permutation(np.arange(1, bs * pages + 1))
    slots = (perm.reshape(bs, pages)[:, :, None] * PS + np.arange(PS)[None, None, :]).reshape(bs, -1)
    r2t = torch.zeros(bs + 1, ctx, dtype=torch.int32, device=dev)
    r2t[1:] = torch.from_numpy(slots.astype(np.int32)).to(dev)
    rpi = torch.arange(1, bs + 1, device=dev)
    lens = torch.full((bs,), ctx, dtype=torch.int64, device=dev)
    kb = torch.randn(bs * pages + 1, HKV, PS, D, device=dev).to(torch.bfloat16)
    vb = torch.randn_like(kb)
    lay = ops.kv_layout_hnd(kb, vb)
    q = torch.randn(bs, HQ, D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    ns = torch.full((bs,), S, dtype=torch.int32, device=dev)
    al = torch.empty(bs, HQ, max(S, 8), D, dtype=torch.float32, device=dev)
    lse = torch.empty(bs, HQ, max(S, 8), device=dev)
    cnt = torch.zeros(bs * HQ, dtype=torch.int32, device=dev) if mc else None
    items = None
    if S > 1 and os.environ.get("ITEMS", "1") == "1":
        items = ops.SplitItems(bs * S, dev).build(ns, None, cap=bs * S)

    def f():
        if S == 1:
            ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, None, None, None, 1, D ** -0.5, page_size=PS, kv_layout=lay)
        else:
            ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, al, lse, ns, max(S, 8) if mc else S, D ** -0.5, page_size=PS,
                                           kv_layout=lay, merge_counters=cnt, split_items=items)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            f()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    if os.environ.get("NOSTAMPS"):  # (a product library: the launch time only; checks the result against the unsplit pass)
        o_s = o.clone()
        ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, None, None, None, 1, D ** -0.5, page_size=PS, kv_layout=lay)
        torch.cuda.synchronize()
        print(f"bs={bs} ctx={ctx} Hq={HQ} Hkv={HKV} S={S}: graph {us:.1f} us/launch ({bs * ctx * HKV * D * 4 / us / 1e6:.2f} TB/s) "
              f"{rxlib.last_dispatch() if False else ''} max|o - o_unsplit| {(o_s.float() - o.float()).abs().max().item():.4f}")
        return
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    n = bs * HKV * S
    buf = (C.c_ulonglong * (6 * n))()
    rc = rxlib.load().rx_dev_decode_timeline(buf, n)
    assert rc == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.float64)
    t0 = t[:, 0].min()
    ent, first, loop_end, epi, xcc, mrg = ((t[:, i] - t0) / 100.0 for i in range(6))  # us
    xcc = t[:, 4]
    byt = bs * ctx * HKV * D * 2 * 2
    pr = first - ent
    lp = loop_end - first
    ep = epi - loop_end
    last = max(epi.max(), (t[:, 5].max() - t0) / 100.0 if t[:, 5].max() > 0 else 0)
    q_ = lambda a, p: float(np.percentile(a, p))  # noqa: E731
    print(f"bs={bs} ctx={ctx} Hq={HQ} Hkv={HKV} S={S} mc={int(bool(mc))} wgs={n}: graph {us:.1f} us/launch ({byt / us / 1e6:.2f} TB/s) | "
          f"span {last:.1f} us | entry skew {ent.max():.2f} | prologue p50 {q_(pr, 50):.2f} p95 {q_(pr, 95):.2f} | loop p5 {q_(lp, 5):.1f} "
          f"p50 {q_(lp, 50):.1f} p95 {q_(lp, 95):.1f} max {lp.max():.1f} | loop-end p50 {q_(loop_end, 50):.1f} last {loop_end.max():.1f} | "
          f"epilogue p50 {q_(ep, 50):.2f} max {ep.max():.2f} | ideal {byt / 6.3e6:.1f} us at 6.3 TB/s")
    med = [f"{int(x)}:{np.median(lp[xcc == x]):.1f}/{np.median(loop_end[xcc == x]):.1f}" for x in sorted(set(xcc.tolist()))]
    print("   per-XCC loop median / loop-end median (us): " + " ".join(med))


for sh in os.environ.get("SHAPES", "128x4096").split(","):
    bs, ctx = (int(v) for v in sh.split("x"))
    for S in (int(x) for x in os.environ.get("SPLITS", "1,2,4").split(",")):
        run(bs, ctx, S, mc=S > 1)
