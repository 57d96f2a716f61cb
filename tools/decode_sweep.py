#!/usr/bin/env python3
"""Dev sweep: dense decode kernel time vs (batch, ctx, forced split count) at Llama-3-8B head geometry.
env: HQ / HKV (head counts, default 32 / 8), SHAPES (e.g. 256x4096,1x32768), SPLITS (e.g. 1,2,8), NS (live splits of the
S allocated), GRAPH=1 (20 calls captured once: GPU time without the host), MC=1 (in-kernel stage 2: merge_counters),
CONTIG=1 (pages in order instead of shuffled), INTERLEAVE=1 (K and V pages side by side in one allocation), VPAD=<bytes> (K and V in one allocation, V displaced)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops

dev = "cuda"
HQ, HKV, D, PS = int(os.environ.get("HQ", "32")), int(os.environ.get("HKV", "8")), int(os.environ.get("D", "128")), 16
def run(bs, ctx, splits_list):
    pages = ctx // PS
    rng = np.random.default_rng(0)
    perm = np.arange(1, bs * pages + 1) if os.environ.get("CONTIG") else rng.permutation(np.arange(1, bs * pages + 1))
    slots = (perm.reshape(bs, pages)[:, :, None] * PS + np.arange(PS)[None, None, :]).reshape(bs, -1)
    r2t = torch.zeros(bs + 1, ctx, dtype=torch.int32, device=dev); r2t[1:] = torch.from_numpy(slots.astype(np.int32)).to(dev)
    rpi = torch.arange(1, bs + 1, device=dev); lens = torch.full((bs,), ctx, dtype=torch.int64, device=dev)
    pool = (bs * pages + 1) * PS
    kb = torch.randn(pool // PS, HKV, PS, D, device=dev).to(torch.bfloat16)
    if os.environ.get("INTERLEAVE"):  # K and V pages of one allocation side by side: [pages, 2, Hkv, page, D] (strides are arguments)
        big = torch.randn(pool // PS, 2, HKV, PS, D, device=dev).to(torch.bfloat16)
        kb = big[:, 0]
    if os.environ.get("VPAD"):  # K and V in ONE allocation, V displaced by VPAD bytes past K's end (DRAM channel / bank aliasing probe)
        pad = int(os.environ["VPAD"]) // 2
        big = torch.empty(2 * kb.numel() + pad + 64, dtype=torch.bfloat16, device=dev)
        big[: kb.numel()].copy_(kb.view(-1)); kb = big[: kb.numel()].view(kb.shape)
        vb = big[kb.numel() + pad: 2 * kb.numel() + pad].view(kb.shape); vb.normal_()
    elif os.environ.get("INTERLEAVE"):
        vb = big[:, 1]
    else:
        vb = torch.randn_like(kb)
    lay = ops.kv_layout_hnd(kb, vb)
    q = torch.randn(bs, HQ, D, device=dev).to(torch.bfloat16); o = torch.empty_like(q)
    ref = torch.zeros(bs, dtype=torch.int32, device=dev)
    ops.get_num_kv_splits(ref, lens.int(), HQ, HKV, 8, 256)
    out = [f"bs={bs} ctx={ctx} reference schedule(max 8)={ref[0].item()}"]
    cnt = torch.zeros(bs * HQ, dtype=torch.int32, device=dev) if os.environ.get("MC") else None  # in-kernel stage 2
    byt = bs * ctx * HKV * D * 2 * 2
    for S in splits_list:
        ns = torch.full((bs,), int(os.environ.get("NS") or S), dtype=torch.int32, device=dev)  # NS: live splits of S allocated
        al = torch.empty(bs, HQ, max(S, 1), D, dtype=torch.float32, device=dev); lse = torch.empty(bs, HQ, max(S, 1), device=dev)
        def f():
            if S == 1: ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, None, None, None, 1, D ** -0.5, page_size=PS, kv_layout=lay)
            else: ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, al, lse, ns, S, D ** -0.5, page_size=PS, kv_layout=lay, merge_counters=cnt)
        for _ in range(3): f()
        torch.cuda.synchronize()
        if os.environ.get("GRAPH"):  # 20 calls captured once: launch-to-launch time on the GPU, no host in the way
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side): f()
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(20): f()
            gr.replay(); torch.cuda.synchronize()
            run20 = gr.replay
        else:
            def run20():
                for _ in range(20): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run20()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append(f"S={S}: {us:.0f} us {byt/us/1e6:.2f} TB/s")
    print(" | ".join(out))
SHAPES = [(64, 2048), (16, 4096), (8, 16384), (1, 32768), (1, 131072), (4, 8192), (32, 1024), (256, 4096)]
if os.environ.get("SHAPES"):  # e.g. SHAPES=256x512,256x1024
    SHAPES = [tuple(int(v) for v in x.split("x")) for x in os.environ["SHAPES"].split(",")]
for bs, ctx in SHAPES:
    run(bs, ctx, [int(x) for x in os.environ.get("SPLITS", "1,2,4,8,16,32,64").split(",")])
