#!/usr/bin/env python3
"""Random skewed batches through the split-slot grid and the live-pairs grid (two and three workgroups per CU), with the
first-pass, the mixed 3 x CUs and the rounds-rule schedules: device counts == host mirror, outputs bit-identical across
grids, finite, and equal to a one-pass run within fp32-merge noise.  python tools/fuzz_split_items.py [trials]"""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

dev = "cuda"
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(123)
bad = 0
for tr in range(trials):
    hq, hkv = [(32, 8), (8, 8), (16, 2), (8, 1)][tr % 4]
    d, ps = 128, [1, 16, 64][tr % 3]
    bs = int(rng.integers(1, 70))
    kind = tr % 5
    if kind == 0:
        lens = rng.integers(1, 3000, size=bs)
    elif kind == 1:
        lens = np.concatenate([rng.integers(8000, 40000, size=1), rng.integers(1, 1500, size=max(bs - 1, 0))])
    elif kind == 2:
        lens = np.concatenate([rng.integers(4000, 12000, size=min(3, bs)), rng.integers(1, 600, size=max(bs - 3, 0))])
    elif kind == 3:
        lens = np.full(bs, int(rng.integers(100, 5000)))
    else:
        lens = np.concatenate([[0, 1], rng.integers(0, 9000, size=max(bs - 2, 0))])[:max(bs, 2)]
    lens = lens.astype(np.int64)
    bs = len(lens)
    pages = [-(-int(n) // ps) for n in lens]
    perm = rng.permutation(np.arange(1, sum(pages) + 2))
    r2t = np.zeros((bs + 1, int(lens.max()) + ps + 1), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        r2t[i + 1, :n] = (perm[pi: pi + pages[i], None] * ps + np.arange(ps)[None]).reshape(-1)[:n]
        pi += pages[i]
    pool = (sum(pages) + 2) * ps
    dt = torch.bfloat16 if tr % 2 else torch.float16
    kb = torch.randn(pool, hkv, d, device=dev).to(dt)
    vb = torch.randn(pool, hkv, d, device=dev).to(dt)
    q = torch.randn(bs, hq, d, device=dev).to(dt)
    r2td = torch.from_numpy(r2t).to(dev)
    rpi = torch.arange(1, bs + 1, device=dev)
    lens_d = torch.from_numpy(lens).to(dev)
    order = torch.argsort(lens_d, descending=True).to(torch.int32)
    group = hq // hkv
    blocks = bs * hkv * ((group + 15) // 16)
    mint = 1024 if 2 * blocks >= 256 else 128
    outs = {}
    for mixed in (0, 768, -1):
        hc = ops.balanced_kv_splits_host(lens, hq, hkv, 32, 512, mint, mixed)
        dv_ = torch.zeros(bs, dtype=torch.int32, device=dev)
        ops.get_num_kv_splits_balanced(dv_, lens_d, hq, hkv, 32, 512, mint, mixed)
        if dv_.cpu().numpy().tolist() != hc.tolist():
            print("MISMATCH counts", tr, mixed, lens[:6], dv_.cpu().numpy()[:6], hc[:6]); bad += 1
        S = int(hc.max())
        if S <= 1:
            continue
        S8 = (S + 7) // 8 * 8
        for grid in ("slots", "pairs2", "pairs3"):
            o = torch.full((bs, hq, d), float("nan"), dtype=dt, device=dev)
            al = torch.zeros(bs, hq, S8, d, dtype=torch.float32, device=dev)
            ls = torch.zeros(bs, hq, S8, dtype=torch.float32, device=dev)
            cnt = torch.zeros(bs * hq, dtype=torch.int32, device=dev)
            si = None
            if grid != "slots":
                si = ops.SplitItems(int(np.maximum(hc, 1).sum()), dev).build(dv_, order, wgs_per_cu=3 if grid == "pairs3" else 0)
            ops.decode_attention_fwd_paged(q, kb, vb, o, r2td, rpi, lens_d, al, ls, dv_, S8, d ** -0.5, page_size=ps,
                                           merge_counters=cnt, request_order=order, split_items=si)
            torch.cuda.synchronize()
            if int(cnt.abs().sum()) != 0:
                print("COUNTERS not reset", tr, mixed, grid); bad += 1
            outs[(mixed, grid)] = o
        a = outs[(mixed, "slots")].view(torch.int16)
        for grid in ("pairs2", "pairs3"):
            if not torch.equal(a, outs[(mixed, grid)].view(torch.int16)):
                print("BITS differ", tr, mixed, grid, lens[:6]); bad += 1
    o1 = torch.full((bs, hq, d), float("nan"), dtype=dt, device=dev)
    ops.decode_attention_fwd_paged(q, kb, vb, o1, r2td, rpi, lens_d, None, None, None, 1, d ** -0.5, page_size=ps)
    torch.cuda.synchronize()
    live = torch.from_numpy(lens > 0).to(dev)
    for k, o in outs.items():
        x = o[live].float()
        if not torch.isfinite(x).all():
            print("NONFINITE", tr, k); bad += 1
        err = (x - o1[live].float()).abs().max().item() if live.any() else 0.0
        if err > (4e-2 if dt == torch.bfloat16 else 6e-3):
            print("ERR vs one pass", tr, k, err); bad += 1
print("fuzz_split_items:", trials, "trials,", bad, "problems")
sys.exit(1 if bad else 0)
