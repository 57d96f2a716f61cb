"""Per-layer time of plain decode vs the cascade over SEVERAL shared prefixes (ops.CascadeGroups) on a batch of G groups:
every group shares its own SHARED-token prefix, every request has UNIQ private tokens; LONERS requests share nothing.
env: GROUPS (4) PER (64) SHARED (3584) UNIQ (512) LONERS (0) HQ (32) HKV (8) PAGE (16) LAYERS (4) ITERS (30)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

E = lambda k, d: int(os.environ.get(k, d))  # noqa: E731
G, per, shared, uniq, loners = E("GROUPS", 4), E("PER", 64), E("SHARED", 3584), E("UNIQ", 512), E("LONERS", 0)
hq, hkv, page, layers, iters, d = E("HQ", 32), E("HKV", 8), E("PAGE", 16), E("LAYERS", 4), E("ITERS", 30), 128
dev, dt = "cuda", torch.bfloat16
rng = np.random.default_rng(0)
bs = G * per + loners
ctx = shared + uniq
pg = lambda n: -(-n // page)  # noqa: E731
n_pages = G * pg(shared) + G * per * pg(uniq) + loners * pg(ctx) + 2
ids = rng.permutation(np.arange(1, n_pages))
take = [0]


def pages(n):
    k = pg(n)
    out = (ids[take[0]: take[0] + k, None] * page + np.arange(page)[None]).reshape(-1)[:n]
    take[0] += k
    return out


r2t = np.zeros((bs + 1, ctx), dtype=np.int32)
order = rng.permutation(bs)  # groups scattered over the batch
groups, row = [], 0
for g in range(G):
    sh = pages(shared)
    members = []
    for _ in range(per):
        b = int(order[row]); row += 1
        r2t[b + 1] = np.concatenate([sh, pages(uniq)])
        members.append(b)
    groups.append((sorted(members), shared))
for _ in range(loners):
    b = int(order[row]); row += 1
    r2t[b + 1] = pages(ctx)
pool = n_pages * page
kbs = [torch.randn(pool, hkv, d, device=dev, dtype=dt) for _ in range(layers)]
vbs = [torch.randn(pool, hkv, d, device=dev, dtype=dt) for _ in range(layers)]
q = torch.randn(bs, hq, d, device=dev, dtype=dt)
o1, o2 = (torch.zeros(bs, hq, d, device=dev, dtype=dt) for _ in range(2))
r2t_d = torch.from_numpy(r2t).to(dev)
rpi = torch.arange(1, bs + 1, device=dev, dtype=torch.int64)
lens = torch.full((bs,), ctx, device=dev, dtype=torch.int64)
sm = d ** -0.5


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters / layers * 1e3


def plain():
    for l in range(layers):
        ops.decode_attention_fwd_paged(q, kbs[l], vbs[l], o1, r2t_d, rpi, lens, None, None, None, 1, sm, page_size=page)


cg = ops.CascadeGroups(bs, hq, hkv, d, dt, dev, max_shared_total=G * shared + 64)
t0 = __import__("time").perf_counter()
cg.plan(r2t_d, rpi, lens, groups)
torch.cuda.synchronize()
plan_ms = (__import__("time").perf_counter() - t0) * 1e3


def casc():
    for l in range(layers):
        cg(q, kbs[l], vbs[l], o2, sm, page_size=page)


tp, tc = timed(plain), timed(casc)
plain(); casc(); torch.cuda.synchronize()
err = (o1.float() - o2.float()).abs().max().item()
kv_bytes = bs * ctx * hkv * d * 2 * 2
uniq_bytes = (G * shared + G * per * uniq + loners * ctx) * hkv * d * 2 * 2
print(f"groups={G} x {per} requests (+{loners} loners), shared {shared} + private {uniq}: plain {tp:.1f} us/layer "
      f"({kv_bytes / tp / 1e6:.2f} TB/s of {kv_bytes / 1e6:.0f} MB), cascade {tc:.1f} us/layer (distinct rows {uniq_bytes / 1e6:.0f} MB), "
      f"x{tp / tc:.2f}; chunks {cg.num_chunks}, plan {plan_ms:.2f} ms (first call), max |diff| {err:.4f}")
