"""Per-layer time of plain vs shared-prefix (cascade) decode on a radix-hit batch.
env: BS (256) SHARED (3584) UNIQ (512) HQ (32) HKV (8) PAGE (16) LAYERS (4) ITERS (50) CHUNKS (auto) FP8 (0)
MLA=1: latent rows (q 576 / v 512 over one kv head; HQ default 16, BS default 64), plain = the MLA decode kernel with the
native split schedule."""
import os

import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

E = lambda k, d: int(os.environ.get(k, d))  # noqa: E731
mla = E("MLA", 0)
bs, shared, uniq, hq, hkv, page = (E("BS", 64 if mla else 256), E("SHARED", 3584), E("UNIQ", 512), E("HQ", 16 if mla else 32),
                                   1 if mla else E("HKV", 8), E("PAGE", 16))
layers, iters, d = E("LAYERS", 4), E("ITERS", 50), (576 if mla else 128)
dv = 512 if mla else d
chunks = E("CHUNKS", 0) or None
fp8 = E("FP8", 0)
dev = "cuda"
rng = np.random.default_rng(0)
ctx = shared + uniq
n_pages = shared // page + bs * (-(-uniq // page)) + 2
ids = rng.permutation(np.arange(1, n_pages))
r2t = np.zeros((bs + 1, ctx), dtype=np.int32)
sh = (ids[: shared // page, None] * page + np.arange(page)[None]).reshape(-1)
pi = shared // page
for i in range(bs):
    k = -(-uniq // page)
    priv = (ids[pi: pi + k, None] * page + np.arange(page)[None]).reshape(-1)[:uniq]
    pi += k
    r2t[i + 1] = np.concatenate([sh, priv])
pool = n_pages * page
dt = torch.bfloat16
mk = lambda: torch.randn(pool, hkv, d, device=dev, dtype=dt)  # noqa: E731
kbs = [mk() for _ in range(layers)]
vbs = [k[..., :dv] for k in kbs] if mla else [mk() for _ in range(layers)]
if fp8:
    kbs = [k.to(torch.float8_e4m3fn).view(torch.uint8) for k in kbs]
    vbs = [v.to(torch.float8_e4m3fn).view(torch.uint8) for v in vbs]
q = torch.randn(bs, hq, d, device=dev, dtype=dt)
o1, o2 = (torch.zeros(bs, hq, dv, device=dev, dtype=dt) for _ in range(2))
r2t_d = torch.from_numpy(r2t).to(dev)
rpi = torch.arange(1, bs + 1, device=dev, dtype=torch.int64)
lens = torch.full((bs,), ctx, device=dev, dtype=torch.int64)
sm = 192 ** -0.5 if mla else d ** -0.5
if mla:  # plain MLA decode at its best: the native split schedule
    S = ops.native_max_kv_splits(bs, hq, 1, 256, 16)
    nsplit = torch.ones(bs, dtype=torch.int32, device=dev)
    if S > 1:
        ops.get_num_kv_splits_native(nsplit, lens.int(), hq, 1, S, 256)
    al = torch.empty(bs, hq, max(S, 1), dv, dtype=torch.float32, device=dev)
    ale = torch.empty(bs, hq, max(S, 1), dtype=torch.float32, device=dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters / layers * 1e3  # us per layer


def plain():
    for l in range(layers):
        if mla:
            ops.decode_attention_fwd_paged(q, kbs[l], vbs[l], o1, r2t_d, rpi, lens, al, ale, nsplit if S > 1 else None, S, sm,
                                           page_size=page)
        else:
            ops.decode_attention_fwd_paged(q, kbs[l], vbs[l], o1, r2t_d, rpi, lens, None, None, None, 1, sm, page_size=page)


cd = ops.CascadeDecode(bs, hq, hkv, d, dt, dev, max_shared=ctx, num_chunks=chunks, v_head_dim=dv)
cd.plan(r2t_d, rpi, lens)
print("shared_len", cd.shared_len(), "chunks", cd.num_chunks, "suffix max_kv_splits", cd.max_kv_splits)


def cascade():
    for l in range(layers):
        cd(q, kbs[l], vbs[l], o2, sm, page_size=page)


t_plain, t_casc = timed(plain), timed(cascade)
t_plan = timed(lambda: [cd.plan(r2t_d, rpi, lens) for _ in range(layers)])
print(f"plain {t_plain:.1f} us/layer  cascade {t_casc:.1f} us/layer  speedup {t_plain / t_casc:.2f}x  plan {t_plan:.1f} us/forward")
print("max |diff|", (o1.float() - o2.float()).abs().max().item())
kv_bytes = bs * ctx * hkv * d * (2 if mla else 2 * (1 if fp8 else 2))
print(f"plain effective {kv_bytes / t_plain / 1e6:.2f} TB/s; cascade effective {kv_bytes / t_casc / 1e6:.2f} TB/s")
