"""Dev fuzz: rx_decode_params.merge_counters (stage 2 inside the stage-1 kernel) against the two-launch form on random
geometries -- outputs must be bit-identical and the counters back at zero.  env: N (300) SEED (0)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

dev = "cuda"
N, SEED = int(os.environ.get("N", 300)), int(os.environ.get("SEED", 0))
rng = np.random.default_rng(SEED)
g = torch.Generator(device=dev).manual_seed(SEED)
n_in_kernel = 0
for it in range(N):
    dtype = [torch.bfloat16, torch.float16][it % 2]
    mla = it % 5 == 4
    S = int(rng.choice([8, 16, 24, 32]))
    if mla:
        hq, hkv, dk, dv = int(rng.choice([5, 16, 32, 128])), 1, 576, 512
        bs = int(rng.integers(1, max(2, (4 << 20) // (hq * S * dv * 4) + 2)))
    else:
        hkv = int(rng.choice([1, 2, 4, 8]))
        hq, dk = hkv * int(rng.choice([1, 2, 4, 8, 20])), int(rng.choice([64, 128]))
        dv = dk
        bs = int(rng.integers(1, 48))
    ps = int(rng.choice([1, 16, 64]))
    max_len = int(rng.choice([40, 300, 2000]))
    lens = rng.integers(0 if it % 7 == 0 else 1, max_len, size=bs)
    pages = [-(-int(n) // ps) for n in lens]
    ids = rng.permutation(np.arange(1, sum(pages) + 2))
    r2t = np.zeros((bs + 1, max_len + ps), dtype=np.int32)
    pi = 0
    for i in range(bs):
        sl = (ids[pi: pi + pages[i], None] * ps + np.arange(ps)[None]).reshape(-1)[: int(lens[i])]
        pi += pages[i]
        r2t[i + 1, : len(sl)] = sl
    pool = (len(ids) + 1) * ps
    kb = torch.randn(pool, hkv, dk, device=dev, generator=g).to(dtype)
    vb = kb[..., :dv] if mla else torch.randn(pool, hkv, dv, device=dev, generator=g).to(dtype)
    fp8 = (not mla and it % 6 == 1) or (mla and it % 10 == 9)
    if fp8:
        kb = kb.to(torch.float8_e4m3fn)
        vb = kb[..., :dv] if mla else vb.to(torch.float8_e4m3fn)
    q = torch.randn(bs, hq, dk, device=dev, generator=g).to(dtype)
    sinks = torch.randn(hq, device=dev, generator=g) if it % 4 == 3 else None
    r2td = torch.from_numpy(r2t).to(dev)
    rpi = torch.arange(1, bs + 1, device=dev)
    lens_d = torch.from_numpy(lens).to(dev)
    nsplit = torch.from_numpy(rng.integers(1, S + 1, size=bs).astype(np.int32)).to(dev)
    al = torch.zeros(bs, hq, S, dv, dtype=torch.float32, device=dev)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=dev)
    cnt = torch.zeros(bs * hq, dtype=torch.int32, device=dev)
    a = torch.full((bs, hq, dv), float("nan"), dtype=dtype, device=dev)
    b = torch.full_like(a, float("nan"))
    kw = dict(page_size=ps, sinks=sinks, logit_cap=float(rng.choice([0.0, 30.0])), v_scale=float(rng.choice([1.0, 0.5])))
    ops.decode_attention_fwd_paged(q, kb, vb, a, r2td, rpi, lens_d, al, lse, nsplit, S, dk ** -0.5, **kw)
    for rep in range(2):
        b.fill_(float("nan"))
        ops.decode_attention_fwd_paged(q, kb, vb, b, r2td, rpi, lens_d, al, lse, nsplit, S, dk ** -0.5,
                                       merge_counters=cnt, **kw)
        torch.cuda.synchronize()
        same = (a == b) | (torch.isnan(a) & torch.isnan(b))
        assert bool(same.all()), (it, rep, "mla" if mla else (hq, hkv, dk), bs, S, ps, str(dtype), fp8,
                                  (a.float() - b.float()).abs().nan_to_num(0).max().item())
        assert int(cnt.abs().sum()) == 0, (it, rep, "counters not back at zero")
    n_in_kernel += int(bs * hq * S * dv * 4 <= (4 << 20))
print(f"{N} cases bit-identical ({n_in_kernel} small enough for the in-kernel form), counters zero")
