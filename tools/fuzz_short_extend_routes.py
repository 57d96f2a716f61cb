"""Dev fuzz: short extends at the AGPR template's head dims through both routes -- the packed template (default since round 4)
and the per-(request, q head) short-extend kernels (option extend_d256_min_rows = 129) -- on random ragged batches and
GQA group sizes 1..16; the two are different kernels, so the outputs are compared at 3 ulps of the 16-bit output (+ the
bf16 P-rounding term |p|-weighted: 2^-8 of max |v|), and NaNs / untouched rows are looked for.  env: N (80) SEED (0)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import lib as rxlib  # noqa: E402
from sglang_amd import ops  # noqa: E402

dev = "cuda"
N, SEED = int(os.environ.get("N", 80)), int(os.environ.get("SEED", 0))
rng = np.random.default_rng(SEED)
worst, routes, bad, ndiff = 0.0, {}, 0, 0
for it in range(N):
    dtype = [torch.bfloat16, torch.float16][it % 2]
    dk, dv = [(256, 256), (64, 64), (192, 128), (96, 96), (192, 192)][it % 5]
    g = int(rng.choice([1, 2, 4, 8, 16]))
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * g
    bs = int(rng.choice([1, 3, 9, 33]))
    pmax = int(rng.choice([256, 700, 2100]))
    emax = int(rng.choice([1, 3, 8, 17, 32, 128 // g if g <= 8 else 8]))
    P = rng.integers(pmax // 2, pmax + 1, size=bs)
    E = rng.integers(1, emax + 1, size=bs)
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    pool = int(P.sum()) + 1
    kb = torch.randn(pool, hkv, dk, generator=gen).to(dtype).to(dev)
    vb = torch.randn(pool, hkv, dv, generator=gen).to(dtype).to(dev)
    T = int(E.sum())
    q = torch.randn(T, hq, dk, generator=gen).to(dtype).to(dev)
    ke = torch.randn(T, hkv, dk, generator=gen).to(dtype).to(dev)
    ve = torch.randn(T, hkv, dv, generator=gen).to(dtype).to(dev)
    qo = torch.from_numpy(np.concatenate([[0], np.cumsum(E)]).astype(np.int64)).to(dev)
    kvp = torch.from_numpy(np.concatenate([[0], np.cumsum(P)]).astype(np.int32)).to(dev)
    kvi = (torch.randperm(pool - 1, generator=gen) + 1).to(torch.int64).to(dev)
    outs = {}
    for name, mr in (("template", 1), ("per_head", 129)):
        with rxlib.option("extend_d256_min_rows", mr):
            o = torch.full((T, hq, dv), float("nan"), dtype=dtype, device=dev)
            ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, kvi, None, True, None, int(E.max()), 1.0, 1.0,
                                     sm_scale=dk ** -0.5, page_size=1)
            torch.cuda.synchronize()
            outs[name] = (o.float().cpu().numpy().astype(np.float64), rxlib.last_dispatch().split("<")[0])
    a, b = outs["template"][0], outs["per_head"][0]
    if np.isnan(a).any() or np.isnan(b).any():
        bad += 1
        print(f"trial {it}: NaN ({outs['template'][1]} / {outs['per_head'][1]}) dk={dk} g={g} hkv={hkv} bs={bs} P<={pmax} E<={emax}")
        continue
    mant = 8 if dtype == torch.bfloat16 else 11
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(b), 2.0 ** -14))) - (mant - 1))
    bound = 3 * ulp + (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * 4.0
    r = float((np.abs(a - b) / bound).max())
    worst = max(worst, r)
    ndiff += int((a != b).any())
    if r > 1:
        bad += 1
        print(f"trial {it}: routes differ by {r:.2f} x bound ({outs['template'][1]} / {outs['per_head'][1]}) dk={dk} g={g} hkv={hkv} bs={bs} P<={pmax} E<={emax}")
    key = outs["template"][1] + " | " + outs["per_head"][1]
    routes[key] = routes.get(key, 0) + 1
print(f"fuzz_short_extend_routes: {N} trials, {bad} problems, worst |diff| / bound {worst:.4f}, trials with any differing element {ndiff}")
print("routes (default | min_rows 129):", routes)
