"""Dev fuzz: the D = 128 extend launcher's forms against each other on random ragged batches -- what the gates pick by
themselves, four waves unpacked, eight waves unpacked, eight waves self-packed (forced), four waves self-packed (forced) -- for GQA 4 / 8, causal, plain
calls.  Every form walks the same tiles per row, so outputs and LSEs must agree to the last bit on ordinary data (the
fast and the boundary tile bodies differ only when a running max moves); reported: max |diff| per form, and the instance
each form dispatched.  env: N (60) SEED (0)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import lib as rxlib  # noqa: E402
from sglang_amd import ops  # noqa: E402

dev = "cuda"
N, SEED = int(os.environ.get("N", 60)), int(os.environ.get("SEED", 0))
rng = np.random.default_rng(SEED)
FORMS = {
    "default": {},
    "4w": {"ext32_autopack": 0, "ext32_small_wg": 1},
    "8w": {"ext32_autopack": 0, "ext32_small_wg": 0},
    "8w-packed": {"ext32_small_wg": 0, "ext32_pack_min_wgs": 0, "ext32_pack_min_tiles": 0},
    "4w-packed": {"ext32_small_wg": 1, "ext32_pack_min_wgs": 0, "ext32_pack_min_tiles": 0},  # (round 5: packed rows on four waves)
}
worst, picked, bad = {k: 0.0 for k in FORMS}, {}, 0
for it in range(N):
    dtype = [torch.bfloat16, torch.float16][it % 2]
    g = int(rng.choice([4, 8]))
    hkv = int(rng.choice([1, 2, 8 // g * 2]))
    hq, d = hkv * g, 128
    bs = int(rng.choice([1, 2, 5, 17, 40, 130]))
    pmax = int(rng.choice([0, 64, 300, 1500, 4000]))
    emax = int(rng.choice([1, 7, 33, 64, 130, 300, 700]))
    P = rng.integers(0, pmax + 1, size=bs) if it % 4 else np.full(bs, pmax)
    E = rng.integers(1, emax + 1, size=bs) if it % 3 else np.full(bs, emax)
    page = int(rng.choice([1, 16]))
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    pages = [-(-int(p) // page) for p in P]
    npg = sum(pages) + 2
    perm = rng.permutation(np.arange(1, npg))
    kvi, kvp, pi = [], [0], 0
    for p, n in zip(P, pages):
        sl = (perm[pi: pi + n, None] * page + np.arange(page)[None]).reshape(-1)[: int(p)]
        pi += n
        kvi.append(sl)
        kvp.append(kvp[-1] + int(p))
    kvi = np.concatenate(kvi).astype(np.int64) if kvp[-1] else np.zeros(0, dtype=np.int64)
    kb = torch.randn(npg * page, hkv, d, generator=gen).to(dtype).to(dev)
    vb = torch.randn(npg * page, hkv, d, generator=gen).to(dtype).to(dev)
    T = int(E.sum())
    q = torch.randn(T, hq, d, generator=gen).to(dtype).to(dev)
    ke = torch.randn(T, hkv, d, generator=gen).to(dtype).to(dev)
    ve = torch.randn(T, hkv, d, generator=gen).to(dtype).to(dev)
    qo = torch.from_numpy(np.concatenate([[0], np.cumsum(E)]).astype(np.int64)).to(dev)
    kvp_t = torch.tensor(kvp, dtype=torch.int32, device=dev)
    kvi_t = torch.from_numpy(kvi).to(dev)
    outs = {}
    for name, opts in FORMS.items():
        ctx = [rxlib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            o = torch.full((T, hq, d), float("nan"), dtype=dtype, device=dev)
            lse = torch.full((T, hq), float("nan"), dtype=torch.float32, device=dev)
            ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp_t, kvi_t, None, True, None, int(E.max()), 1.0, 1.0,
                                     sm_scale=d ** -0.5, lse_extend=lse, page_size=page, avg_kv_len_hint=int(P.mean()))
            torch.cuda.synchronize()
            outs[name] = (o.float(), lse, rxlib.last_dispatch())
        finally:
            for c in reversed(ctx):
                c.__exit__()
    ref = outs["4w"]
    for name, (o, lse, inst) in outs.items():
        if torch.isnan(o).any() or torch.isnan(lse).any():
            bad += 1
            print(f"trial {it}: NaN in form {name} ({inst}) bs={bs} g={g} hkv={hkv} pmax={pmax} emax={emax} page={page}")
        dd = max((o - ref[0]).abs().max().item(), (lse - ref[1]).abs().max().item())
        worst[name] = max(worst[name], dd)
        if dd > 0:
            print(f"trial {it}: form {name} ({inst}) differs from 4w by {dd:.3e}  bs={bs} g={g} hkv={hkv} pmax={pmax} emax={emax} page={page}")
    key = outs["default"][2].split("<")[1]
    picked[key] = picked.get(key, 0) + 1
print(f"fuzz_extend_forms: {N} trials, {bad} with NaN; max |diff| vs the four-wave unpacked form: {worst}")
print("instances the gates picked:", picked)
