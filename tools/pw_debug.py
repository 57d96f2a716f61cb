#!/usr/bin/env python3
"""Dev triage of rx::extend_pw_kernel: the same extend call through the eight-wave kernel (RX_EXT_PW=0) and the
four-wave kernel (RX_EXT_PW=2), both against an fp32 torch reference; prints where they differ."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

DEV = "cuda"


def ref(q, ke, ve, kb, vb, qo, kvp, kvi, sm):
    T, hq, d = q.shape
    hkv = ke.shape[1]
    g = hq // hkv
    out = torch.zeros(T, hq, d, dtype=torch.float32, device=q.device)
    for i in range(len(qo) - 1):
        s, e = int(qo[i]), int(qo[i + 1])
        idx = kvi[int(kvp[i]):int(kvp[i + 1])].long()
        k = torch.cat([kb[idx].float(), ke[s:e].float()])
        v = torch.cat([vb[idx].float(), ve[s:e].float()])
        P = idx.numel()
        for h in range(hq):
            sc = q[s:e, h].float() @ k[:, h // g].T * sm
            n = e - s
            mask = torch.arange(P + n, device=q.device)[None, :] <= (P + torch.arange(n, device=q.device))[:, None]
            sc = sc.masked_fill(~mask, float("-inf"))
            out[s:e, h] = torch.softmax(sc, -1) @ v[:, h // g]
    return out


def case(dtype, prefix, extend, hq, hkv, ps, seed=0, poison=False):
    d = 128
    g = torch.Generator(device=DEV).manual_seed(seed)
    bs = len(prefix)
    npg = sum(-(-p // ps) for p in prefix) + 2
    pool = npg * ps
    kb = torch.randn(pool, hkv, d, device=DEV, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, device=DEV, generator=g).to(dtype)
    perm = torch.randperm(npg - 1, device=DEV, generator=g) + 1
    kvi, kvp, pi = [], [0], 0
    for p in prefix:
        n = -(-p // ps)
        pages = perm[pi:pi + n]
        pi += n
        sl = (pages[:, None] * ps + torch.arange(ps, device=DEV)[None]).reshape(-1)[:p]
        kvi.append(sl)
        kvp.append(kvp[-1] + p)
    kvi = torch.cat(kvi).to(torch.int64) if sum(prefix) else torch.zeros(0, dtype=torch.int64, device=DEV)
    kvp = torch.tensor(kvp, dtype=torch.int32, device=DEV)
    qo = torch.tensor(np.concatenate([[0], np.cumsum(extend)]), dtype=torch.int64, device=DEV)
    T = sum(extend)
    q = torch.randn(T, hq, d, device=DEV, generator=g).to(dtype)
    ke = torch.randn(T, hkv, d, device=DEV, generator=g).to(dtype)
    ve = torch.randn(T, hkv, d, device=DEV, generator=g).to(dtype)
    sm = d ** -0.5
    want = ref(q, ke, ve, kb, vb, qo.cpu(), kvp.cpu(), kvi, sm)
    res = {}
    for mode in ("0", "2"):
        os.environ["RX_EXT_PW"] = mode
        o = torch.full((T, hq, d), float("nan"), dtype=dtype, device=DEV)
        ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, kvi, None, True, None, max(extend), 1.0, 1.0,
                                 sm_scale=sm, page_size=ps)
        torch.cuda.synchronize()
        res[mode] = o.float()
    e0 = (res["0"] - want).abs()
    e2 = (res["2"] - want).abs()
    print(f"{str(dtype):15s} prefix {prefix} extend {extend} hq {hq}/{hkv} ps {ps}: old max err {e0.max():.3e}  pw max err "
          f"{e2.nan_to_num(nan=9e9).max():.3e}  nan {int(torch.isnan(res['2']).sum())}")
    if not (e2.nan_to_num(nan=9e9).max() < 0.05):
        bad = (e2.nan_to_num(nan=9e9) > 0.05).any(-1)   # [T, hq]
        rows = bad.any(1).nonzero().flatten().tolist()
        print("   bad rows:", rows[:40], "...", len(rows), "of", T, "; bad heads:", bad.any(0).nonzero().flatten().tolist())


if __name__ == "__main__":
    for rep in range(2):
        for dt in (torch.bfloat16, torch.float16):
            case(dt, [70, 0, 257], [33, 64, 5], 8, 2, 1, seed=3)
            case(dt, [70, 0, 257], [33, 64, 5], 8, 2, 16, seed=3)
            case(dt, [0], [64], 4, 1, 1)
            case(dt, [0], [300], 4, 1, 16)
            case(dt, [64], [64], 4, 1, 16)
            case(dt, [200], [1], 4, 4, 16)
            case(dt, [1000, 300], [512, 700], 8, 2, 16)
            case(dt, [3584] * 2, [512] * 2, 8, 2, 16)
            case(dt, [4096 + 77], [300], 4, 1, 16)
