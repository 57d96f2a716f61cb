#!/usr/bin/env python3
"""Dev fuzz: score_mod = relative_bias_score_mod on random ragged batches against the fp64 oracle, at the parity bar
(tests/parity_util.check_out): two-stage and unified extends (D 128 on the 32x32x16 kernel, other head dims on the generic
one), with and without a sliding window / logit cap / fp32 aux, and decode (D 64 / 128 biased MFMA instances, other dims
generic; split KV).  env: N (60) SEED (0)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as parity  # noqa: E402
from oracle import radix_oracle as orc  # noqa: E402
from sglang_amd import lib as rxlib  # noqa: E402
from sglang_amd import ops  # noqa: E402

dev = "cuda"
N, SEED = int(os.environ.get("N", 60)), int(os.environ.get("SEED", 0))
rng = np.random.default_rng(SEED)


def bits(t):
    return t.detach().cpu().contiguous().view(torch.uint16).numpy() if t.dtype == torch.bfloat16 else t.detach().cpu().numpy()


T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
worst, picked = 0.0, {}
for it in range(N):
    dtype = [torch.bfloat16, torch.float16][it % 2]
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * int(rng.choice([1, 2, 4, 8]))
    d = int(rng.choice([128, 128, 64, 80]))
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    extent = int(rng.choice([1, 3, 17, 64, 100, 200]))
    aux_dt = torch.float32 if rng.random() < 0.5 else dtype
    sm = d ** -0.5
    if it % 3 == 2:  # ---- decode
        bs = int(rng.integers(1, 6))
        lens = rng.integers(1, 700, size=bs)
        pool = int(lens.sum()) + 9
        kb = torch.randn(pool, hkv, d, generator=gen).to(dtype)
        vb = torch.randn(pool, hkv, d, generator=gen).to(dtype)
        q = torch.randn(bs, hq, d, generator=gen).to(dtype)
        aux = (1.5 * torch.randn(bs, hq, extent, generator=gen)).to(aux_dt)
        kvp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        kvi = (rng.permutation(pool - 1)[: int(lens.sum())] + 1).astype(np.int64)
        want, absw = parity.want_and_absw(orc.decode_attention, (bits(q), bits(kb), bits(vb), kvp, kvi, sm), (2,), score_bias=aux.double().numpy())
        S = int(rng.choice([1, 4, 8]))
        o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=dev)
        al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=dev)
        ls = torch.zeros(bs, hq, S, dtype=torch.float32, device=dev)
        ns = torch.from_numpy(np.minimum(S, np.maximum(1, lens // 64)).astype(np.int32)).to(dev)
        ops.decode_attention_fwd(q.to(dev), kb.to(dev), vb.to(dev), o, T_(kvp), T_(kvi), al, ls, ns, S, sm, 1.0, 1.0,
                                 score_mod=ops.relative_bias_score_mod, aux_tensors=[aux.to(dev)])
    else:  # ---- extend: two-stage or unified
        bs = int(rng.integers(1, 5))
        pre = rng.integers(0, 400, size=bs) * (rng.random(bs) < 0.8)
        ext = rng.integers(1, 300, size=bs)
        T = int(ext.sum())
        unified = it % 3 == 1
        window = int(rng.choice([-1, -1, 20, 150]))
        cap = float(rng.choice([0.0, 0.0, 30.0]))
        tot = pre + ext
        pool = int(tot.sum()) + 9
        kb = torch.randn(pool, hkv, d, generator=gen).to(dtype)
        vb = torch.randn(pool, hkv, d, generator=gen).to(dtype)
        q = torch.randn(T, hq, d, generator=gen).to(dtype)
        aux = (1.5 * torch.randn(T, hq, extent, generator=gen)).to(aux_dt)
        qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
        perm = rng.permutation(pool - 1) + 1
        o = torch.full((T, hq, d), float("nan"), dtype=dtype, device=dev)
        kw = dict(sm_scale=sm, logit_cap=cap, sliding_window_size=window)
        if unified:
            kvp = np.concatenate([[0], np.cumsum(tot)]).astype(np.int32)
            kvi = perm[: int(tot.sum())].astype(np.int64)
            want, absw = parity.want_and_absw(orc.extend_attention_unified, (bits(q), bits(kb), bits(vb), qo, kvp, kvi, pre), (2,),
                                              score_bias=aux.double().numpy(), **kw)
            ops.extend_attention_fwd_unified(q.to(dev), o, kb.to(dev), vb.to(dev), 1.0, 1.0, T_(qo), T_(kvp), T_(kvi),
                                             T_(pre.astype(np.int32)), int(ext.max()), score_mod=ops.relative_bias_score_mod,
                                             aux_tensors=[aux.to(dev)], **kw)
        else:
            ke = torch.randn(T, hkv, d, generator=gen).to(dtype)
            ve = torch.randn(T, hkv, d, generator=gen).to(dtype)
            kvp = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
            kvi = perm[: int(pre.sum())].astype(np.int64)
            want, absw = parity.want_and_absw(orc.extend_attention, (bits(q), bits(ke), bits(ve), bits(kb), bits(vb), qo, kvp, kvi), (2, 4),
                                              is_causal=True, score_bias=aux.double().numpy(), **kw)
            ops.extend_attention_fwd(q.to(dev), ke.to(dev), ve.to(dev), o, kb.to(dev), vb.to(dev), T_(qo), T_(kvp), T_(kvi), None, True,
                                     None, int(ext.max()), 1.0, 1.0, score_mod=ops.relative_bias_score_mod,
                                     aux_tensors=[aux.to(dev)], **kw)
    torch.cuda.synchronize()
    name = rxlib.last_dispatch().split("<")[0]
    picked[name] = picked.get(name, 0) + 1
    got = o.float().cpu().numpy()
    live = np.isfinite(want).all(axis=-1) & (np.abs(want).sum(axis=-1) > 0)  # (rows that see nothing: 0/0 in the reference)
    assert np.isfinite(got[live]).all(), (it, "nan")
    parity.check_out(got[live], want[live], dtype, ("fuzz score bias", it, name), ulps=1, absw=absw[live])
print(f"fuzz_score_bias ok: {N} cases at the parity bar; kernels: {picked}")
