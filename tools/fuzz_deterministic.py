#!/usr/bin/env python3
"""Dev fuzz of the deterministic-inference mode's promise (tests/test_gpu_deterministic.py holds three fixed cases): one
prompt is prefilled whole and alone, then again with a random part of it cached and random strangers in the batch, then
decoded alone and in a batch -- the extend rows and the decode row must come out the same to the last bit.  Random head
geometry (GQA 1 .. 8), head dim 128 (32x32x16 kernel) / 64 / 96 (generic kernel), page size, dtype, cut point.
env: N (24) SEED (0)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_backend import _Harness  # noqa: E402

from sglang_amd.forward_batch import ForwardBatch  # noqa: E402

DEV = "cuda"
N, SEED = int(os.environ.get("N", 24)), int(os.environ.get("SEED", 0))
rng = np.random.default_rng(SEED)


def extend(hs, rows, prefix_lens, extend_lens, q, k, v):
    seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
    loc = hs.alloc_extend(rows, list(prefix_lens), seq_lens)
    fb = ForwardBatch.for_extend(torch.tensor(rows, dtype=torch.int64, device=DEV), torch.tensor(seq_lens, device=DEV), loc,
                                 list(prefix_lens), list(extend_lens))
    hs.backend.init_forward_metadata(fb)
    return hs.layer(q, k, v, fb, hs.backend)


def decode(hs, rows, seq_lens, q):
    seq_t = torch.tensor(seq_lens, dtype=torch.int64)
    fb = ForwardBatch.for_decode(torch.tensor(rows, dtype=torch.int64, device=DEV), seq_t.to(DEV),
                                 torch.zeros(len(rows), dtype=torch.int64, device=DEV), seq_t)
    hs.backend.init_forward_metadata(fb)
    return hs.layer(q, None, None, fb, hs.backend, save_kv_cache=False)


for it in range(N):
    dtype = [torch.bfloat16, torch.float16][it % 2]
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * int(rng.choice([1, 2, 4, 8]))
    d = int(rng.choice([128, 128, 64, 96]))
    ps = int(rng.choice([1, 16, 32]))
    L = int(rng.integers(40, 900))
    cut = int(rng.integers(1, L))
    mk = lambda: _Harness(ps, hq, hkv, d, dtype, "shuffled_pages" if ps > 1 else "contiguous", ["paged", "indices"][it % 2],  # noqa: E731
                          max_ctx=2048, size=16384, server_args_extra={"enable_deterministic_inference": True})
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    q = torch.randn(L, hq * d, generator=g).to(dtype).to(DEV)
    k = torch.randn(L, hkv * d, generator=g).to(dtype).to(DEV)
    v = torch.randn(L, hkv * d, generator=g).to(dtype).to(DEV)
    qd = torch.randn(1, hq * d, generator=g).to(dtype).to(DEV)
    hs = mk()
    r = hs.r2t.alloc(1)
    o_whole = extend(hs, r, [0], [L], q, k, v)
    o_dec = decode(hs, r, [L], qd)
    # again: `cut` tokens cached, strangers around
    hs2 = mk()
    n_other = int(rng.integers(0, 4))
    rows = hs2.r2t.alloc(1 + n_other)
    me = int(rng.integers(0, 1 + n_other))
    others = [i for i in range(1 + n_other) if i != me]
    loc = hs2.alloc_extend([rows[me]], [0], [cut])
    hs2.pool.set_kv_buffer(hs2.layer, loc, k[:cut].view(cut, hkv, d), v[:cut].view(cut, hkv, d))
    o_pre = [int(rng.integers(0, 300)) for _ in others]
    if sum(o_pre):
        hs2.fill_prefix([rows[i] for i in others], o_pre)
    pre, ext = [0] * (1 + n_other), [0] * (1 + n_other)
    pre[me], ext[me] = cut, L - cut
    for i, p_ in zip(others, o_pre):
        pre[i], ext[i] = p_, int(rng.integers(1, 200))
    qs, ks, vs = [], [], []
    for i in range(1 + n_other):
        if i == me:
            qs.append(q[cut:]); ks.append(k[cut:]); vs.append(v[cut:])
        else:
            qs.append(hs2.rand(ext[i], hq * d)); ks.append(hs2.rand(ext[i], hkv * d)); vs.append(hs2.rand(ext[i], hkv * d))
    o2 = extend(hs2, rows, pre, ext, torch.cat(qs), torch.cat(ks), torch.cat(vs))
    off = sum(ext[:me])
    mine = o2[off: off + L - cut]
    assert torch.equal(mine.view(torch.int16), o_whole[cut:].view(torch.int16)), (
        it, "extend", dict(hq=hq, hkv=hkv, d=d, ps=ps, L=L, cut=cut, me=me, pre=pre, ext=ext), (mine.float() - o_whole[cut:].float()).abs().max().item())
    seq2 = [p + e for p, e in zip(pre, ext)]
    qd2 = torch.cat([qd if i == me else hs2.rand(1, hq * d) for i in range(1 + n_other)])
    o_dec2 = decode(hs2, rows, seq2, qd2)
    assert torch.equal(o_dec2[me].view(torch.int16), o_dec[0].view(torch.int16)), (it, "decode", dict(hq=hq, hkv=hkv, d=d, ps=ps, L=L, seq2=seq2))
print(f"fuzz_deterministic ok: {N} prompts, extend rows and decode rows bit-identical alone / cut + batched")
