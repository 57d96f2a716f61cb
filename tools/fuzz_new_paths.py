"""Dev fuzz: random geometries through the paths added late in round 1, each against the plain HIP path
(itself oracle-tested): shared-prefix decode vs per-request decode; GQA-packed and split-KV verify vs the
per-head extend.  env: N (200) SEED (0)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

dev = "cuda"
N, SEED = int(os.environ.get("N", 200)), int(os.environ.get("SEED", 0))
rng = np.random.default_rng(SEED)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
worst = {"cascade": 0.0, "packed": 0.0, "split": 0.0}
for it in range(N):
    dtype = [torch.bfloat16, torch.float16][it % 2]
    hkv = int(rng.choice([1, 2, 4, 8]))
    g = int(rng.choice([1, 2, 4, 8]))
    hq, d = hkv * g, 128
    page = int(rng.choice([1, 16, 64]))
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    if it % 3 == 0:  # ---- cascade decode
        bs = int(rng.integers(1, 40))
        shared = int(rng.choice([0, 64, 200, 1024, 1500])) // page * page
        lens = shared + rng.integers(1, 300, size=bs)
        ctx = int(lens.max()) + page
        npg_sh = shared // page
        per = [-(-int(n) // page) - npg_sh for n in lens]
        ids = rng.permutation(np.arange(1, npg_sh + sum(per) + 2))
        r2t = np.zeros((bs + 1, ctx), dtype=np.int32)
        sh = (ids[:npg_sh, None] * page + np.arange(page)[None]).reshape(-1)
        pi = npg_sh
        for i in range(bs):
            pv = (ids[pi: pi + per[i], None] * page + np.arange(page)[None]).reshape(-1)
            pi += per[i]
            row = np.concatenate([sh, pv])[: int(lens[i])]
            r2t[i + 1, : len(row)] = row
        pool = (len(ids) + 1) * page
        kb = torch.randn(pool, hkv, d, generator=gen).to(dtype).to(dev)
        vb = torch.randn(pool, hkv, d, generator=gen).to(dtype).to(dev)
        q = torch.randn(bs, hq, d, generator=gen).to(dtype).to(dev)
        rpi, ln = T(np.arange(1, bs + 1, dtype=np.int64)), T(lens.astype(np.int64))
        ref, o = torch.zeros_like(q), torch.zeros_like(q)
        sm = d ** -0.5
        ops.decode_attention_fwd_paged(q, kb, vb, ref, T(r2t), rpi, ln, None, None, None, 1, sm, page_size=page)
        cd = ops.CascadeDecode(max(bs, int(rng.integers(bs, bs + 50))), hq, hkv, d, dtype, dev, max_shared=ctx,
                               min_shared=int(rng.choice([1, 64, 256])))
        cd.plan(T(r2t), rpi, ln)
        cd(q, kb, vb, o, sm, page_size=page)
        e = (o.float() - ref.float()).abs().max().item()
        worst["cascade"] = max(worst["cascade"], e)
        assert e <= 2e-2, ("cascade", it, bs, hq, hkv, page, shared, e)
    else:            # ---- verify shapes
        bs = int(rng.integers(1, 12))
        nd = int(rng.choice([1, 2, 4, 7, 8, 16]))
        prefix = rng.choice([0, 1, 63, 64, 65, 500, 1300, 4000], size=bs).astype(np.int64)
        pool = int(prefix.sum()) + 64
        kb = torch.randn(pool, hkv, d, generator=gen).to(dtype).to(dev)
        vb = torch.randn(pool, hkv, d, generator=gen).to(dtype).to(dev)
        q = torch.randn(bs * nd, hq, d, generator=gen).to(dtype).to(dev)
        ke = torch.randn(bs * nd, hkv, d, generator=gen).to(dtype).to(dev)
        ve = torch.randn(bs * nd, hkv, d, generator=gen).to(dtype).to(dev)
        kv_indptr = T(np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32))
        kv_indices = T((rng.permutation(pool - 1)[: int(prefix.sum())] + 1).astype(np.int64))
        qo = T((np.arange(bs + 1) * nd).astype(np.int64))
        rows = []
        for p in prefix:
            m = np.ones((nd, int(p) + nd), dtype=np.uint8)
            m[:, int(p):] = np.tril(rng.integers(0, 2, size=(nd, nd))) | np.eye(nd, dtype=np.int64)
            rows.append(m.reshape(-1))
        mask = T(np.concatenate(rows))
        mi = T(np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64))
        ref, o1, o2 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
        args = (kb, vb, qo, kv_indptr, kv_indices, mask, True, mi, nd, 1.0, 1.0)
        ops.extend_attention_fwd(q, ke, ve, ref, *args)
        ops.extend_attention_fwd_gqa_packed(q, ke, ve, o1, *args)
        e1 = (o1.float() - ref.float()).abs().max().item()
        worst["packed"] = max(worst["packed"], e1)
        assert e1 <= 1e-2, ("packed", it, bs, nd, hq, hkv, e1)
        if g > 1 or True:
            S = int(rng.choice([1, 2, 5, 32]))
            ops.verify_attention_splitkv(q, ke, ve, o2, kb, vb, qo, kv_indptr, kv_indices, mask, mi, nd, S, 1.0, 1.0)
            # partials are rounded to 16 bits before the merge: compare relative to the output's magnitude
            e2 = ((o2.float() - ref.float()).abs() / ref.float().abs().clamp(min=1.0)).max().item()
            worst["split"] = max(worst["split"], e2)
            # (two roundings: the 16-bit partial and the merged output -- up to ~1.5 ulp of the output)
            assert e2 <= (1.6e-2 if dtype == torch.bfloat16 else 3e-3), ("split", it, bs, nd, hq, hkv, S, prefix.tolist(), e2)
print("fuzz ok", N, "cases; worst |diff| vs the plain HIP path:", worst)
