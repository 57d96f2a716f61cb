"""Randomised differential tests: seeded random shapes, lengths, page sizes, layouts and options through
the C ABI vs the fp64 oracle.  Complements the fixed case matrices: every draw exercises a different mix of
ragged lengths (incl. 0 and tile-boundary values), GQA ratios, index dtypes, splits, masks and scales."""
import numpy as np
import pytest

import parity_util as parity
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_amd import ops as _ops

    return _ops


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _paged(rng, lens, page_size):
    pages_per_req = [max(1, (int(n) + page_size - 1) // page_size) for n in lens]
    n_pages = sum(pages_per_req) + 2
    page_ids = rng.permutation(np.arange(1, n_pages))
    r2t = np.zeros((len(lens) + 1, int(max(max(lens), 1)) + page_size), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        sl = np.concatenate([np.arange(p * page_size, (p + 1) * page_size)
                             for p in page_ids[pi: pi + pages_per_req[i]]])
        pi += pages_per_req[i]
        r2t[i + 1, : int(n)] = sl[: int(n)]
    return r2t, n_pages * page_size


LEN_POOL = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 500, 777, 1025]


@pytest.mark.parametrize("seed", range(12))
def test_random_decode(ops, seed):
    rng = np.random.default_rng(1000 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    d = int(rng.choice([64, 128, 128, 80]))
    hkv = int(rng.choice([1, 2, 4, 8]))
    hq = hkv * int(rng.choice([1, 2, 4, 8]))
    page_size = int(rng.choice([1, 4, 16, 32, 64]))
    bs = int(rng.integers(1, 9))
    lens = rng.choice(LEN_POOL, size=bs).astype(np.int64)
    r2t, pool = _paged(rng, lens, page_size)
    g = torch.Generator().manual_seed(seed)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    ks, vs = float(rng.choice([1.0, 0.7])), float(rng.choice([1.0, 1.3]))
    cap = float(rng.choice([0.0, 0.0, 30.0]))
    sinks = torch.randn(hq, generator=g) if seed % 3 == 0 else None
    sm = d ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want = orc.decode_attention(_bits(q), _bits(kb), _bits(vb), kv_indptr, kv_indices, sm, k_scale=ks, v_scale=vs,
                                logit_cap=cap, sinks=None if sinks is None else sinks.numpy())
    absw = orc.decode_attention(_bits(q), _bits(kb), parity.abs_values(_bits(vb)), kv_indptr, kv_indices, sm, k_scale=ks,
                                v_scale=vs, logit_cap=cap, sinks=None if sinks is None else sinks.numpy())
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    qd, kbd, vbd = q.to(DEV), kb.to(DEV), vb.to(DEV)
    sd = None if sinks is None else sinks.to(DEV)
    # native paged walk, single pass
    o = torch.zeros(bs, hq, d, dtype=dtype, device=DEV)
    ops.decode_attention_fwd_paged(qd, kbd, vbd, o, T(r2t), T(rpi if seed % 2 else rpi.astype(np.int32)), T(lens),
                                   None, None, None, 1, sm, ks, vs, cap, sd, page_size=page_size)
    parity.check_out(o.float().cpu().numpy(), want, dtype, "paged/single", absw=absw)   # the north star's element-wise bound
    # reference contract: kv_indices (int32 or int64) + K3 splits + stage 2
    S = int(rng.choice([2, 4, 8, 16]))
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, T(lens).int(), hq, hkv, S, int(rng.choice([64, 256])))
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o2 = torch.zeros_like(o)
    kvi = T(kv_indices if seed % 2 else kv_indices.astype(np.int32))
    ops.decode_attention_fwd(qd, kbd, vbd, o2, T(kv_indptr), kvi, al, lse, nsplit, S, sm, ks, vs, logit_cap=cap,
                             sinks=sd, page_size=page_size)
    parity.check_out(o2.float().cpu().numpy(), want, dtype, "indices/split", absw=absw)


@pytest.mark.parametrize("seed", range(12))
def test_random_extend(ops, seed):
    rng = np.random.default_rng(2000 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    d = int(rng.choice([64, 128, 128, 128, 96]))
    hkv = int(rng.choice([1, 2, 4]))
    hq = hkv * int(rng.choice([1, 2, 4]))
    page_size = int(rng.choice([1, 16, 32]))
    bs = int(rng.integers(1, 6))
    prefix = rng.choice([0, 0, 1, 17, 64, 65, 130, 300], size=bs).astype(np.int64)
    ext = rng.choice([1, 2, 31, 32, 33, 64, 100, 257], size=bs).astype(np.int64)
    r2t, pool = _paged(rng, np.maximum(prefix, 1), page_size)
    g = torch.Generator().manual_seed(seed)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    T_ = int(ext.sum())
    q = torch.randn(T_, hq, d, generator=g).to(dtype)
    ke = torch.randn(T_, hkv, d, generator=g).to(dtype)
    ve = torch.randn(T_, hkv, d, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, prefix)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    causal = bool(seed % 4 != 3)
    window = int(rng.choice([-1, -1, 40]))
    cap = float(rng.choice([0.0, 0.0, 25.0]))
    ks, vs = float(rng.choice([1.0, 0.8])), float(rng.choice([1.0, 1.2]))
    sinks = torch.randn(hq, generator=g) if seed % 3 == 1 else None
    sm = d ** -0.5
    want, want_lse = orc.extend_attention(_bits(q), _bits(ke), _bits(ve), _bits(kb), _bits(vb), qo, kv_indptr,
                                          kv_indices, is_causal=causal, sm_scale=sm, k_scale=ks, v_scale=vs,
                                          logit_cap=cap, sliding_window_size=window,
                                          sinks=None if sinks is None else sinks.numpy(), return_lse=True)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    o = torch.zeros_like(q, device=DEV)
    lse = torch.zeros(T_, hq, dtype=torch.float32, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV),
                             T(qo if seed % 2 else qo.astype(np.int32)), T(kv_indptr),
                             T(kv_indices if seed % 2 else kv_indices.astype(np.int32)), None, causal, None,
                             int(ext.max()), ks, vs, sm_scale=sm, logit_cap=cap, sliding_window_size=window,
                             sinks=None if sinks is None else sinks.to(DEV), lse_extend=lse, page_size=page_size)
    if d == 128 and hq > hkv and sinks is None:  # GQA-packed query rows: the same numbers, row for row
        o2 = torch.zeros_like(o)
        ops.extend_attention_fwd_gqa_packed(q.to(DEV), ke.to(DEV), ve.to(DEV), o2, kb.to(DEV), vb.to(DEV),
                                            T(qo if seed % 2 else qo.astype(np.int32)), T(kv_indptr),
                                            T(kv_indices if seed % 2 else kv_indices.astype(np.int32)), None, causal,
                                            None, int(ext.max()), ks, vs, sm_scale=sm, logit_cap=cap,
                                            sliding_window_size=window, page_size=page_size)
        ok = torch.isfinite(o.float()).all(dim=-1) & torch.isfinite(o2.float()).all(dim=-1)
        assert (o.float()[ok] - o2.float()[ok]).abs().max().item() <= 2e-2
    got = o.float().cpu().numpy().astype(np.float64)
    seen = np.isfinite(want_lse)  # a window can hide everything from a row: 0/0 in the reference
    absw = orc.extend_attention(_bits(q), _bits(ke), parity.abs_values(_bits(ve)), _bits(kb), parity.abs_values(_bits(vb)),
                                qo, kv_indptr, kv_indices, is_causal=causal, sm_scale=sm, k_scale=ks, v_scale=vs, logit_cap=cap,
                                sliding_window_size=window, sinks=None if sinks is None else sinks.numpy())
    parity.check_out(got[seen], want[seen], dtype, "random extend", absw=absw[seen])   # the north star's element-wise bound
    np.testing.assert_allclose(lse.cpu().numpy()[seen], want_lse[seen], atol=3e-3, rtol=1e-3)


@pytest.mark.parametrize("seed", range(12))
def test_random_extend_wide_heads(ops, seed):
    """Random ragged batches on the AGPR-accumulator extend kernels (rx_extend_d256.hip at 256 / 256, 192 / 192,
    192 / 128 and rx_extend_mla.hip at the latent shape 576 / 512): random page sizes, GQA groups, causal or not,
    k / v scales, int32 / int64 indices, long and short extends, the new tokens' v aliased or not (latent shape)."""
    rng = np.random.default_rng(7000 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    dk, dv, mla = [(256, 256, False), (192, 128, False), (192, 192, False), (576, 512, True)][seed % 4]
    hkv = 1 if mla else int(rng.choice([1, 2]))
    hq = hkv * int(rng.choice([8, 16] if mla else [2, 4, 8]))
    page_size = int(rng.choice([1, 16, 32]))
    bs = int(rng.integers(1, 10))
    prefix = rng.choice([0, 0, 1, 17, 64, 65, 130, 300, 700], size=bs).astype(np.int64)
    ext = rng.choice([1, 2, 31, 64, 100, 129, 257], size=bs).astype(np.int64)
    r2t, pool = _paged(rng, np.maximum(prefix, 1), page_size)
    g = torch.Generator().manual_seed(seed)
    kb = (torch.randn(pool, hkv, dk, generator=g) * 0.6).to(dtype)
    vb = kb[..., :dv] if mla else torch.randn(pool, hkv, dv, generator=g).to(dtype)
    T_ = int(ext.sum())
    q = torch.randn(T_, hq, dk, generator=g).to(dtype)
    ke = (torch.randn(T_, hkv, dk, generator=g) * 0.6).to(dtype)
    own_v = (not mla) or bool(rng.integers(0, 2))
    ve = torch.randn(T_, hkv, dv, generator=g).to(dtype) if own_v else ke[..., :dv]
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, prefix)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    causal = bool(seed % 5 != 4)
    ks, vs = float(rng.choice([1.0, 0.8])), float(rng.choice([1.0, 1.2]))
    sm = (192 if mla else dk) ** -0.5
    want, want_lse = orc.extend_attention(_bits(q), _bits(ke), _bits(ve), _bits(kb), _bits(vb.contiguous()), qo, kv_indptr,
                                          kv_indices, is_causal=causal, sm_scale=sm, k_scale=ks, v_scale=vs, return_lse=True)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    o = torch.zeros(T_, hq, dv, dtype=dtype, device=DEV)
    lse = torch.zeros(T_, hq, dtype=torch.float32, device=DEV)
    kbd, ked = kb.to(DEV), ke.to(DEV)
    vbd = kbd[..., :dv] if mla else vb.to(DEV)
    ved = ve.to(DEV) if own_v else ked[..., :dv]
    ops.extend_attention_fwd(q.to(DEV), ked, ved, o, kbd, vbd, T(qo if seed % 2 else qo.astype(np.int32)), T(kv_indptr),
                             T(kv_indices if seed % 3 else kv_indices.astype(np.int32)), None, causal, None,
                             int(ext.max()), ks, vs, sm_scale=sm, lse_extend=lse, page_size=page_size)
    absw = orc.extend_attention(_bits(q), _bits(ke), parity.abs_values(_bits(ve)), _bits(kb), parity.abs_values(_bits(vb.contiguous())),
                                qo, kv_indptr, kv_indices, is_causal=causal, sm_scale=sm, k_scale=ks, v_scale=vs)
    parity.check_out(o.float().cpu().numpy().astype(np.float64), want, dtype, ("random wide heads", dk, dv), absw=absw)
    np.testing.assert_allclose(lse.cpu().numpy(), want_lse, atol=3e-3, rtol=1e-3)


@pytest.mark.parametrize("seed", range(6))
def test_random_byte_and_index_kernels(ops, seed):
    """store / kv-index build / move: bit-exact vs the oracle on random shapes and dtypes."""
    rng = np.random.default_rng(3000 + seed)
    g = torch.Generator().manual_seed(seed)
    # K1 store with strided sources, random loc incl. the reserved slot
    n, row = int(rng.integers(1, 300)), int(rng.choice([8, 64, 128, 1024]))
    pool = n + 50
    k = torch.randn(n, row * 2, generator=g).to(torch.bfloat16)[:, :row]   # row-strided view
    v = torch.randn(n, row, generator=g).to(torch.bfloat16)
    loc = rng.permutation(pool - 1)[:n] + 1
    loc[rng.integers(0, n)] = 0
    kc = torch.zeros(pool, row, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros(pool, row, dtype=torch.bfloat16, device=DEV)
    loc_t = torch.from_numpy(loc.astype(np.int64 if seed % 2 else np.int32)).to(DEV)
    ops.store_cache(k.to(DEV), v.to(DEV), kc, vc, loc_t)
    wk, wv = np.zeros((pool, row), np.uint16), np.zeros((pool, row), np.uint16)
    orc.store_kv(_bits(k.contiguous()), _bits(v), wk, wv, loc)
    assert np.array_equal(_bits(kc), wk) and np.array_equal(_bits(vc), wv)
    # K2 kv-index build with kv_start
    bs, ctx = int(rng.integers(1, 40)), int(rng.integers(8, 700))
    r2t = rng.integers(1, 1 << 20, size=(bs + 3, ctx)).astype(np.int32)
    rpi = rng.permutation(bs + 3)[:bs].astype(np.int64)
    start = rng.integers(0, ctx // 2, size=bs).astype(np.int32)
    lens = np.array([rng.integers(0, ctx - s + 1) for s in start], dtype=np.int64)
    want_p, want_i = orc.build_kv_indices(r2t, rpi, lens, kv_start=start)
    kvp = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    kvi = torch.zeros(max(int(lens.sum()), 1), dtype=torch.int64 if seed % 2 else torch.int32, device=DEV)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    ops.build_kv_indices(T(r2t), T(rpi), T(lens), kvp, kvi, T(start))
    assert np.array_equal(kvp.cpu().numpy(), want_p)
    assert np.array_equal(kvi.cpu().numpy()[: int(lens.sum())].astype(np.int64), want_i)
