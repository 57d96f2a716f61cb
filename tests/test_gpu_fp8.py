"""GPU parity for fp8 e4m3fn KV pools (--kv-cache-dtype fp8_e4m3 and the MLA fp8 latent rows of
BASELINE config 5): the quantising store is bit-exact against the oracle's restatement of the
reference's torch casts; decode / extend / MLA decode read the fp8 bytes, upcast them exactly and
are compared with the fp64 oracle run on the dequantised pool.
"""
import numpy as np
import pytest

import parity_util as parity
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"
FP8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    from sglang_amd import ops as _ops

    return _ops


def _bits(t):  # 16-bit tensor -> what the oracle eats
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _f32(t):
    return t.detach().float().cpu().numpy()


def _dq(u8):  # pool bytes -> float32 values (oracle decode)
    return orc.fp8_e4m3fn_decode(u8.detach().cpu().numpy())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("layout", ["nhd", "hnd", "mla"])
@pytest.mark.parametrize("scales", [(1.0, 1.0), (0.37, 2.5)])
def test_store_fp8_bit_exact(ops, dtype, layout, scales):
    g = torch.Generator().manual_seed(7)
    n, hkv, dk, dv, page = 77, (1 if layout == "mla" else 4), (512 if layout == "mla" else 128), (
        64 if layout == "mla" else 128), 16
    k = (torch.randn(n, hkv * dk, generator=g) * 3).to(dtype)
    v = (torch.randn(n, hkv * dv, generator=g) * 3).to(dtype)
    k[0, :4] = torch.tensor([0.0, -0.0, 448.0, -300.0]).to(dtype)
    npages = 12
    slots = npages * page
    loc = torch.randperm(slots - 1, generator=g)[:n] + 1
    loc[5] = 0  # reserved slot: skipped for the MHA pools
    ks, vs = scales
    if layout == "mla":
        buf = torch.zeros(slots, 1, dk + dv, dtype=torch.uint8, device=DEV)
        lay = ops._kv_layout(buf[..., :dk], buf[..., dk:], 1)
        skip = -1
    elif layout == "nhd":
        kb = torch.zeros(slots, hkv, dk, dtype=torch.uint8, device=DEV)
        vb = torch.zeros(slots, hkv, dv, dtype=torch.uint8, device=DEV)
        lay = ops._kv_layout(kb, vb, page)
        skip = 0
    else:
        kb = torch.zeros(npages, hkv, page, dk, dtype=torch.uint8, device=DEV)
        vb = torch.zeros(npages, hkv, page, dv, dtype=torch.uint8, device=DEV)
        lay = ops.kv_layout_hnd(kb, vb)
        skip = 0
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.store_cache_fp8(k.to(DEV), v.to(DEV), lay, loc.to(DEV), hkv, dk, dv, size_limit=slots, k_scale=ks,
                        v_scale=vs, reserved_skip_index=skip, err_flag=err)
    assert int(err.item()) == 0
    is_bf = dtype == torch.bfloat16
    wk = orc.quantize_kv_fp8(_f32(k), ks, is_bf).reshape(n, hkv, dk)
    wv = orc.quantize_kv_fp8(_f32(v), vs, is_bf).reshape(n, hkv, dv)
    locn = loc.numpy()
    if layout == "mla":
        got = buf.cpu().numpy()
        want = np.zeros_like(got)
        want[locn, 0, :dk], want[locn, 0, dk:] = wk[:, 0], wv[:, 0]
        assert np.array_equal(got, want)
        return
    if layout == "nhd":
        gk, gv = kb.cpu().numpy(), vb.cpu().numpy()
    else:  # [pages, H, page, D] -> [slots, H, D]
        gk = kb.cpu().numpy().transpose(0, 2, 1, 3).reshape(slots, hkv, dk)
        gv = vb.cpu().numpy().transpose(0, 2, 1, 3).reshape(slots, hkv, dv)
    wantk, wantv = np.zeros_like(gk), np.zeros_like(gv)
    keep = locn != 0
    wantk[locn[keep]], wantv[locn[keep]] = wk[keep], wv[keep]
    assert np.array_equal(gk, wantk) and np.array_equal(gv, wantv)


def test_store_fp8_errors(ops):
    k = torch.zeros(2, 128, dtype=torch.bfloat16, device=DEV)
    kb = torch.zeros(8, 1, 128, dtype=torch.bfloat16, device=DEV)  # not an fp8 pool
    from sglang_amd.lib import RadixHipError

    with pytest.raises(RadixHipError, match="fp8"):
        ops.store_cache_fp8(k, k, ops._kv_layout(kb, kb, 1), torch.tensor([1, 2], device=DEV), 1, 128, 128,
                            size_limit=8)
    kb8 = torch.zeros(8, 1, 128, dtype=torch.uint8, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.store_cache_fp8(k, k, ops._kv_layout(kb8, kb8, 1), torch.tensor([1, 99], device=DEV), 1, 128, 128,
                        size_limit=8, err_flag=err)
    assert int(err.item()) == 1  # slot 99 dropped and flagged


def _paged(rng, lens, page_size, max_extra=0):
    pages_per_req = [(int(n) + page_size - 1) // page_size for n in lens]
    n_pages = sum(pages_per_req) + 3
    page_ids = rng.permutation(np.arange(1, n_pages))
    r2t = np.zeros((len(lens) + 1, int(max(lens)) + page_size + max_extra), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        sl = np.concatenate([np.arange(p * page_size, (p + 1) * page_size)
                             for p in page_ids[pi: pi + pages_per_req[i]]])
        pi += pages_per_req[i]
        r2t[i + 1, : int(n)] = sl[: int(n)]
    return r2t, n_pages * page_size


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,hkv,d,page_size", [(32, 8, 128, 16), (8, 1, 128, 1), (4, 4, 64, 32)])
def test_decode_fp8_pool_vs_oracle(ops, dtype, hq, hkv, d, page_size):
    rng = np.random.default_rng(hq + d)
    lens = np.array([1, 31, 32, 33, 257, 500, 64], dtype=np.int64)
    bs = len(lens)
    r2t, pool = _paged(rng, lens, page_size)
    g = torch.Generator().manual_seed(3)
    kb = torch.randn(pool, hkv, d, generator=g).to(FP8)
    vb = torch.randn(pool, hkv, d, generator=g).to(FP8)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    sm, ks, vs = d ** -0.5, 0.8, 1.25
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want = orc.decode_attention(_bits(q), _dq(kb.view(torch.uint8)), _dq(vb.view(torch.uint8)), kv_indptr,
                                kv_indices, sm, k_scale=ks, v_scale=vs)
    # (the fp8 rows are upcast EXACTLY: the result carries the one output rounding of a 16-bit pool's -- 1 ulp, with the
    # P-rounding term of parity_util.check_out)
    absw = orc.decode_attention(_bits(q), _dq(kb.view(torch.uint8)), np.abs(_dq(vb.view(torch.uint8))), kv_indptr,
                                kv_indices, sm, k_scale=ks, v_scale=vs)
    qd, kbd, vbd = q.to(DEV), kb.to(DEV), vb.to(DEV)
    o = torch.zeros(bs, hq, d, dtype=dtype, device=DEV)
    T = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    ops.decode_attention_fwd_paged(qd, kbd, vbd, o, T(r2t), T(rpi), T(lens), None, None, None, 1, sm, ks, vs,
                                   page_size=page_size)
    parity.check_out(_f32(o), want, dtype, "fp8 pool / single", ulps=1, absw=absw)
    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, T(lens).int(), hq, hkv, S, 256)
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o2 = torch.zeros_like(o)
    ops.decode_attention_fwd(qd, kbd.view(torch.uint8), vbd.view(torch.uint8), o2, T(kv_indptr), T(kv_indices),
                             al, lse, nsplit, S, sm, ks, vs, page_size=page_size)
    parity.check_out(_f32(o2), want, dtype, "fp8 pool / split", ulps=1, absw=absw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,page_size", [(16, 1), (128, 16)])
@pytest.mark.parametrize("form", ["k_split", "t64"])   # option decode_mla8_t64: the 32-token K-split kernel / the 64-token-tile kernel
def test_decode_mla_fp8_rows_vs_oracle(ops, dtype, hq, page_size, form):
    from sglang_amd import lib as rxlib
    with rxlib.option("decode_mla8_t64", int(form == "t64")):
        _decode_mla_fp8_rows_vs_oracle(ops, dtype, hq, page_size)
        want_kernel = "decode_mla8_t64_kernel" if form == "t64" else "decode_mla8_dma_kernel"
        assert want_kernel in rxlib.last_dispatch(), rxlib.last_dispatch()


def _decode_mla_fp8_rows_vs_oracle(ops, dtype, hq, page_size):
    rng = np.random.default_rng(hq)
    lens = np.array([1, 31, 32, 33, 700, 64, 2049], dtype=np.int64)
    bs = len(lens)
    r2t, pool = _paged(rng, lens, page_size)
    g = torch.Generator().manual_seed(hq)
    kv = torch.randn(pool, 1, 576, generator=g).to(FP8)
    q = torch.randn(bs, hq, 576, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    sm = (128 + 64) ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    kvn = _dq(kv.view(torch.uint8))
    want = orc.decode_attention(_bits(q), kvn, kvn[..., :512], kv_indptr, kv_indices, sm)
    absw = orc.decode_attention(_bits(q), kvn, np.abs(kvn[..., :512]), kv_indptr, kv_indices, sm)
    kvd, qd = kv.to(DEV), q.to(DEV)
    T = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    o = torch.zeros(bs, hq, 512, dtype=dtype, device=DEV)
    ops.decode_attention_fwd_paged(qd, kvd, kvd[..., :512], o, T(r2t), T(rpi), T(lens), None, None, None, 1, sm,
                                   page_size=page_size)
    parity.check_out(_f32(o), want, dtype, "mla fp8 rows / single", absw=absw)   # the north star's element-wise bound
    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, T(lens).int(), hq, 1, S, 256)
    al = torch.zeros(bs, hq, S, 512, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o2 = torch.zeros_like(o)
    ops.decode_attention_fwd(qd, kvd, kvd[..., :512], o2, T(kv_indptr), T(kv_indices), al, lse, nsplit, S, sm,
                             1.0, 1.0, page_size=page_size)
    parity.check_out(_f32(o2), want, dtype, "mla fp8 rows / split", absw=absw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("page_size", [1, 16])
def test_extend_fp8_prefix_pool_vs_oracle(ops, dtype, page_size):
    rng = np.random.default_rng(page_size)
    hq, hkv, d = 8, 2, 128
    prefix = np.array([0, 17, 64, 300, 129], dtype=np.int64)
    ext = np.array([5, 64, 33, 257, 1], dtype=np.int64)
    bs = len(ext)
    r2t, pool = _paged(rng, np.maximum(prefix, 1), page_size)
    g = torch.Generator().manual_seed(11)
    kb = torch.randn(pool, hkv, d, generator=g).to(FP8)
    vb = torch.randn(pool, hkv, d, generator=g).to(FP8)
    T_ = int(ext.sum())
    q = torch.randn(T_, hq, d, generator=g).to(dtype)
    ke = torch.randn(T_, hkv, d, generator=g).to(dtype)
    ve = torch.randn(T_, hkv, d, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, prefix)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    sm, ks, vs = d ** -0.5, 0.9, 1.1
    want = orc.extend_attention(_bits(q), _bits(ke), _bits(ve), _dq(kb.view(torch.uint8)),
                                _dq(vb.view(torch.uint8)), qo, kv_indptr, kv_indices, is_causal=True,
                                sm_scale=sm, k_scale=ks, v_scale=vs)
    T = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    o = torch.zeros_like(q, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), T(qo), T(kv_indptr),
                             T(kv_indices), None, True, None, int(ext.max()), ks, vs, sm_scale=sm,
                             page_size=page_size)
    absw = orc.extend_attention(_bits(q), _bits(ke), parity.abs_values(_bits(ve)), _dq(kb.view(torch.uint8)),
                                np.abs(_dq(vb.view(torch.uint8))), qo, kv_indptr, kv_indices, is_causal=True,
                                sm_scale=sm, k_scale=ks, v_scale=vs)
    parity.check_out(_f32(o), want, dtype, "fp8 prefix pool", ulps=1, absw=absw)
    # GQA-packed query rows over the same fp8 prefix pool: the same result
    o2 = torch.zeros_like(o)
    ops.extend_attention_fwd_gqa_packed(q.to(DEV), ke.to(DEV), ve.to(DEV), o2, kb.to(DEV), vb.to(DEV), T(qo),
                                        T(kv_indptr), T(kv_indices), None, True, None, int(ext.max()), ks, vs,
                                        sm_scale=sm, page_size=page_size)
    parity.check_out(_f32(o2), want, dtype, "fp8 prefix pool / packed rows", ulps=1, absw=absw)


def test_fp8_pools_roundtrip_through_the_pool_classes(ops):
    """MHATokenToKVPool / MLATokenToKVPool with dtype float8_e4m3fn: set -> get is the oracle's cast,
    get_mla_kv_buffer upcasts exactly."""
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, MLATokenToKVPool

    g = torch.Generator().manual_seed(5)
    layer = RadixAttention(8, 128, 128 ** -0.5, 2, layer_id=1)
    for hnd in (False, True):
        pool = MHATokenToKVPool(64, 16, FP8, 2, 128, 2, DEV, use_hnd=hnd)
        k = torch.randn(9, 2, 128, generator=g).bfloat16().to(DEV)
        v = torch.randn(9, 2, 128, generator=g).bfloat16().to(DEV)
        loc = torch.tensor([3, 17, 18, 19, 40, 41, 63, 64, 70], device=DEV)
        pool.set_kv_buffer(layer, loc, k, v, 0.5, 2.0)
        kb, vb = pool.get_kv_buffer(1)
        assert kb.dtype == FP8
        kb8 = kb.view(torch.uint8)
        rows = (kb8.permute(0, 2, 1, 3).reshape(-1, 2, 128) if hnd else kb8)[loc]
        assert np.array_equal(rows.cpu().numpy(), orc.quantize_kv_fp8(_f32(k), 0.5, True))
        assert pool.check_errors() == 0
    mla = MLATokenToKVPool(64, 1, FP8, 512, 64, 2, DEV)
    nope = torch.randn(7, 1, 512, generator=g).bfloat16().to(DEV)
    rope = torch.randn(7, 1, 64, generator=g).bfloat16().to(DEV)
    loc = torch.tensor([1, 5, 6, 30, 31, 63, 64], device=DEV)
    mla.set_mla_kv_buffer(layer, loc, nope, rope)
    rows = mla.get_key_buffer(1).view(torch.uint8)[loc, 0].cpu().numpy()
    assert np.array_equal(rows[:, :512], orc.quantize_kv_fp8(_f32(nope[:, 0]), 1.0, True))
    assert np.array_equal(rows[:, 512:], orc.quantize_kv_fp8(_f32(rope[:, 0]), 1.0, True))
    n2, r2 = mla.get_mla_kv_buffer(layer, loc)
    assert n2.dtype == torch.bfloat16 and n2.shape == (7, 1, 512) and r2.shape == (7, 1, 64)
    assert np.array_equal(_f32(n2[:, 0]), orc.fp8_e4m3fn_decode(rows[:, :512]))
    assert np.array_equal(_f32(r2[:, 0]), orc.fp8_e4m3fn_decode(rows[:, 512:]))
    # the 16-bit MLA pool's read side goes through the same kernel (pure copy)
    mla16 = MLATokenToKVPool(64, 1, torch.bfloat16, 512, 64, 2, DEV)
    mla16.set_mla_kv_buffer(layer, loc, nope, rope)
    n3, r3 = mla16.get_mla_kv_buffer(layer, loc)
    assert torch.equal(n3, nope) and torch.equal(r3, rope)


def test_fused_fp8_qkv_kv_cache_matches_the_reference_tests_bytes(ops, golden_dir):
    """ops.fused_fp8_qkv_kv_cache (the reference operator's name and signature, kernels/ops/kvcache/
    fused_fp8_qkv_kv_cache.py:35-80) on golden F21: q / k / v sliced out of one fused qkv tensor (strided rows), NHD fp8
    pools, device scalar scales; q_out, the K rows and the V rows bit-exact; untouched slots stay zero."""
    import os

    z = np.load(os.path.join(golden_dir, "fused_fp8_qkv.npz"))
    for c in range(int(z["n_cases"][0])):
        hq, hkv, hd, n, slots, has_scale, is_bf16 = (int(x) for x in z[f"c{c}.meta"])
        raw = torch.from_numpy(z[f"c{c}.qkv"])
        qkv = (raw.view(torch.bfloat16) if is_bf16 else raw).to(DEV)
        q_dim, kv_dim = hq * hd, hkv * hd
        q, k, v = qkv[:, :q_dim], qkv[:, q_dim: q_dim + kv_dim].view(n, hkv, hd), qkv[:, q_dim + kv_dim:].view(n, hkv, hd)
        loc = torch.from_numpy(z[f"c{c}.loc"]).to(DEV)
        ks = vs = None
        if has_scale:
            ks = torch.tensor(float(z[f"c{c}.scale"][0]), dtype=torch.float32, device=DEV)
            vs = torch.tensor(float(z[f"c{c}.scale"][1]), dtype=torch.float32, device=DEV)
        for with_q in (True, False):
            k_cache = torch.zeros(slots, hkv, hd, dtype=FP8, device=DEV)
            v_cache = torch.zeros(slots, hkv, hd, dtype=FP8, device=DEV)
            q_out = ops.fused_fp8_qkv_kv_cache(q if with_q else None, k, v, k_cache, v_cache, loc, ks, vs)
            if with_q:
                assert q_out.dtype == FP8 and tuple(q_out.shape) == (n, q_dim)
                assert np.array_equal(q_out.view(torch.uint8).cpu().numpy(), z[f"c{c}.q_fp8"]), c
            else:
                assert q_out is None
            kb = k_cache.view(torch.uint8).view(slots, kv_dim).cpu().numpy()
            vb = v_cache.view(torch.uint8).view(slots, kv_dim).cpu().numpy()
            assert np.array_equal(kb[z[f"c{c}.loc"]], z[f"c{c}.k_fp8"]) and np.array_equal(vb[z[f"c{c}.loc"]], z[f"c{c}.v_fp8"]), c
            rest = np.setdiff1d(np.arange(slots), z[f"c{c}.loc"])
            assert not kb[rest].any() and not vb[rest].any()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fused_fp8_qkv_kv_cache_edge_values_hnd_pool_and_errors(ops, dtype, golden_dir):
    """Edge values (saturation to +-448, ties, subnormals, -0.0) against the golden bytes AND the oracle; an HND page pool
    through kv_layout; int32 cache_loc; the reference's error for a non-16-bit k; an out-of-range slot is dropped + flagged."""
    import os

    z = np.load(os.path.join(golden_dir, "fused_fp8_qkv.npz"))
    dn = "bf16" if dtype == torch.bfloat16 else "fp16"
    for sc in (1.0, 0.5, 3.0, 0.3):
        raw = torch.from_numpy(z[f"edge_{dn}_{sc}.x"])
        x = (raw.view(torch.bfloat16) if dtype == torch.bfloat16 else raw).to(DEV)
        row = x.repeat(4)[:128].contiguous().view(1, 1, 128)          # one token, one head of 128
        want = np.tile(z[f"edge_{dn}_{sc}.fp8"], 4)[:128]
        k_cache = torch.zeros(3, 1, 128, dtype=FP8, device=DEV)
        v_cache = torch.zeros(3, 1, 128, dtype=FP8, device=DEV)
        s = torch.tensor(sc, dtype=torch.float32, device=DEV)
        ops.fused_fp8_qkv_kv_cache(None, row, row, k_cache, v_cache, torch.tensor([2], dtype=torch.int32, device=DEV), s, s)
        got = k_cache.view(torch.uint8)[2, 0].cpu().numpy()
        assert np.array_equal(got, want), (sc, got[:32], want[:32])
        assert np.array_equal(got, orc.quantize_fused_fp8(row.float().cpu().numpy().reshape(-1), sc))
        assert np.array_equal(v_cache.view(torch.uint8)[2, 0].cpu().numpy(), want)
    # HND pool [pages, Hkv, page, D] through kv_layout, random rows, vs the oracle
    g = torch.Generator().manual_seed(3)
    n, hkv, d, page, pages = 37, 4, 128, 16, 5
    k = (torch.randn(n, hkv, d, generator=g) * 2).to(dtype).to(DEV)
    v = (torch.randn(n, hkv, d, generator=g) * 2).to(dtype).to(DEV)
    q = torch.randn(n, 8 * d, generator=g).to(dtype).to(DEV)
    kh = torch.zeros(pages, hkv, page, d, dtype=FP8, device=DEV)
    vh = torch.zeros(pages, hkv, page, d, dtype=FP8, device=DEV)
    loc = (torch.randperm(pages * page - 1, generator=g)[:n] + 1).to(DEV)
    ks, vs = torch.tensor(0.37, device=DEV), torch.tensor(2.5, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    q8 = ops.fused_fp8_qkv_kv_cache(q, k, v, kh, vh, loc, ks, vs, kv_layout=ops.kv_layout_hnd(kh, vh), err_flag=err)
    got_k = kh.view(torch.uint8)[loc // page, :, loc % page].cpu().numpy()
    got_v = vh.view(torch.uint8)[loc // page, :, loc % page].cpu().numpy()
    assert np.array_equal(got_k, orc.quantize_fused_fp8(k.float().cpu().numpy(), np.float32(0.37)))
    assert np.array_equal(got_v, orc.quantize_fused_fp8(v.float().cpu().numpy(), np.float32(2.5)))
    assert np.array_equal(q8.view(torch.uint8).cpu().numpy(), orc.quantize_fused_fp8(q.float().cpu().numpy()))
    assert int(err.item()) == 0
    # the custom-op out-variants (torch.ops.radix_hip.*: the caller owns q's fp8 copy) write the same bytes
    import sglang_amd.custom_ops  # noqa: F401
    kc2, vc2 = torch.zeros(pages * page, hkv, d, dtype=FP8, device=DEV), torch.zeros(pages * page, hkv, d, dtype=FP8, device=DEV)
    q8b = torch.zeros_like(q8)
    torch.ops.radix_hip.fused_fp8_qkv_kv_cache_out(q, k, v, q8b, kc2, vc2, loc, ks, vs)
    assert torch.equal(q8b.view(torch.uint8), q8.view(torch.uint8))
    assert np.array_equal(kc2.view(torch.uint8)[loc].cpu().numpy(), got_k) and np.array_equal(vc2.view(torch.uint8)[loc].cpu().numpy(), got_v)
    kc3, vc3 = torch.zeros_like(kc2), torch.zeros_like(vc2)
    torch.ops.radix_hip.fused_fp8_kv_cache(k, v, kc3, vc3, loc, ks, vs)
    assert torch.equal(kc3.view(torch.uint8), kc2.view(torch.uint8)) and torch.equal(vc3.view(torch.uint8), vc2.view(torch.uint8))
    bad = loc.clone()
    bad[3] = pages * page + 5
    ops.fused_fp8_qkv_kv_cache(None, k, v, kh, vh, bad, ks, vs, kv_layout=ops.kv_layout_hnd(kh, vh), err_flag=err)
    assert int(err.item()) & 1
    with pytest.raises(RuntimeError, match="Unsupported dtype"):
        ops.fused_fp8_qkv_kv_cache(None, k.float(), v.float(), kh, vh, loc)
