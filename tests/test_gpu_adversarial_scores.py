"""Adversarial score sequences for every kernel that carries the sum-check softmax (VERDICT r04 "what's weak" 1).

Round 4 replaced the per-block row maximum of the online softmax by a check of the lane's partial row sum: on the
common path P = exp2(s c - m) is taken against the STANDING reference max m, and only when a lane's sum over its 16
values exceeds 4096 is the block redone the classic way (true max, thresholded update: m moves only when exceeded by
more than 2^8).  With N(0, 1) test data the redo fires on a row's first tile only; the code that fires in the MIDDLE
of a row -- the redo, the accumulator rescale, the l update -- needs scores built for it.  Patterns (in nats, as the
score of key n for every query; gaussian noise of sigma ~0.1 on top):

  ramp      +0.02 per key: the reference max is overtaken again and again, ~1.8 log2 units per 64-key tile
  spikes    flat, a +40 spike every ~300 keys, each higher than the last: one huge jump, then exp2 underflow after it
  plateau   one early peak (9.5), then a plateau 5.5 below it: everything sits inside the 2^8 slack for a whole row
  straddle  a quiet first tile, then 32-key blocks at +5.30 / +5.62 / alternating: a lane's 16-value sum lands at
            ~3200 / ~4300 -- on both sides of the 4096 limit, with the true max INSIDE the slack (a redo that must not
            move m) -- and a final block at +9 (a redo that must)
  falling   -0.03 per key from a high start: P underflows to zero long before the row ends

Every case names the kernel instance it must run (rx_last_dispatch) and is held to parity_util.check_out against the
fp64 oracle (extend_attention.py:372-631 online softmax; decode_attention.py:383-608), LSE too where the call returns it.
A second test drives rows whose prefix is masked out completely (-inf against -inf) through the kernels that take a
custom mask."""
import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
PATTERNS = ["ramp", "spikes", "plateau", "straddle", "falling"]
TN = {torch.bfloat16: "rx::BF16", torch.float16: "rx::F16"}


def _np(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _base(pattern, n):
    if pattern == "ramp":
        return torch.arange(n, dtype=torch.float32) * 0.02
    if pattern == "spikes":
        b = torch.zeros(n)
        for j, pos in enumerate(range(50, n, 300)):
            b[pos] = 40.0 + 7.0 * j
        return b
    if pattern == "plateau":
        b = torch.full((n,), 4.0)
        b[3] = 9.5
        return b
    if pattern == "straddle":
        b = torch.zeros(n)
        levels = [5.30, 5.62]
        for blk, lo in enumerate(range(64, n - 96, 32)):
            if blk % 3 == 2:
                b[lo: lo + 32: 2], b[lo + 1: lo + 32: 2] = levels[0], levels[1]
            else:
                b[lo: lo + 32] = levels[blk % 3]
        b[n - 96: n - 64] = 9.0
        return b
    if pattern == "falling":
        return 30.0 - torch.arange(n, dtype=torch.float32) * 0.03
    raise ValueError(pattern)


# (name, dk, dv, hq, hkv, options, expected instance prefix with {T})
EXTEND_CASES = [
    ("mfma32_pk4_8w", 128, 128, 8, 2, {"ext32_small_wg": 0, "ext32_pack_min_wgs": 0}, "extend_mfma32_kernel<{T}, long, true, false, 8, false, true, 4>"),
    ("mfma32_pk8_8w", 128, 128, 8, 1, {"ext32_small_wg": 0, "ext32_pack_min_wgs": 0}, "extend_mfma32_kernel<{T}, long, true, false, 8, false, true, 8>"),
    ("mfma32_plain_4w", 128, 128, 2, 1, {"ext32_small_wg": 1}, "extend_mfma32_kernel<{T}, long, true, false, 4, false, true, 0>"),
    ("d256_g1", 256, 256, 2, 2, {}, "extend_d256_kernel<{T}, 256, 256, false>"),
    ("d256_g4", 256, 256, 8, 2, {}, "extend_d256_kernel<{T}, 256, 256, false>"),
    ("d192_g1", 192, 128, 2, 2, {}, "extend_d256_kernel<{T}, 192, 128, false>"),
    ("d192_g4", 192, 128, 8, 2, {}, "extend_d256_kernel<{T}, 192, 128, false>"),
    ("d96_g1", 96, 96, 2, 2, {}, "extend_d256_kernel<{T}, 96, 96, false>"),
    ("d96_g4", 96, 96, 8, 2, {}, "extend_d256_kernel<{T}, 96, 96, false>"),
    ("d64_g1", 64, 64, 2, 2, {}, "extend_d256_kernel<{T}, 64, 64, false>"),
    ("d64_g4", 64, 64, 8, 2, {}, "extend_d256_kernel<{T}, 64, 64, false>"),
    ("mfma16_d64", 64, 64, 4, 2, {"extend_d256_at64": 0}, "extend_mfma_kernel<{T}, 64, long, true, false, true"),
    ("mla_latent", 576, 512, 16, 1, {}, "extend_mla_kernel<{T}, false>"),
]


def _extend_inputs(pattern, dtype, dk, dv, hq, hkv, P, E, seed, mla=False):
    g = torch.Generator().manual_seed(seed)
    n = P + E
    sm = (192.0 if mla else float(dk)) ** -0.5
    q = torch.randn(E, hq, dk, generator=g) * 0.05
    q[:, :, 0] = 1.0 / sm            # score(key n) = k[n][0] (+ noise from the other coordinates)
    kfull = torch.randn(n, hkv, dk, generator=g) * 0.3
    kfull[:, :, 0] = _base(pattern, n)[:, None]
    vfull = kfull[..., :dv].clone() if mla else torch.randn(n, hkv, dv, generator=g)
    return q.to(dtype), kfull.to(dtype), vfull.to(dtype), sm


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("case", EXTEND_CASES, ids=[c[0] for c in EXTEND_CASES])
def test_extend_kernels_under_adversarial_scores(case, pattern, dtype):
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    name, dk, dv, hq, hkv, opts, expect = case
    mla = name == "mla_latent"
    P, E = 1024 + 37, 192
    rng = np.random.default_rng(len(name) + len(pattern))
    q, kfull, vfull, sm = _extend_inputs(pattern, dtype, dk, dv, hq, hkv, P, E, seed=3 + dk, mla=mla)
    n = P + E
    pool = n + 3
    slots = rng.permutation(pool - 1)[:n] + 1
    kb = torch.zeros(pool, hkv, dk, dtype=dtype)
    vb = torch.zeros(pool, hkv, dv, dtype=dtype)
    kb[slots] = kfull
    vb[slots] = vfull
    kv_indptr = np.array([0, P], dtype=np.int32)
    kv_indices = slots[:P].astype(np.int64)
    qo = np.array([0, E], dtype=np.int64)
    ke, ve = kfull[P:].contiguous(), vfull[P:].contiguous()
    want, want_lse, absw = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices, sm_scale=sm,
                                                return_lse=True, return_absw=True)  # (the |V| twin from the same pass)
    o = torch.full((E, hq, dv), float("nan"), dtype=dtype, device=DEV)
    lse = torch.zeros(E, hq, dtype=torch.float32, device=DEV)
    ctx = [rxlib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        kbd = kb.to(DEV)
        ked = ke.to(DEV)
        if mla:   # v as the reference's model code passes it: views of the k tensors' first 512 columns
            vbd, ved = kbd[..., :dv], ked[..., :dv]
        else:
            vbd, ved = vb.to(DEV), ve.to(DEV)
        ops.extend_attention_fwd(q.to(DEV), ked, ved, o, kbd, vbd, _t(qo), _t(kv_indptr), _t(kv_indices), None, True, None, E, 1.0,
                                 1.0, sm_scale=sm, lse_extend=lse, avg_kv_len_hint=P + 2048)
        torch.cuda.synchronize()
        got_name = rxlib.last_dispatch()
    finally:
        for c in reversed(ctx):
            c.__exit__(None, None, None)
    assert got_name.startswith(expect.format(T=TN[dtype])), got_name
    got = o.float().cpu().numpy()
    assert not np.isnan(got).any()
    parity.check_out(got, want, dtype, ("adversarial", name, pattern), ulps=1, absw=absw)
    np.testing.assert_allclose(lse.cpu().numpy(), want_lse, atol=5e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("rows", ["16bit", "fp8", "fp8_k_split"])
def test_mla_decode_kernels_under_adversarial_scores(rows, pattern, dtype):
    """rx::decode_mla_kernel (16-bit latent rows) and rx::decode_mla8_dma_kernel (fp8 rows), single pass and split-KV: the
    same patterns along the context of two requests (one ending inside a tile)."""
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    hq, ps = 16, 16
    lens = np.array([1500, 833], dtype=np.int64)
    bs = len(lens)
    sm = 192.0 ** -0.5
    g = torch.Generator().manual_seed(11)
    rng = np.random.default_rng(len(pattern))
    npages = int(sum(-(-int(x) // ps) for x in lens)) + 2
    pool = npages * ps
    pages = rng.permutation(np.arange(1, npages))
    r2t = np.zeros((bs + 1, 1600), dtype=np.int32)
    kv = torch.zeros(pool, 1, 576)
    pi = 0
    for i, n_ in enumerate(lens):
        k = -(-int(n_) // ps)
        sl = np.concatenate([np.arange(p * ps, (p + 1) * ps) for p in pages[pi: pi + k]])[:n_]
        pi += k
        r2t[i + 1, :n_] = sl
        rows_i = torch.randn(int(n_), 576, generator=g) * 0.3
        rows_i[:, 0] = _base(pattern, int(n_))
        kv[torch.from_numpy(sl), 0] = rows_i
    q = torch.randn(bs, hq, 576, generator=g) * 0.05
    q[:, :, 0] = 1.0 / sm
    q = q.to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    k_split = rows == "fp8_k_split"   # the 32-token form that splits QK^T over the waves (option decode_mla8_t64 = 0)
    rows = "fp8" if k_split else rows
    if rows == "fp8":
        kvq = kv.to(torch.float8_e4m3fn)
        kvn = kvq.float().numpy().astype(np.float64)           # the dequantised rows: what the kernel computes with
        kvd = kvq.to(DEV)
        expect = "decode_mla8_dma_kernel" if k_split else "decode_mla8_t64_kernel"
    else:
        kvq = kv.to(dtype)
        kvn = _np(kvq)
        kvd = kvq.to(DEV)
        expect = "decode_mla_kernel"
    want, absw = orc.decode_attention(_np(q), kvn, kvn[..., :512], kv_indptr, kv_indices, sm, return_absw=True)
    qd = q.to(DEV)
    o = torch.full((bs, hq, 512), float("nan"), dtype=dtype, device=DEV)
    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, _t(lens).int(), hq, 1, S, 256)
    al = torch.zeros(bs, hq, S, 512, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    if rows == "16bit":   # single pass (the 16-bit kernel's one-split form)
        ops.decode_attention_fwd_paged(qd, kvd, kvd[..., :512], o, _t(r2t), _t(rpi), _t(lens), None, None, None, 1, sm, page_size=ps)
        torch.cuda.synchronize()
        assert rxlib.last_dispatch().startswith(expect), rxlib.last_dispatch()
        parity.check_out(o.float().cpu().numpy(), want, dtype, ("adversarial", rows, pattern, "single"), absw=absw)
    o2 = torch.full_like(o, float("nan"))
    with rxlib.option("decode_mla8_t64", 0 if k_split else 1):
        ops.decode_attention_fwd(qd, kvd, kvd[..., :512], o2, _t(kv_indptr), _t(kv_indices), al, lse, nsplit, S, sm, 1.0, 1.0, page_size=ps)
        torch.cuda.synchronize()
    assert rxlib.last_dispatch().startswith(expect), rxlib.last_dispatch()
    parity.check_out(o2.float().cpu().numpy(), want, dtype, ("adversarial", rows, pattern, "split"), absw=absw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("dk,hq,hkv", [(128, 4, 2), (64, 4, 2), (256, 4, 2), (80, 2, 2)], ids=["d128", "d64", "d256", "d80"])
def test_rows_with_a_fully_masked_prefix(dk, hq, hkv, dtype):
    """-inf against -inf: under a tree mask some query rows see NOTHING of the cached prefix (every prefix tile is a tile of
    -inf scores against a reference max that is still -inf) and then only a few of the new tokens; others see everything.
    extend_attention.py:474-475 pins the all-masked maximum to -1e20 for exactly this."""
    from sglang_amd import ops

    P, E = 300, 40
    rng = np.random.default_rng(dk)
    g = torch.Generator().manual_seed(dk)
    pool = P + 4
    kb = torch.randn(pool, hkv, dk, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, dk, generator=g).to(dtype)
    q = torch.randn(E, hq, dk, generator=g).to(dtype)
    ke = torch.randn(E, hkv, dk, generator=g).to(dtype)
    ve = torch.randn(E, hkv, dk, generator=g).to(dtype)
    slots = (rng.permutation(pool - 1)[:P] + 1).astype(np.int64)
    mask = np.ones((E, P + E), dtype=np.uint8)
    mask[:, P:] = np.tril(np.ones((E, E), dtype=np.uint8))
    blind = np.arange(E) % 3 == 1                # every third row: no prefix at all, of the new tokens only itself and one earlier
    mask[blind, :P] = 0
    for i in np.nonzero(blind)[0]:
        mask[i, P:] = 0
        mask[i, P + i] = 1
        mask[i, P + max(0, i - 5)] = 1
    mask[7, : P - 3] = 0                           # one row that sees only the last three prefix keys
    kv_indptr = np.array([0, P], dtype=np.int32)
    qo = np.array([0, E], dtype=np.int64)
    mi = np.array([0, mask.size], dtype=np.int64)
    sm = dk ** -0.5
    want, absw = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, slots, custom_mask=mask.reshape(-1),
                                      mask_indptr=mi, sm_scale=sm, skip_prefix_custom_mask=False, return_absw=True)
    o = torch.full((E, hq, dk), float("nan"), dtype=dtype, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo), _t(kv_indptr), _t(slots),
                             _t(mask.reshape(-1)), True, _t(mi), E, 1.0, 1.0, sm_scale=sm, skip_prefix_custom_mask=False)
    torch.cuda.synchronize()
    got = o.float().cpu().numpy()
    assert not np.isnan(got).any()
    parity.check_out(got, want, dtype, ("masked prefix rows", dk), ulps=1, absw=absw)
