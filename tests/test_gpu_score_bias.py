"""score_mod = relative_bias_score_mod with aux_tensors = [rel_logits] on the HIP path (round 5, ABI 14).

The reference's one Triton score_mod (kernels/ops/attention/score_mod.py:44-56: qk + Aux0[q_idx, head, q_pos - kv_pos]
inside [0, aux0_len); used by srt/models/inkling_common/attn.py:934-946) is a built-in of the kernels here
(rx_extend_params.score_bias / rx_decode_params.score_bias).  Checked against the golden produced by the reference's
Triton kernels (tests/golden/score_bias.npz, F19) and against the fp64 oracle on larger seeded cases, through the
same entry points and argument names as the reference (score_mod=..., aux_tensors=[...])."""
import os

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


def _np(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _t(a):
    if a.dtype == np.uint16:
        return torch.from_numpy(a.copy()).view(torch.bfloat16).to(DEV)
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _cases(npz):
    out = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        out.setdefault(case, {})[field] = npz[key]
    return out


def test_score_bias_golden(ops, golden_dir):
    """F19, all nine cases: two-stage extend (D = 128 on the 32x32x16 kernel, D = 64 on the generic one; window + cap),
    unified extend, decode (grouped / MHA, split KV)."""
    from sglang_amd import lib as rxlib

    cases = _cases(np.load(os.path.join(golden_dir, "score_bias.npz")))
    for name, c in cases.items():
        aux = _t(c["aux"])
        auxf = c["aux"].astype(np.float64)
        q, kb, vb = _t(c["q"]), _t(c["kb"]), _t(c["vb"])
        o = torch.zeros_like(q)
        d = q.shape[-1]
        if name.startswith("ext_"):
            ops.extend_attention_fwd(q, _t(c["k_ext"]), _t(c["v_ext"]), o, kb, vb, _t(c["qo_indptr"]), _t(c["kv_indptr"]),
                                     _t(c["kv_indices"]), None, True, None, int(np.diff(c["qo_indptr"]).max()), 1.0, 1.0,
                                     sm_scale=float(c["sm_scale"]), logit_cap=float(c["cap"]),
                                     sliding_window_size=int(c["window"]), score_mod=ops.relative_bias_score_mod,
                                     aux_tensors=[aux])
            fn, args, vi = orc.extend_attention, (c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], c["qo_indptr"],
                                                  c["kv_indptr"], c["kv_indices"]), (2, 4)
            kw = dict(is_causal=True, sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]), logit_cap=float(c["cap"]))
            expect = "extend_mfma32_kernel" if d == 128 else "extend_generic_kernel"
        elif name.startswith("uni_"):
            ops.extend_attention_fwd_unified(q, o, kb, vb, 1.0, 1.0, _t(c["qo_indptr"]), _t(c["kv_indptr"]), _t(c["kv_indices"]),
                                             _t(c["prefix_lens"]), int(np.diff(c["qo_indptr"]).max()),
                                             sm_scale=float(c["sm_scale"]), score_mod=ops.relative_bias_score_mod,
                                             aux_tensors=[aux])
            fn, args, vi = orc.extend_attention_unified, (c["q"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"],
                                                          c["kv_indices"], c["prefix_lens"]), (2,)
            kw = dict(sm_scale=float(c["sm_scale"]))
            expect = "extend_mfma32_kernel" if d == 128 else "extend_generic_kernel"
        else:
            bs, hq, _ = q.shape
            S = int(c["max_splits"])
            al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
            lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
            ops.decode_attention_fwd(q, kb, vb, o, _t(c["kv_indptr"]), _t(c["kv_indices"]), al, lse, _t(c["nsplit"]), S,
                                     float(c["sm_scale"]), 1.0, 1.0, score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])
            fn, args, vi = orc.decode_attention, (c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"], float(c["sm_scale"])), (2,)
            kw = {}
            expect = "decode_mfma_bias_kernel"
        torch.cuda.synchronize()
        assert rxlib.last_dispatch().startswith(expect), (name, rxlib.last_dispatch())
        got = _np(o).astype(np.float64)
        want = c["o"].astype(np.float64)
        ok = np.isfinite(want).all(axis=-1)
        ref, absw = parity.want_and_absw(fn, args, vi, score_bias=auxf, **kw)
        parity.check_out(got[ok], want[ok], torch.float16, (name, "vs triton golden"), ulps=2, absw=2 * absw[ok])  # (both sides round the result and their P operand to 16 bits)
        parity.check_out(got[ok], ref[ok], torch.float16, (name, "vs oracle"), ulps=1, absw=absw[ok])


def _paged(kb, vb, ps, hnd):
    """[slots, Hkv, D] token-major pools -> the device pool in the chosen layout, with the kv_layout descriptor."""
    from sglang_amd import ops

    if ps == 1:
        return kb.to(DEV), vb.to(DEV), 1, None
    n = kb.shape[0] // ps
    k4, v4 = kb[: n * ps].view(n, ps, *kb.shape[1:]), vb[: n * ps].view(n, ps, *vb.shape[1:])
    if hnd:
        kd, vd = k4.permute(0, 2, 1, 3).contiguous().to(DEV), v4.permute(0, 2, 1, 3).contiguous().to(DEV)
        return kd, vd, ps, ops.kv_layout_hnd(kd, vd)
    return k4.contiguous().to(DEV).view(n * ps, *kb.shape[1:]), v4.contiguous().to(DEV).view(n * ps, *vb.shape[1:]), ps, None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", ["gqa4_long", "mha_short_extent", "gqa8_paged", "d80_generic"])
def test_extend_score_bias_vs_oracle(ops, dtype, case):
    """Extends long enough that the D = 128 kernel mixes pipelined tiles (keys out of the bias's reach) with biased
    boundary tiles: several 64-token tiles per request either side of the reach, ragged lengths, extents that are not
    tile multiples, 16-bit and fp32 aux tensors, paged HND pools, a strided aux tensor."""
    from sglang_amd import lib as rxlib

    hq, hkv, d, prefix, ext, extent, aux_dt, ps, hnd = {
        "gqa4_long": (8, 2, 128, [300, 70, 0, 513], [200, 260, 131, 40], 100, torch.float32, 1, False),
        "mha_short_extent": (2, 2, 128, [640, 129], [64, 300], 5, None, 1, False),
        "gqa8_paged": (8, 1, 128, [400, 33], [150, 290], 129, None, 16, True),
        "d80_generic": (4, 2, 80, [70, 9], [33, 60], 16, torch.float32, 1, False),
    }[case]
    aux_dt = aux_dt or dtype
    g = torch.Generator().manual_seed(len(case) * 7 + (dtype == torch.float16))
    rng = np.random.default_rng(len(case))
    prefix, ext = np.array(prefix), np.array(ext)
    bs, T = len(prefix), int(ext.sum())
    pool = (int(prefix.sum()) + 40) // 16 * 16 + 16
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(T, hq, d, generator=g).to(dtype)
    ke = torch.randn(T, hkv, d, generator=g).to(dtype)
    ve = torch.randn(T, hkv, d, generator=g).to(dtype)
    # (a strided aux tensor: [T, Hq, extent] carved out of a wider buffer, last dim contiguous)
    aux_full = (1.5 * torch.randn(T, hq + 1, extent + 3, generator=g)).to(aux_dt)
    aux = aux_full[:, :hq, :extent]
    kvp = np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)
    kvi = (rng.permutation(pool - 1)[: int(prefix.sum())] + 1).astype(np.int64)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    sm = d ** -0.5
    want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kvp, kvi), (2, 4),
                                      is_causal=True, sm_scale=sm, score_bias=aux.double().numpy())
    kd, vd, page, lay = _paged(kb, vb, ps, hnd)
    o = torch.full((T, hq, d), float("nan"), dtype=dtype, device=DEV)
    auxd = aux_full.to(DEV)[:, :hq, :extent]
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kd, vd, _t(qo), _t(kvp), _t(kvi), None, True, None,
                             int(ext.max()), 1.0, 1.0, sm_scale=sm, page_size=page, kv_layout=lay,
                             score_mod=ops.relative_bias_score_mod, aux_tensors=[auxd])
    torch.cuda.synchronize()
    name = rxlib.last_dispatch()
    if d == 128:  # the feature instance of the 32x32x16 kernel: <T, IdxT, LINEAR, VSCALE, NW, KV8, PLAIN = false, 0>
        assert name.startswith("extend_mfma32_kernel") and name.endswith("false, false, 0>"), name
    else:
        assert name.startswith("extend_generic_kernel"), name
    parity.check_out(_np(o.float()), want, dtype, ("score bias extend", case), ulps=1, absw=absw)
    # the bias must matter in this case, and switching it off must give the plain result again
    plain = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kvp, kvi, is_causal=True, sm_scale=sm)
    assert np.abs(plain - want).max() > 5e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", ["gqa4_splits", "mha64", "gqa16_r2t", "d96_generic"])
def test_decode_score_bias_vs_oracle(ops, dtype, case):
    """Decode: requests longer and shorter than the bias extent, split KV (the bias lives in the LAST splits only) and one
    pass, kv_indices and the req_to_token lookup, 16 q heads per kv head, a head dim outside the biased MFMA instances."""
    from sglang_amd import lib as rxlib

    hq, hkv, d, lens, extent, aux_dt, splits, r2t = {
        "gqa4_splits": (8, 2, 128, [700, 33, 1, 2049, 128], 100, torch.float32, 8, False),
        "mha64": (4, 4, 64, [300, 17], 40, None, 1, False),
        "gqa16_r2t": (16, 1, 128, [1000, 64, 257], 257, None, 4, True),
        "d96_generic": (6, 2, 96, [200, 40], 64, torch.float32, 2, False),
    }[case]
    aux_dt = aux_dt or dtype
    g = torch.Generator().manual_seed(len(case) * 11 + (dtype == torch.float16))
    rng = np.random.default_rng(len(case) + 3)
    lens = np.array(lens)
    bs = len(lens)
    pool = int(lens.sum()) + 9
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    aux = (1.5 * torch.randn(bs, hq, extent, generator=g)).to(aux_dt)
    kvp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    kvi = (rng.permutation(pool - 1)[: int(lens.sum())] + 1).astype(np.int64)
    sm = d ** -0.5
    want, absw = parity.want_and_absw(orc.decode_attention, (_np(q), _np(kb), _np(vb), kvp, kvi, sm), (2,),
                                      score_bias=aux.double().numpy())
    o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=DEV)
    al = torch.zeros(bs, hq, splits, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, splits, dtype=torch.float32, device=DEV)
    ns = torch.tensor([min(splits, max(1, int(n) // 64)) for n in lens], dtype=torch.int32, device=DEV)
    if r2t:
        r2t_tab = torch.zeros(bs + 1, int(lens.max()), dtype=torch.int32)
        for b in range(bs):
            r2t_tab[b + 1, : lens[b]] = torch.from_numpy(kvi[kvp[b]: kvp[b + 1]].astype(np.int32))
        ops.decode_attention_fwd_paged(q.to(DEV), kb.to(DEV), vb.to(DEV), o, r2t_tab.to(DEV),
                                       torch.arange(1, bs + 1, dtype=torch.int64, device=DEV),
                                       torch.from_numpy(lens).to(DEV), al, lse, ns, splits, sm, 1.0, 1.0,
                                       score_mod=ops.relative_bias_score_mod, aux_tensors=[aux.to(DEV)])
    else:
        ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o, _t(kvp), _t(kvi), al, lse, ns, splits, sm, 1.0, 1.0,
                                 score_mod=ops.relative_bias_score_mod, aux_tensors=[aux.to(DEV)])
    torch.cuda.synchronize()
    name = rxlib.last_dispatch()
    assert name.startswith("decode_mfma_bias_kernel" if d in (64, 128) else "decode_generic_kernel"), name
    # (split KV: partials merged in fp32, one output rounding)
    parity.check_out(_np(o.float()), want, dtype, ("score bias decode", case), ulps=1, absw=absw)
    plain = orc.decode_attention(_np(q), _np(kb), _np(vb), kvp, kvi, sm)
    assert np.abs(plain - want).max() > 5e-2


def test_score_mod_contract(ops):
    """The argument contract of the reference (score_mod.py:30-41) and what cannot cross a C ABI."""
    d, hq, hkv = 128, 4, 2
    q = torch.randn(3, hq, d, device=DEV, dtype=torch.bfloat16)
    kb = torch.randn(40, hkv, d, device=DEV, dtype=torch.bfloat16)
    o = torch.zeros_like(q)
    kvp = torch.tensor([0, 5, 9, 20], dtype=torch.int32, device=DEV)
    kvi = torch.arange(1, 21, dtype=torch.int64, device=DEV)
    al = torch.zeros(3, hq, 1, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(3, hq, 1, dtype=torch.float32, device=DEV)
    call = lambda **kw: ops.decode_attention_fwd(q, kb, kb, o, kvp, kvi, al, lse, None, 1, d ** -0.5, 1.0, 1.0, **kw)  # noqa: E731
    with pytest.raises(NotImplementedError):
        call(score_mod=lambda *a: a[0], aux_tensors=[torch.zeros(3, hq, 4, device=DEV)])
    with pytest.raises(AssertionError):
        call(score_mod=ops.relative_bias_score_mod, aux_tensors=[])
    with pytest.raises(AssertionError):
        call(score_mod=ops.relative_bias_score_mod, aux_tensors=[torch.zeros(3, hq, device=DEV)])
    with pytest.raises(TypeError):
        call(score_mod=ops.relative_bias_score_mod, aux_tensors=[torch.zeros(3, hq, 4, device=DEV, dtype=torch.float16)])
    call(aux_tensors=[torch.zeros(1, device=DEV)])  # (aux without a score_mod: ignored, as unpack_aux_tensors does)

    # an object that merely carries the reference function's name (its triton JITFunction does) selects the built-in
    class _Jit:
        __name__ = "relative_bias_score_mod"

    aux = torch.randn(3, hq, 8, device=DEV)
    o1 = torch.zeros_like(q)
    ops.decode_attention_fwd(q, kb, kb, o1, kvp, kvi, al, lse, None, 1, d ** -0.5, 1.0, 1.0, score_mod=_Jit(), aux_tensors=[aux])
    call(score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])
    assert torch.equal(o, o1)
    # GQA-packed rows and the bias do not combine (the packed instances are PLAIN)
    from sglang_amd.lib import RadixHipError
    qo = torch.tensor([0, 1, 2, 3], dtype=torch.int64, device=DEV)
    with pytest.raises(RadixHipError):
        ops.extend_attention_fwd(q, kb[:3], kb[:3], o, kb, kb, qo, kvp, kvi, None, True, None, 1, 1.0, 1.0, q_pack=2,
                                 score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])


@pytest.mark.parametrize("index_mode", ["paged", "indices"])
def test_backend_forward_with_score_mod(ops, index_mode):
    """RadixAttention.forward(..., score_mod=, aux_tensors=) through HipRadixAttnBackend (the reference passes both as
    keyword arguments of the layer call: radix_attention.py:219-232, triton_backend.py:1259-1260,1723-1724): an extend
    step over cached prefixes, then a decode step, page 16, GQA 4; the oracle reads the same pool rows through
    req_to_token."""
    from test_gpu_backend import _Harness, _bits

    from sglang_amd.forward_batch import ForwardBatch

    hq, hkv, d, ps, extent = 8, 2, 128, 16, 48
    hs = _Harness(ps, hq, hkv, d, torch.bfloat16, "shuffled_pages", index_mode)
    prefix_lens, extend_lens = [100, 0, 37], [70, 129, 5]
    bs = len(prefix_lens)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, prefix_lens)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
    loc = hs.alloc_extend(rows, list(prefix_lens), seq_lens)
    T = sum(extend_lens)
    q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)
    aux = (1.5 * torch.randn(T, hq, extent, generator=hs.gen)).to(DEV)
    fb = ForwardBatch.for_extend(rpi, torch.tensor(seq_lens, device=DEV), loc, list(prefix_lens), list(extend_lens))
    hs.backend.init_forward_metadata(fb)
    o = hs.layer(q, k, v, fb, hs.backend, score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = hs.r2t.req_to_token.cpu().numpy()

    def lists(lens):
        idx = np.concatenate([r2t[r, :n] for r, n in zip(rows, lens)]).astype(np.int64)
        return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), idx

    kvp, kvi = lists(seq_lens)
    qo = np.concatenate([[0], np.cumsum(extend_lens)]).astype(np.int64)
    want, absw = parity.want_and_absw(orc.extend_attention_unified, (_bits(q.view(T, hq, d)), _bits(kb), _bits(vb), qo, kvp, kvi,
                                                                   np.array(prefix_lens)), (2,), sm_scale=d ** -0.5,
                                      score_bias=aux.double().cpu().numpy())
    parity.check_out(o.view(T, hq, d).float().cpu().numpy().astype(np.float64), want, torch.bfloat16, ("backend extend", index_mode), absw=absw)
    plain = hs.layer(q, k, v, fb, hs.backend, save_kv_cache=False)
    assert (plain.float() - o.float()).abs().max().item() > 5e-2

    # ---- one decode step on top
    seq2 = [s + 1 for s in seq_lens]
    seq_t = torch.tensor(seq2, dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, s - 1]) for r, s in zip(rows, seq_lens)], dtype=torch.int64, device=DEV)
    loc2 = hs.alloc.alloc_decode(seq_t.to(DEV), seq_t, last)
    hs.r2t.req_to_token[rpi, torch.tensor(seq_lens, device=DEV)] = loc2.to(torch.int32)
    q2, k2, v2 = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    aux2 = (1.5 * torch.randn(bs, hq, extent, generator=hs.gen)).to(DEV)
    fb2 = ForwardBatch.for_decode(rpi, seq_t.to(DEV), loc2, seq_t)
    hs.backend.init_forward_metadata(fb2)
    o2 = hs.layer(q2, k2, v2, fb2, hs.backend, score_mod=ops.relative_bias_score_mod, aux_tensors=[aux2])
    r2t = hs.r2t.req_to_token.cpu().numpy()
    kvp2, kvi2 = lists(seq2)
    kb, vb = hs.pool.get_kv_buffer(0)
    want2, absw2 = parity.want_and_absw(orc.decode_attention, (_bits(q2.view(bs, hq, d)), _bits(kb), _bits(vb), kvp2, kvi2, d ** -0.5),
                                        (2,), score_bias=aux2.double().cpu().numpy())
    parity.check_out(o2.view(bs, hq, d).float().cpu().numpy().astype(np.float64), want2, torch.bfloat16, ("backend decode", index_mode), absw=absw2)
    assert hs.pool.check_errors() == 0


@pytest.mark.parametrize("index_mode", ["paged", "indices"])
def test_backend_decode_score_mod_on_a_sliding_window_layer(ops, index_mode):
    """a sliding-window layer that passes score_mod (Inkling's local layers,
    sliding_window_size = local_extent - 1) must attend the WINDOW list with the window's split schedule
    (triton_backend.py:1770-1781, also when score_mod is set), with rel = (len - 1) - n taken over that list."""
    from test_gpu_backend import _Harness, _bits

    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch

    hq, hkv, d, ps, W, extent = 8, 2, 128, 16, 48, 64
    hs = _Harness(ps, hq, hkv, d, torch.bfloat16, "shuffled_pages", index_mode)
    hs.backend.sliding_window_size = W
    hs.backend.window_kv_indptr = torch.zeros_like(hs.backend.kv_indptr)
    swa_layer = RadixAttention(hq, d, d ** -0.5, hkv, 0, sliding_window_size=W)
    seq_lens = [10, 170, 48, 333]
    bs = len(seq_lens)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, seq_lens)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq_t = torch.tensor([s + 1 for s in seq_lens], dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, s - 1]) for r, s in zip(rows, seq_lens)], dtype=torch.int64, device=DEV)
    dloc = hs.alloc.alloc_decode(seq_t.to(DEV), seq_t, last)
    hs.r2t.req_to_token[rpi, torch.tensor(seq_lens, device=DEV)] = dloc.to(torch.int32)
    q1, k1, v1 = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    aux = (1.5 * torch.randn(bs, hq, extent, generator=hs.gen)).to(DEV)
    fbd = ForwardBatch.for_decode(rpi, seq_t.to(DEV), dloc, seq_t)
    hs.backend.init_forward_metadata(fbd)
    assert hs.backend.forward_metadata.window_kv_indptr is not None
    o_swa = swa_layer(q1, k1, v1, fbd, hs.backend, score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])
    o_full = hs.layer(q1, k1, v1, fbd, hs.backend, save_kv_cache=False, score_mod=ops.relative_bias_score_mod, aux_tensors=[aux])
    kb, vb = hs.pool.get_kv_buffer(0)
    sl = seq_t.numpy()
    wl = np.minimum(sl, W)
    r2t = _bits(hs.r2t.req_to_token)
    kvp, kvi = orc.build_kv_indices(r2t, np.array(rows), wl, kv_start=sl - wl)
    want, absw = parity.want_and_absw(orc.decode_attention, (_bits(q1.view(bs, hq, d)), _bits(kb), _bits(vb), kvp, kvi, d ** -0.5),
                                      (2,), score_bias=aux.double().cpu().numpy())
    parity.check_out(o_swa.view(bs, hq, d).float().cpu().numpy().astype(np.float64), want, torch.bfloat16,
                     ("backend swa decode + score_mod", index_mode), absw=absw)
    kvp_f, kvi_f = orc.build_kv_indices(r2t, np.array(rows), sl)
    want_f, absw_f = parity.want_and_absw(orc.decode_attention, (_bits(q1.view(bs, hq, d)), _bits(kb), _bits(vb), kvp_f, kvi_f, d ** -0.5),
                                          (2,), score_bias=aux.double().cpu().numpy())
    parity.check_out(o_full.view(bs, hq, d).float().cpu().numpy().astype(np.float64), want_f, torch.bfloat16,
                     ("backend full decode + score_mod in the swa model", index_mode), absw=absw_f)
    # the long requests' window rows differ from their full-context rows
    assert (o_swa.float() - o_full.float()).abs().max().item() > 1e-2
    assert hs.pool.check_errors() == 0
