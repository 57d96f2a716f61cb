"""Fused QK-norm + RoPE (+ KV store): ops.fused_qk_norm_rope / rx_qknorm_rope_store_kv against the oracle
(oracle.fused_qk_norm_rope, pinned by tests/golden/qknorm_rope.npz to the reference's RMSNorm.forward_native +
apply_rotary_emb) and against the golden itself.  Reference: fused_qk_norm_rope, kernels/ops/attention/fused_qknorm_rope.py:37-186
(kernel kernels/jit/csrc/elementwise/fused_qknorm_rope.cuh); its own test kernels/aot/tests/test_fused_qk_norm_rope.py
holds it to rtol 5e-2 / atol 1e-1 of the bf16 module pair -- here the bar is the output's own rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qknorm_rope.npz")


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _ulp(x, dtype):
    """spacing of the 16-bit format at |x| (float64 array)"""
    mant = 8 if dtype == torch.bfloat16 else 11
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -14)))
    return 2.0 ** (e - (mant - 1))


def _check(got, want, dtype, ulps, tag, extra=0.0):
    err = np.abs(got.astype(np.float64) - want)
    bound = ulps * _ulp(want, dtype) + extra * np.abs(want).max()
    bad = err > bound
    assert not bad.any(), (tag, float(err.max()), float((err / bound).max()))


def _cases():
    z = np.load(GOLDEN)
    cases = {}
    for key in z.files:
        c, f = key.split(".", 1)
        cases.setdefault(c, {})[f] = z[key]
    return cases


@pytest.mark.parametrize("name", ["neox128", "gptj128", "partial64of128", "gptj64", "neox256", "partial_gptj32of64"])
def test_fused_qk_norm_rope_matches_the_reference_pair_golden(name):
    """The golden's inputs through the op, both frequency forms: on the fly from `base` (the reference kernel's way) and
    from the golden's cos / sin rows as a cache; outputs within one rounding of the reference pair's fp32 result (plus, on
    the fly, the fp32 angle's own error at positions up to 4095), v untouched."""
    from sglang_amd import ops

    c = _cases()[name]
    dtype = torch.float16 if c["qkv"].dtype == np.float16 else torch.bfloat16
    hq, hkv, d, rot = int(c["hq"]), int(c["hkv"]), int(c["head_dim"]), int(c["rotary_dim"])
    qkv0 = torch.from_numpy(c["qkv"].copy())
    qkv0 = qkv0.view(torch.bfloat16) if dtype == torch.bfloat16 else qkv0
    w = lambda a: (torch.from_numpy(a.copy()).view(torch.bfloat16) if dtype == torch.bfloat16 else torch.from_numpy(a.copy())).to(DEV)  # noqa: E731
    n = qkv0.shape[0]
    want_q, want_k = c["q_out"].astype(np.float64), c["k_out"].astype(np.float64)
    for form in ("on_the_fly", "cache", "int64_positions"):
        qkv = qkv0.clone().to(DEV)
        kw = {}
        pos = torch.from_numpy(c["positions"].astype(np.int32)).to(DEV)
        if form == "cache":
            kw["cos_sin_cache"] = torch.from_numpy(c["cos_sin"].astype(np.float32)).to(DEV)
            pos = torch.arange(n, dtype=torch.int32, device=DEV)
        if form == "int64_positions":
            pos = pos.to(torch.int64)
        ops.fused_qk_norm_rope(qkv, hq, hkv, hkv, d, float(c["eps"]), w(c["q_weight"]), w(c["k_weight"]), float(c["base"]),
                               bool(c["is_neox"]), pos, 1.0, 0.0, 0.0, 1.0, rot, **kw)
        torch.cuda.synchronize()
        got = qkv.float().cpu().numpy()
        # angle error of the fp32 on-the-fly path: pos * freq with both factors rounded -> up to ~4e-4 rad at pos 4095
        extra = 0.0 if form == "cache" else 6e-4
        _check(got[:, : hq * d].reshape(n, hq, d), want_q, dtype, 1.01, (name, form, "q"), extra)
        _check(got[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d), want_k, dtype, 1.01, (name, form, "k"), extra)
        assert torch.equal(qkv[:, (hq + hkv) * d:].cpu(), qkv0[:, (hq + hkv) * d:]), "v must not be touched"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("neox", [True, False], ids=["neox", "gptj"])
@pytest.mark.parametrize("geom", [(32, 8, 128, 128), (12, 1, 128, 64), (16, 8, 64, 64), (5, 5, 96, 32), (4, 2, 256, 256), (3, 1, 512, 128)],
                         ids=["qwen3_8b", "glm_tp8_partial", "d64", "d96_partial", "d256", "d512"])
def test_fused_qk_norm_rope_vs_oracle_with_yarn_and_attention_factor(dtype, neox, geom):
    """Model geometries of the reference's test (Qwen3 head counts, GLM TP8 12 / 1 / 1) and beyond (head dims 96 / 512, which
    the reference kernel refuses), YaRN blending (factor 4, ramp 8..24) and an attention factor, against the fp64 oracle at
    the output's own rounding; 257 tokens so that the last workgroup is partial."""
    from sglang_amd import ops

    hq, hkv, d, rot = geom
    n = 257
    g = torch.Generator().manual_seed(hq * 1000 + d + int(neox))
    qkv0 = torch.randn(n, (hq + 2 * hkv) * d, generator=g).to(dtype)
    qw, kw_ = (torch.randn(d, generator=g) * 5.0).to(dtype), (torch.randn(d, generator=g) * 5.0).to(dtype)
    pos = torch.randint(0, 3000, (n,), generator=g).to(torch.int32)
    for factor, low, high, att in ((1.0, 0.0, 0.0, 1.0), (4.0, 8.0, 24.0, 1.0 + 0.1 * np.log(4.0))):
        if factor != 1.0 and rot < 64:
            low, high = 2.0, 10.0
        qkv = qkv0.clone().to(DEV)
        ops.fused_qk_norm_rope(qkv, hq, hkv, hkv, d, 1e-6, qw.to(DEV), kw_.to(DEV), 10000.0, neox, pos.to(DEV), factor, low, high,
                               att, rot)
        torch.cuda.synchronize()
        q = qkv0[:, : hq * d].reshape(n, hq, d)
        k = qkv0[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d)
        want_q, want_k = orc.fused_qk_norm_rope(_bits(q), _bits(k), _bits(qw), _bits(kw_), pos.numpy(), 1e-6, 10000.0, neox,
                                                factor, low, high, att, rot)
        got = qkv.float().cpu().numpy()
        _check(got[:, : hq * d].reshape(n, hq, d), want_q, dtype, 1.01, ("q", geom, factor), 5e-4)
        _check(got[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d), want_k, dtype, 1.01, ("k", geom, factor), 5e-4)
        assert torch.equal(qkv[:, (hq + hkv) * d:].cpu(), qkv0[:, (hq + hkv) * d:])


@pytest.mark.parametrize("pool", ["bf16_nhd", "bf16_hnd_paged", "fp8"])
def test_fused_qk_norm_rope_with_the_kv_store_in_the_same_launch(pool):
    """This library's extension (as rope_store_kv): the finished k rows and the v rows land in the paged pool in the same
    launch -- 16-bit pools hold exactly the bits the op left in qkv's k part (NHD and HND page layouts), fp8 pools the
    quant-on-write of store_kv_fp8 of those rows; skipped slots and out-of-range slots behave as in the store kernels."""
    from sglang_amd import lib as L
    from sglang_amd import ops

    hq, hkv, d, n, page = 8, 2, 128, 37, 16
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(7)
    qkv0 = torch.randn(n, (hq + 2 * hkv) * d, generator=g).to(dtype)
    qw, kw_ = (torch.randn(d, generator=g) * 2.0).to(dtype).to(DEV), (torch.randn(d, generator=g) * 2.0).to(dtype).to(DEV)
    pos = torch.arange(n, dtype=torch.int32) + 50
    slots = 8 * page
    loc = (torch.randperm(slots - 1, generator=g)[:n] + 1).to(torch.int64)
    loc[5] = 0                                  # reserved slot: skipped
    fp8 = pool == "fp8"
    pdt = torch.float8_e4m3fn if fp8 else dtype
    if pool == "bf16_hnd_paged":
        kb = torch.zeros(slots // page, hkv, page, d, dtype=pdt, device=DEV)
        vb = torch.zeros_like(kb)
        lay = ops.kv_layout_hnd(kb, vb)
    else:
        kb = torch.zeros(slots, hkv, d, dtype=pdt, device=DEV)
        vb = torch.zeros_like(kb)
        lay = ops._kv_layout(kb, vb, 1)
    k_scale, v_scale = (0.5, 2.0) if fp8 else (1.0, 1.0)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    qkv = qkv0.clone().to(DEV)
    ops.fused_qk_norm_rope(qkv, hq, hkv, hkv, d, 1e-6, qw, kw_, 10000.0, True, pos.to(DEV), layout=lay, loc=loc.to(DEV),
                           size_limit=slots, k_scale=k_scale, v_scale=v_scale, err_flag=err)
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    plain = qkv0.clone().to(DEV)
    ops.fused_qk_norm_rope(plain, hq, hkv, hkv, d, 1e-6, qw, kw_, 10000.0, True, pos.to(DEV))
    assert torch.equal(plain, qkv), "the store must not change what the op leaves in qkv"
    k_rows = qkv[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d)
    v_rows = qkv[:, (hq + hkv) * d:].reshape(n, hkv, d)
    if fp8:  # the same rows through the stand-alone quant-on-write store
        kb2, vb2 = torch.zeros_like(kb), torch.zeros_like(vb)
        ops.store_cache_fp8(k_rows.contiguous(), v_rows.contiguous(), ops._kv_layout(kb2, vb2, 1), loc.to(DEV), hkv, d, d,
                            size_limit=slots, k_scale=k_scale, v_scale=v_scale)
        torch.cuda.synchronize()
        assert torch.equal(kb.view(torch.uint8), kb2.view(torch.uint8)) and torch.equal(vb.view(torch.uint8), vb2.view(torch.uint8))
    else:
        for t in range(n):
            s = int(loc[t])
            if pool == "bf16_hnd_paged":
                kr, vr = kb[s // page, :, s % page], vb[s // page, :, s % page]
            else:
                kr, vr = kb[s], vb[s]
            if s == 0:
                assert not kr.any() and not vr.any()
            else:
                assert torch.equal(kr, k_rows[t]) and torch.equal(vr, v_rows[t]), t
    # an out-of-range slot raises the device error flag and writes nothing for that token
    bad = loc.clone()
    bad[3] = slots + 5
    ops.fused_qk_norm_rope(qkv0.clone().to(DEV), hq, hkv, hkv, d, 1e-6, qw, kw_, 10000.0, True, pos.to(DEV), layout=lay,
                           loc=bad.to(DEV), size_limit=slots, k_scale=k_scale, v_scale=v_scale, err_flag=err)
    torch.cuda.synchronize()
    assert int(err.item()) & L.RX_DEVERR_SLOT_OOB


def test_fused_qk_norm_rope_rejects_what_it_cannot_do():
    from sglang_amd import ops

    qkv = torch.zeros(4, (4 + 2 + 2) * 128, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(128, dtype=torch.bfloat16, device=DEV)
    pos = torch.zeros(4, dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):
        ops.fused_qk_norm_rope(qkv[:, :-1], 4, 2, 2, 128, 1e-6, w, w, 10000.0, True, pos)
    with pytest.raises(ValueError):
        ops.fused_qk_norm_rope(qkv, 4, 2, 2, 128, 1e-6, w[:64], w, 10000.0, True, pos)
    with pytest.raises(TypeError):
        ops.fused_qk_norm_rope(qkv.float(), 4, 2, 2, 128, 1e-6, w, w, 10000.0, True, pos)
    with pytest.raises(RuntimeError):
        ops.fused_qk_norm_rope(qkv, 4, 2, 2, 128, 1e-6, w, w, 10000.0, True, pos, rotary_dim=130)
    with pytest.raises(RuntimeError):
        ops.fused_qk_norm_rope(qkv, 4, 2, 2, 128, 1e-6, w, w, 0.0, True, pos)


def test_fused_qk_norm_rope_out_custom_op_matches_the_direct_call():
    """torch.ops.radix_hip.fused_qk_norm_rope_out: the reference's op name and argument order
    (kernels/ops/attention/fused_qknorm_rope.py:33-52), mutates qkv in place."""
    from sglang_amd import custom_ops  # noqa: F401  (registers the op)
    from sglang_amd import ops

    g = torch.Generator().manual_seed(3)
    qkv0 = torch.randn(19, (8 + 2 + 2) * 128, generator=g).to(torch.bfloat16).to(DEV)
    qw, kw_ = (torch.randn(128, generator=g)).to(torch.bfloat16).to(DEV), (torch.randn(128, generator=g)).to(torch.bfloat16).to(DEV)
    pos = torch.arange(19, dtype=torch.int32, device=DEV) + 7
    a, b = qkv0.clone(), qkv0.clone()
    torch.ops.radix_hip.fused_qk_norm_rope_out(a, qw, kw_, pos, 8, 2, 2, 128, 1e-6, 10000.0, False, 1.0, 0.0, 0.0, 1.0, 64)
    ops.fused_qk_norm_rope(b, 8, 2, 2, 128, 1e-6, qw, kw_, 10000.0, False, pos, 1.0, 0.0, 0.0, 1.0, 64)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, qkv0)


def test_fast_and_generic_kernels_agree_at_long_context_positions():
    """the fast kernel builds its frequencies as exp2(p * kf) with kf = -2 / rot * log2(base); with the approximate
    __log2f in the kernel the relative error of kf was multiplied by the position inside the angle and the two kernels
    (which one runs depends only on alignment) drifted apart at long contexts.  kf now comes from the host in double.
    Positions up to 200 k: the aligned call (fast kernel) and the same rows at a misaligned head row stride (generic
    kernel) agree, and both meet the fp64 oracle at the angle error the fp32 product pos * freq itself carries."""
    from sglang_amd import ops

    hq, hkv, d, rot, n = 4, 2, 128, 128, 64
    g = torch.Generator().manual_seed(3)
    qkv0 = torch.randn(n, (hq + 2 * hkv) * d, generator=g).to(torch.bfloat16)
    qw, kw_ = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16), (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16)
    pos = torch.cat([torch.randint(120_000, 200_000, (n - 2,), generator=g), torch.tensor([131071, 199999])]).to(torch.int32)
    outs = []
    for misalign in (False, True):
        if misalign:   # rows that start 2 bytes off a 16-byte boundary: the generic kernel
            buf = torch.zeros(n * (hq + 2 * hkv) * d + 1, dtype=torch.bfloat16, device=DEV)
            qkv = buf[1:].view(n, -1)
            qkv.copy_(qkv0)
        else:
            qkv = qkv0.clone().to(DEV)
        ops.fused_qk_norm_rope(qkv, hq, hkv, hkv, d, 1e-6, qw.to(DEV), kw_.to(DEV), 500000.0, True, pos.to(DEV), 1.0, 0.0, 0.0, 1.0, rot)
        torch.cuda.synchronize()
        outs.append(qkv.float().cpu().numpy())
    q = qkv0[:, : hq * d].reshape(n, hq, d)
    k = qkv0[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d)
    want_q, want_k = orc.fused_qk_norm_rope(_bits(q), _bits(k), _bits(qw), _bits(kw_), pos.numpy(), 1e-6, 500000.0, True, 1.0, 0.0, 0.0, 1.0, rot)
    # the angle pos * freq in fp32: relative 2^-23 of up to 2e5 rad -> ~0.025 rad at the lowest pair index; values are O(3)
    for got in outs:
        assert np.abs(got[:, : hq * d].reshape(n, hq, d) - want_q).max() < 0.12
        assert np.abs(got[:, hq * d: (hq + hkv) * d].reshape(n, hkv, d) - want_k).max() < 0.12
    assert np.abs(outs[0] - outs[1]).max() < 0.1
