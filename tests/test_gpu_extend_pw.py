"""rx::extend_pw_kernel (csrc/rx_extend_pw.hip): the four-waves x 64-rows form of the D = 128 extend, forced with
RX_EXT_PW=2 (the dispatcher re-reads the variable per call), against the fp64 oracle of extend_attention_fwd
(kernels/ops/attention/extend_attention.py:664-812) and against the eight-wave kernel on the same inputs.

Cases aim at the kernel's own machinery: the row-offset table and its pipeline (prefixes beyond two table blocks =
512 rows), the ring of four tile slots, runs of fully visible tiles entered / left at every parity, the causal
diagonal and ragged ends (boundary tiles), several 256-row query blocks, requests shorter than a wave, no prefix at
all, HND paged pools (page stride != page * token stride), int32 indices, non-causal / skip flags, LSE and sinks."""
import os

import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_amd import ops as _ops

    return _ops


@pytest.fixture(autouse=True)
def _force_pw():
    old = os.environ.get("RX_EXT_PW")
    os.environ["RX_EXT_PW"] = "2"
    yield
    if old is None:
        os.environ.pop("RX_EXT_PW", None)
    else:
        os.environ["RX_EXT_PW"] = old


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _make(dtype, prefix, extend, hq, hkv, ps, hnd, seed, idx_dtype=torch.int64):
    d = 128
    g = torch.Generator().manual_seed(seed)
    npg = sum(-(-p // ps) for p in prefix) + 3
    perm = (torch.randperm(npg - 1, generator=g) + 1)
    kvi, kvp, pi = [], [0], 0
    for p in prefix:
        n = -(-p // ps)
        pages = perm[pi: pi + n]
        pi += n
        kvi.append((pages[:, None] * ps + torch.arange(ps)[None]).reshape(-1)[:p])
        kvp.append(kvp[-1] + p)
    kvi = torch.cat(kvi) if sum(prefix) else torch.zeros(0, dtype=torch.int64)
    kb = torch.randn(npg * ps, hkv, d, generator=g).to(dtype)
    vb = torch.randn(npg * ps, hkv, d, generator=g).to(dtype)
    T = sum(extend)
    q = torch.randn(T, hq, d, generator=g).to(dtype)
    ke = torch.randn(T, hkv, d, generator=g).to(dtype)
    ve = torch.randn(T, hkv, d, generator=g).to(dtype)
    qo = np.concatenate([[0], np.cumsum(extend)]).astype(np.int64)
    dev = dict(q=q.to(DEV), ke=ke.to(DEV), ve=ve.to(DEV), qo=torch.from_numpy(qo).to(DEV),
               kvp=torch.tensor(kvp, dtype=torch.int32, device=DEV), kvi=kvi.to(idx_dtype).to(DEV))
    if hnd:  # [pages, Hkv, page, D]: a token's rows are page-strided
        dev["kb"] = kb.view(npg, ps, hkv, d).permute(0, 2, 1, 3).contiguous().to(DEV)
        dev["vb"] = vb.view(npg, ps, hkv, d).permute(0, 2, 1, 3).contiguous().to(DEV)
    else:
        dev["kb"], dev["vb"] = kb.to(DEV), vb.to(DEV)
    host = dict(q=_bits(q), ke=_bits(ke), ve=_bits(ve), kb=_bits(kb), vb=_bits(vb), qo=qo,
                kvp=np.asarray(kvp, dtype=np.int32), kvi=kvi.numpy())
    return dev, host, T


CASES = [
    # prefix lens, extend lens, Hq, Hkv, page, HND
    ("tiny_no_prefix", [0], [1], 4, 4, 1, False),
    ("one_wave_ragged", [0, 5], [63, 65], 4, 1, 16, False),
    ("diag_only_blocks", [0], [300], 4, 2, 16, True),
    ("prefix_runs_even_odd", [64, 128, 192, 257], [64, 1, 130, 7], 8, 2, 16, True),
    ("table_pipeline_long_prefix", [1400, 515], [77, 300], 4, 1, 16, True),
    ("two_query_blocks", [700], [512], 4, 1, 32, True),
    ("page1_linear", [333, 64], [200, 64], 8, 8, 1, False),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_pw_kernel_vs_oracle_and_eight_wave_kernel(ops, case, dtype):
    name, prefix, extend, hq, hkv, ps, hnd = case
    dev, host, T = _make(dtype, prefix, extend, hq, hkv, ps, hnd, seed=len(name))
    lay = ops.kv_layout_hnd(dev["kb"], dev["vb"]) if hnd else None
    sm = 128 ** -0.5
    outs = {}
    for mode in ("0", "2"):
        os.environ["RX_EXT_PW"] = mode
        o = torch.full((T, hq, 128), float("nan"), dtype=dtype, device=DEV)
        lse = torch.zeros(T, hq, dtype=torch.float32, device=DEV)
        ops.extend_attention_fwd(dev["q"], dev["ke"], dev["ve"], o, dev["kb"], dev["vb"], dev["qo"], dev["kvp"], dev["kvi"],
                                 None, True, None, max(extend), 1.0, 1.0, sm_scale=sm, lse_extend=lse, page_size=ps,
                                 kv_layout=lay)
        torch.cuda.synchronize()
        outs[mode] = (o.float().cpu().numpy(), lse.cpu().numpy())
    want, want_lse = orc.extend_attention(host["q"], host["ke"], host["ve"], host["kb"], host["vb"], host["qo"], host["kvp"],
                                          host["kvi"], sm_scale=sm, return_lse=True)
    absw = None
    if dtype == torch.bfloat16:  # bf16 P rounding on the first causal rows (parity_util.check_out)
        absw = orc.extend_attention(host["q"], host["ke"], parity.abs_values(host["ve"]), host["kb"],
                                    parity.abs_values(host["vb"]), host["qo"], host["kvp"], host["kvi"], sm_scale=sm)
    got, got_lse = outs["2"]
    assert not np.isnan(got).any(), name
    parity.check_out(got, want, dtype, ("pw", name), ulps=1, absw=absw)
    np.testing.assert_allclose(got_lse, want_lse, atol=5e-3, rtol=2e-3)
    # the two kernels round the same quantities the same way; they differ by the order of fp32 sums only
    ref8, _ = outs["0"]
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    assert np.abs(got - ref8).max() <= tol * max(1.0, np.abs(want).max()), (name, np.abs(got - ref8).max())


@pytest.mark.parametrize("flags", [dict(causal=False), dict(skip_prefix=True), dict(skip_extend=True, causal=False),
                                   dict(sinks=True), dict(idx32=True)],
                         ids=["non_causal", "skip_prefix", "skip_extend", "sinks", "int32_indices"])
def test_pw_kernel_flags(ops, flags):
    dtype, prefix, extend, hq, hkv, ps = torch.float16, [200, 321], [100, 260], 8, 2, 16
    dev, host, T = _make(dtype, prefix, extend, hq, hkv, ps, True, seed=7,
                         idx_dtype=torch.int32 if flags.get("idx32") else torch.int64)
    lay = ops.kv_layout_hnd(dev["kb"], dev["vb"])
    sm = 128 ** -0.5
    causal = flags.get("causal", True)
    sinks = torch.linspace(-1.0, 1.0, hq) if flags.get("sinks") else None
    o = torch.full((T, hq, 128), float("nan"), dtype=dtype, device=DEV)
    ops.extend_attention_fwd(dev["q"], dev["ke"], dev["ve"], o, dev["kb"], dev["vb"], dev["qo"], dev["kvp"], dev["kvi"],
                             None, causal, None, max(extend), 1.0, 1.0, sm_scale=sm, page_size=ps, kv_layout=lay,
                             skip_prefix=bool(flags.get("skip_prefix")), skip_extend=bool(flags.get("skip_extend")),
                             sinks=None if sinks is None else sinks.to(DEV))
    torch.cuda.synchronize()
    want = orc.extend_attention(host["q"], host["ke"], host["ve"], host["kb"], host["vb"], host["qo"], host["kvp"],
                                host["kvi"], is_causal=causal, sm_scale=sm, skip_prefix=bool(flags.get("skip_prefix")),
                                skip_extend=bool(flags.get("skip_extend")),
                                sinks=None if sinks is None else sinks.numpy().astype(np.float64))
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("pw flags", sorted(flags)), ulps=1)


def test_pw_kernel_forced_jump_path(ops):
    """The thresholded running max: keys whose scores GROW along the sequence force the rare branch (a block's maximum
    more than 2^8 above the row's reference: jump test -> max exchange -> O^T / l rescale) in the pipelined tiles
    (cdna_hip_programming.md rule 26: a data-dependent branch needs an input that takes it)."""
    dtype, hq, hkv, ps = torch.float16, 4, 1, 16
    prefix, extend = [1024], [128]
    dev, host, T = _make(dtype, prefix, extend, hq, hkv, ps, True, seed=11)
    # scale the cached keys of every 256-token stretch up: the scores of later stretches exceed the earlier maxima by > 8 log2 units
    P = prefix[0]
    ramp = torch.tensor([1.0, 4.0, 9.0, 16.0]).repeat_interleave(P // 4)
    slots = dev["kvi"].long()
    kb_host = torch.from_numpy(host["kb"]).float()   # fp16 case: host["kb"] is a numpy float16 array
    kb_host[slots.cpu()] *= ramp[:, None, None]
    kb16 = kb_host.to(dtype)
    npg = kb16.shape[0] // ps
    dev["kb"] = kb16.view(npg, ps, hkv, 128).permute(0, 2, 1, 3).contiguous().to(DEV)
    lay = ops.kv_layout_hnd(dev["kb"], dev["vb"])
    sm = 128 ** -0.5
    outs = {}
    for mode in ("0", "2"):
        os.environ["RX_EXT_PW"] = mode
        o = torch.full((T, hq, 128), float("nan"), dtype=dtype, device=DEV)
        ops.extend_attention_fwd(dev["q"], dev["ke"], dev["ve"], o, dev["kb"], dev["vb"], dev["qo"], dev["kvp"], dev["kvi"],
                                 None, True, None, max(extend), 1.0, 1.0, sm_scale=sm, page_size=ps, kv_layout=lay)
        torch.cuda.synchronize()
        outs[mode] = o.float().cpu().numpy()
    want = orc.extend_attention(host["q"], host["ke"], host["ve"], _bits(kb16), host["vb"], host["qo"], host["kvp"],
                                host["kvi"], sm_scale=sm)
    assert not np.isnan(outs["2"]).any()
    # scores of magnitude ~100 make the softmax one-hot up to a few near ties, where P's 11-bit rounding shows as up to
    # 2 ulp of the output: the bound is 2 ulp here, and the four-wave kernel may not be worse than the eight-wave one
    parity.check_out(outs["2"], want, dtype, "pw forced jumps", ulps=2)
    e8, e4 = np.abs(outs["0"] - want).max(), np.abs(outs["2"] - want).max()
    assert e4 <= 1.25 * e8 + 1e-4, (e4, e8)
