"""Fused-RoPE MLA decode (VERDICT r03 "next" 9; kernels/ops/attention/rocm_mla_decode_rope.py:45-439, called from
srt/models/deepseek_common/attention_forward_methods/forward_mla_fused_rope_rocm.py:177-215): q_pe and the newest token's
k_pe are rotated INSIDE rx::decode_mla_kernel.

* the reference's contract (the newest row is in the pool, k_pe not rotated; the rotated k_pe comes back in k_pe_tokens)
  against the golden vectors of the reference's own kernel (tests/golden/mla_rope.npz) and against the fp64 oracle;
* this library's one-launch form (k_new: RoPE + KV store + attention): the same output, and the pool row afterwards IS the
  rotated row."""
import os

import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _cases(npz):
    out = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        out.setdefault(case, {})[field] = npz[key]
    return out


def test_reference_contract_against_the_reference_kernels_golden(golden_dir):
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    for name, c in _cases(np.load(os.path.join(golden_dir, "mla_rope.npz"))).items():
        q, kb = T(c["q"]), T(c["kb"])
        bs, hq, _ = q.shape
        kb0 = kb.clone()
        o = torch.full((bs, hq, 512), float("nan"), dtype=torch.float16, device=DEV)
        kpe = torch.zeros(bs, 1, 64, dtype=torch.float16, device=DEV)
        S = int(c["splits"])
        ops.decode_attention_fwd_grouped_rope(q, kb, kb[..., :512], o, T(c["kv_indptr"]), T(c["kv_indices"]), kpe, 512, 64,
                                              T(c["cos_sin"]), T(c["positions"]), None, S, float(c["sm_scale"]),
                                              use_rope=True, is_neox_style=bool(c["neox"]))
        torch.cuda.synchronize()
        assert rxlib.last_dispatch().startswith("decode_mla_kernel<rx::F16"), rxlib.last_dispatch()
        assert torch.equal(kb, kb0)                                  # the reference's form leaves the pool alone
        want, _ = orc.decode_attention_grouped_rope(c["q"], c["kb"], c["kv_indptr"], c["kv_indices"], c["cos_sin"],
                                                    c["positions"], float(c["sm_scale"]), is_neox=bool(c["neox"]))
        # |V| twin: the values are the first 512 columns of the same rows -- take |.| of those columns only
        kabs = c["kb"].copy()
        kabs[..., :512] = np.abs(kabs[..., :512])
        absw = _absw_rope(c["q"], c["kb"], kabs, c, bool(c["neox"]))
        parity.check_out(_bits(o), want, torch.float16, (name, "vs oracle"), ulps=1, absw=absw)
        # the reference kernel's own fp16 output: both sides round the result and P to 16 bits
        parity.check_out(_bits(o), c["o"].astype(np.float64), torch.float16, (name, "vs reference kernel golden"), ulps=2,
                         absw=2 * absw)
        _, kpe_want = orc.decode_attention_grouped_rope(c["q"], c["kb"], c["kv_indptr"], c["kv_indices"], c["cos_sin"],
                                                        c["positions"], float(c["sm_scale"]), is_neox=bool(c["neox"]))
        got_k = _bits(kpe).astype(np.float64).reshape(bs, 64)
        assert np.abs(got_k - kpe_want).max() <= 2.0 ** -10 * max(1.0, np.abs(kpe_want).max())          # one fp16 rounding
        assert np.abs(got_k - c["k_pe_out"].astype(np.float64).reshape(bs, 64)).max() <= 2.0 ** -9     # vs the reference's rounding


def _absw_rope(q, kb, kb_absv, c, neox):
    """sum_j p_j |v_j| for check_out: the probabilities of the true problem (keys from kb), the values |kb[..., :512]|."""
    qf, kf = orc.to_f64(q), orc.to_f64(kb).reshape(kb.shape[0], -1)
    va = orc.to_f64(kb_absv).reshape(kb.shape[0], -1)[:, :512]
    pos = np.asarray(c["positions"])
    bs, hq, _ = qf.shape
    q_rot = qf.copy()
    q_rot[..., 512:] = orc.rope(qf[..., 512:], pos, c["cos_sin"], neox, 64)
    out = np.zeros((bs, hq, 512))
    for b in range(bs):
        idx = np.asarray(c["kv_indices"][int(c["kv_indptr"][b]): int(c["kv_indptr"][b + 1])]).astype(np.int64)
        if idx.size == 0:
            continue
        rows = kf[idx].copy()
        rows[-1, 512:] = orc.rope(rows[-1][None, None, 512:], pos[b: b + 1], c["cos_sin"], neox, 64)[0, 0]
        s_ = (q_rot[b] @ rows.T) * float(c["sm_scale"])
        p_ = np.exp(s_ - s_.max(axis=-1, keepdims=True))
        out[b] = (p_ / p_.sum(axis=-1, keepdims=True)) @ np.abs(va[idx])
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("neox", [True, False], ids=["neox", "gptj"])
@pytest.mark.parametrize("form", ["pool_row", "k_new"])
@pytest.mark.parametrize("geom", [(16, 1, 2), (128, 16, 8), (5, 16, 1)], ids=["h16_page1", "h128_page16_8splits", "h5_single"])
def test_fused_rope_decode_matches_oracle(dtype, neox, form, geom):
    """Ragged batch (a length-1 request, tile-crossing lengths, a request whose newest token opens a new tile), 16 / 128 /
    5 q heads (one, eight and a partial q block), page 1 and paged HND rows, single pass and split-KV, the cos / sin table
    in fp32 and in the 16-bit dtype; both forms of the newest row."""
    from sglang_amd import ops

    hq, page, S = geom
    rng = np.random.default_rng(hq * 7 + page + S)
    g = torch.Generator().manual_seed(hq + page)
    lens = np.array([1, 33, 700, 64, 2049, 65], dtype=np.int64)
    bs = len(lens)
    pages = [-(-int(n) // page) for n in lens]
    n_pages = sum(pages) + 3
    ids = rng.permutation(np.arange(1, n_pages))
    kvi, pi = [], 0
    for i, n in enumerate(lens):
        sl = (ids[pi: pi + pages[i], None] * page + np.arange(page)[None]).reshape(-1)[: int(n)]
        pi += pages[i]
        kvi.append(sl)
    kv_indices = np.concatenate(kvi).astype(np.int64)
    kv_indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    pool = n_pages * page
    kb = (torch.randn(pool, 1, 576, generator=g) * 0.5).to(dtype)
    q = (torch.randn(bs, hq, 576, generator=g) * 0.5).to(dtype)
    maxpos = 4096
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    fr = torch.outer(torch.arange(maxpos).float(), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1)
    if hq == 128:
        cache = cache.to(dtype)                                    # a 16-bit table, as a model in that dtype keeps it
    positions = torch.from_numpy(lens - 1 + rng.integers(0, 1000, size=bs))
    last = torch.from_numpy(np.array([kvi[i][-1] for i in range(bs)], dtype=np.int64))
    new_rows = kb[last, 0].clone()                                 # the step's rows, k_pe not rotated
    sm = 192 ** -0.5
    cache_o = cache.float().numpy() if cache.dtype != torch.float32 else cache.numpy()
    want, kpe_want = orc.decode_attention_grouped_rope(_bits(q), _bits(kb), kv_indptr, kv_indices, cache_o, positions.numpy(),
                                                       sm, is_neox=neox)
    kabs = kb.clone()
    kabs[..., :512] = kabs[..., :512].abs()
    absw = _absw_rope(_bits(q), _bits(kb), _bits(kabs), dict(positions=positions.numpy(), cos_sin=cache_o, kv_indices=kv_indices,
                                                           kv_indptr=kv_indptr, sm_scale=sm), neox)
    kd = kb.to(DEV)
    if form == "k_new":                                            # the pool does NOT hold the step's rows yet
        kd[last.to(DEV)] = 7.0
    if page > 1:                                                    # paged HND rows with two pad tokens per page
        kp = torch.zeros(n_pages, 1, page + 2, 576, dtype=dtype, device=DEV)
        kp[:, :, :page] = kd.view(n_pages, page, 1, 576).permute(0, 2, 1, 3)
        kview = kp[:, :, :page]
        lay = ops.kv_layout_hnd(kview, kview[..., :512])
        kargs = dict(page_size=page, kv_layout=lay)
        k_arg, v_arg = kview, kview[..., :512]
        rows_of = lambda: kview.permute(0, 2, 1, 3).reshape(pool, 576)  # noqa: E731
    else:
        kargs, k_arg, v_arg = {}, kd, kd[..., :512]
        rows_of = lambda: kd.view(pool, 576)  # noqa: E731
    before = rows_of().clone()
    o = torch.full((bs, hq, 512), float("nan"), dtype=dtype, device=DEV)
    kpe = torch.zeros(bs, 1, 64, dtype=dtype, device=DEV)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    ops.decode_attention_fwd_grouped_rope(q.to(DEV), k_arg, v_arg, o, T(kv_indptr), T(kv_indices), kpe, 512, 64, cache.to(DEV),
                                          positions.to(DEV), None, S, sm, use_rope=True, is_neox_style=neox,
                                          k_new=new_rows.to(DEV) if form == "k_new" else None, **kargs)
    torch.cuda.synchronize()
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("fused rope", form, geom, neox), ulps=1, absw=absw)
    u = 2.0 ** (-10 if dtype == torch.float16 else -7)
    got_k = kpe.float().cpu().numpy().reshape(bs, 64).astype(np.float64)
    assert np.abs(got_k - kpe_want).max() <= u * max(1.0, np.abs(kpe_want).max())      # one 16-bit rounding of the rotation
    after = rows_of()
    if form == "pool_row":
        assert torch.equal(after, before)
    else:
        lastd = last.to(DEV)
        untouched = torch.ones(pool, dtype=torch.bool, device=DEV)
        untouched[lastd] = False
        assert torch.equal(after[untouched], before[untouched])
        assert torch.equal(after[lastd, :512].cpu(), new_rows[:, :512])                 # the latent part, bit for bit
        assert torch.equal(after[lastd, 512:].cpu().view(torch.int16), kpe.view(bs, 64).cpu().view(torch.int16))  # the rotated k_pe


def test_fused_rope_rejects_what_it_cannot_do():
    from sglang_amd import ops
    from sglang_amd.lib import RadixHipError

    kb = torch.zeros(8, 1, 576, dtype=torch.float8_e4m3fn, device=DEV)
    q = torch.zeros(1, 16, 576, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(1, 16, 512, dtype=torch.bfloat16, device=DEV)
    cache = torch.zeros(16, 64, device=DEV)
    args = (torch.tensor([0, 3], dtype=torch.int32, device=DEV), torch.tensor([1, 2, 3], device=DEV),
            torch.zeros(1, 1, 64, dtype=torch.bfloat16, device=DEV), 512, 64, cache, torch.tensor([2], device=DEV), None, 1, 0.1)
    with pytest.raises(RadixHipError, match="16-bit latent rows"):
        ops.decode_attention_fwd_grouped_rope(q, kb, kb[..., :512], o, *args, use_rope=True)
    kb16 = torch.zeros(8, 1, 576, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RadixHipError, match="rotary_dim 64"):
        ops.decode_attention_fwd_grouped_rope(q, kb16, kb16[..., :512], o, args[0], args[1], args[2], 512, 32, cache, args[6],
                                              None, 1, 0.1, use_rope=True)
