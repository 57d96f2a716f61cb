"""Long causal extends of a GQA-4 model run GQA-packed by themselves (rx_extend32.hip: PLAIN instance with the packing
factor as a compile-time constant).  Packing only regroups the query rows into workgroups -- (token, q head of the group)
pairs instead of the tokens of one q head -- so outputs and LSEs must be bit-identical to the unpacked launch
(option ext32_autopack = 0) and inside the usual bound of the fp64 oracle (extend_attention_fwd,
kernels/ops/attention/extend_attention.py:664-812)."""
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


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("shape", [([1800, 2300], [300, 513]), ([0], [1100]), ([4000, 5, 1999], [256, 700, 257])],
                         ids=["two_requests", "no_prefix", "ragged"])
@pytest.mark.parametrize("heads", [(8, 2), (8, 1)], ids=["gqa4", "gqa8"])
def test_autopacked_extend_is_bit_identical_and_matches_oracle(dtype, shape, heads):
    from sglang_amd import ops

    prefix, extend = shape
    (hq, hkv), d, ps = heads, 128, 16
    g = torch.Generator().manual_seed(len(prefix) * 7 + sum(extend))
    npg = sum(-(-p // ps) for p in prefix) + 3
    perm = torch.randperm(npg - 1, generator=g) + 1
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
    sm = d ** -0.5
    outs = {}
    from sglang_amd import lib as rxlib

    for mode in ("1", "0"):
        with rxlib.option("ext32_autopack", int(mode)):
            o = torch.full((T, hq, d), float("nan"), dtype=dtype, device=DEV)
            lse = torch.zeros(T, hq, dtype=torch.float32, device=DEV)
            ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV),
                                     torch.from_numpy(qo).to(DEV), torch.tensor(kvp, dtype=torch.int32, device=DEV),
                                     kvi.to(DEV), None, True, None, max(extend), 1.0, 1.0, sm_scale=sm, lse_extend=lse,
                                     page_size=1, avg_kv_len_hint=int(np.mean(prefix)) + 2048)  # (the hint: the eight-wave launch)
            torch.cuda.synchronize()
            # the dispatch record says which instance ran: the packed PLAIN one (PKC = group) or the unpacked one
            assert rxlib.last_dispatch().endswith("true, %d>" % (hq // hkv if mode == "1" else 0)), rxlib.last_dispatch()
            outs[mode] = (o, lse)
    assert torch.equal(outs["1"][0].view(torch.int16), outs["0"][0].view(torch.int16))
    assert torch.equal(outs["1"][1], outs["0"][1])
    want = orc.extend_attention(_bits(q), _bits(ke), _bits(ve), _bits(kb), _bits(vb), qo, np.asarray(kvp, dtype=np.int32),
                                kvi.numpy(), sm_scale=sm)
    absw = None
    if dtype == torch.bfloat16:
        absw = orc.extend_attention(_bits(q), _bits(ke), parity.abs_values(_bits(ve)), _bits(kb), parity.abs_values(_bits(vb)), qo,
                                    np.asarray(kvp, dtype=np.int32), kvi.numpy(), sm_scale=sm)
    got = outs["1"][0].float().cpu().numpy()
    assert not np.isnan(got).any()
    parity.check_out(got, want, dtype, ("autopack", shape), ulps=1, absw=absw)
