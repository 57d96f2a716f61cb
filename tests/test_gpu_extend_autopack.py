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
        # (ext32_pack_min_wgs = 0: these batches are far below the chip-coverage gate of the self-packing)
        with rxlib.option("ext32_autopack", int(mode)), rxlib.option("ext32_pack_min_wgs", 0):
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


def test_self_packing_gates_tile_estimate_and_chip_coverage():
    """The launcher's gates (rx_extend32.hip, round 4): GQA-4 rows pack by themselves from four estimated tiles up
    (prefix hint + half the longest extend, in 64-token tiles) and only while the packed grid -- requests x kv heads x
    256-row blocks -- still covers the chip's CUs; below either gate the unpacked form runs.  Same bits both ways, for
    short extends over a prefix too (the case that gained most: 2 k + 64 tokens)."""
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    hq, hkv, d = 8, 2, 128
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def run(bs, P, E, **opts):
        g = torch.Generator().manual_seed(bs * 1000 + P + E)
        pool = bs * P + 1
        kb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
        vb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
        q = torch.randn(bs * E, hq, d, generator=g).to(torch.bfloat16).to(DEV)
        ke = torch.randn(bs * E, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
        ve = torch.randn(bs * E, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
        qo = torch.arange(bs + 1, dtype=torch.int64, device=DEV) * E
        kvp = torch.arange(bs + 1, dtype=torch.int32, device=DEV) * P
        kvi = torch.randperm(bs * P, generator=g).to(torch.int64).to(DEV) + 1
        o = torch.full((bs * E, hq, d), float("nan"), dtype=torch.bfloat16, device=DEV)
        ctx = [rxlib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, kvi, None, True, None, E, 1.0, 1.0, sm_scale=d ** -0.5,
                                     page_size=1, avg_kv_len_hint=P)
            torch.cuda.synchronize()
            return o, rxlib.last_dispatch()
        finally:
            for c in reversed(ctx):
                c.__exit__()

    big = -(-cus // hkv)                      # requests whose packed grid (one block each) just covers the chip
    o_p, name = run(big, 512, 64)
    assert name.endswith("4, false, true, 4>"), name          # 8 estimated tiles, grid >= CUs: packed, on four waves (round 5: < 24 tiles)
    o_8, name = run(big, 512, 64, ext32_pack4_tiles=0)
    assert name.endswith("8, false, true, 4>"), name          # ... that gate is an option: eight waves, same bits
    assert torch.equal(o_p.view(torch.int16), o_8.view(torch.int16))
    _, name = run(big, 1600, 64)
    assert name.endswith("8, false, true, 4>"), name          # 25 estimated tiles: packed, eight waves
    o_u, name = run(big, 512, 64, ext32_autopack=0)
    assert name.endswith("4, false, true, 0>"), name          # switched off: the four-wave unpacked form of that estimate
    assert torch.equal(o_p.view(torch.int16), o_u.view(torch.int16)) and not torch.isnan(o_p.float()).any()
    _, name = run(big - 8, 512, 64)
    assert name.endswith("true, 0>"), name                    # the packed grid would leave CUs idle: unpacked
    _, name = run(big, 64, 64)
    assert name.endswith("4, false, true, 0>"), name          # one estimated tile: next to nothing to do, unpacked
    _, name = run(big - 8, 512, 64, ext32_pack_min_wgs=0)
    assert name.endswith("4, false, true, 4>"), name          # the gate is an option


@pytest.mark.parametrize("dims", [(256, 256), (64, 64), (192, 128), (96, 96)], ids=["d256", "d64", "d192_128", "d96"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_short_extends_over_a_prefix_take_the_packed_template(dims, dtype):
    """Round 4: a short extend (a few new tokens per request -- draft / verify-sized, chunk tails, follow-up turns) at the
    AGPR template's head dims runs that template with the kv head's group packed into one row block, from four estimated
    tiles up (rx_extend_d256.hip: extend_d256_supports; 2.5-6x the per-head launch, tools/probe/short_ext.py); with
    next to no prefix, or with option extend_d256_min_rows above its rows, the call takes the short-extend kernel as
    before.  Both against the oracle at the bar."""
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    dk, dv = dims
    hq, hkv = 8, 2
    prefix, extend = [700, 333, 1025], [8, 1, 17]
    g = torch.Generator().manual_seed(dk + dv)
    pool = sum(prefix) + 1
    kb = torch.randn(pool, hkv, dk, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, dv, generator=g).to(dtype)
    T = sum(extend)
    q = torch.randn(T, hq, dk, generator=g).to(dtype)
    ke = torch.randn(T, hkv, dk, generator=g).to(dtype)
    ve = torch.randn(T, hkv, dv, generator=g).to(dtype)
    qo = np.concatenate([[0], np.cumsum(extend)]).astype(np.int64)
    kvp = np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)
    kvi = (torch.randperm(pool - 1, generator=g) + 1).numpy().astype(np.int64)
    sm = dk ** -0.5
    want = orc.extend_attention(_bits(q), _bits(ke), _bits(ve), _bits(kb), _bits(vb), qo, kvp, kvi, sm_scale=sm)
    absw = None
    if dtype == torch.bfloat16:
        absw = orc.extend_attention(_bits(q), _bits(ke), parity.abs_values(_bits(ve)), _bits(kb), parity.abs_values(_bits(vb)),
                                    qo, kvp, kvi, sm_scale=sm)

    def run(**opts):
        ctx = [rxlib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            o = torch.full((T, hq, dv), float("nan"), dtype=dtype, device=DEV)
            ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), torch.from_numpy(qo).to(DEV),
                                     torch.from_numpy(kvp).to(DEV), torch.from_numpy(kvi).to(DEV), None, True, None,
                                     max(extend), 1.0, 1.0, sm_scale=sm, page_size=1)
            torch.cuda.synchronize()
            return o.float().cpu().numpy(), rxlib.last_dispatch()
        finally:
            for c in reversed(ctx):
                c.__exit__()

    got, name = run()
    assert name.startswith("extend_d256_kernel<") and name.endswith("g4"), name      # 17 tokens x 4 heads = 68 rows: packed
    parity.check_out(got, want, dtype, ("short extend, packed template", dims), ulps=1, absw=absw)
    got, name = run(extend_d256_min_rows=129)
    assert not name.startswith("extend_d256_kernel<"), name                           # the per-head short-extend kernel
    parity.check_out(got, want, dtype, ("short extend, per-head kernel", dims), ulps=1, absw=absw)


def test_redo_counters_of_the_counting_instance():
    """rx_debug_counters + option ext32_count_redo: the counting twin of the bench's kernel instance returns the same bits
    and counts.  N(0, 1) scores: a redo on every wave's FIRST block only (reference max still -inf);
    scores that jump by +12 nats in the middle of the prefix: more."""
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    hq, hkv, d, ps, P, E, bs = 8, 2, 128, 16, 1024, 256, 2
    g = torch.Generator().manual_seed(4)
    npg = bs * (P // ps) + 2
    kb = torch.randn(npg, hkv, ps, d, generator=g).to(torch.bfloat16)      # an HND pool (the bench's layout: not linear)
    vb = torch.randn(npg, hkv, ps, d, generator=g).to(torch.bfloat16)
    perm = torch.randperm(npg - 1, generator=g) + 1
    kvi = torch.cat([(perm[i * (P // ps): (i + 1) * (P // ps)][:, None] * ps + torch.arange(ps)[None]).reshape(-1) for i in range(bs)])
    kvp = torch.arange(bs + 1, dtype=torch.int32) * P
    T = bs * E
    q = torch.randn(T, hq, d, generator=g).to(torch.bfloat16)
    ke = torch.randn(T, hkv, d, generator=g).to(torch.bfloat16)
    ve = torch.randn(T, hkv, d, generator=g).to(torch.bfloat16)
    qo = torch.arange(bs + 1, dtype=torch.int64) * E

    def run(kbuf, count):
        o = torch.zeros(T, hq, d, dtype=torch.bfloat16, device=DEV)
        with rxlib.option("ext32_count_redo", int(count)), rxlib.option("ext32_small_wg", 0), rxlib.option("ext32_pack_min_wgs", 0):
            kd, vd = kbuf.to(DEV), vb.to(DEV)
            ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kd, vd, qo.to(DEV), kvp.to(DEV), kvi.to(DEV), None, True,
                                     None, E, 1.0, 1.0, sm_scale=d ** -0.5, page_size=ps, kv_layout=ops.kv_layout_hnd(kd, vd))
            torch.cuda.synchronize()
            return o, rxlib.last_dispatch()

    rxlib.debug_counters(reset=True)
    o0, n0 = run(kb, False)
    assert n0.startswith("extend_mfma32_kernel") and rxlib.debug_counters() == (0, 0)
    o1, n1 = run(kb, True)
    assert n1.startswith("extend_mfma32_count_kernel") and torch.equal(o0, o1)
    blocks, redone = rxlib.debug_counters(reset=True)
    wgs = bs * hkv * (E * 4 // 256)                       # (request, kv head, 256-row block); 8 waves each
    assert redone == wgs * 8, (blocks, redone)            # the FIRST block of each wave (m = -inf), nothing else
    assert blocks >= wgs * 8 * 2 * (P // 64) and rxlib.debug_counters() == (0, 0)
    kb2 = kb.clone()
    half = kvi[P // 2:P]
    kb2[half // ps, :, half % ps, :] *= 4.0               # the second half of request 0's prefix: scores four times as large
    _, _ = run(kb2, True)
    _, redone2 = rxlib.debug_counters(reset=True)
    assert redone2 > redone
