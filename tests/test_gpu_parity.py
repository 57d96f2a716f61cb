"""GPU parity tests: the HIP path (through the C ABI, via sglang_amd.ops) against the CPU
oracle on identical seeded inputs, and against the golden vectors produced by the reference.

Bars: bit-exact for integer / byte / index work; attention within 1e-2 (the reference's own
bf16 tolerance, test_triton_attention_kernels.py:559) and -- the north-star bar -- max abs
error <= 2e-3 vs the fp64 oracle for fp16 inputs (fp16 output rounding alone is 1e-3 at |o|~1).
"""
import json
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


def _t(a, dtype=None):
    """numpy (uint16 bf16 bits | float16 | ints) -> cuda tensor."""
    if a.dtype == np.uint16:
        return torch.from_numpy(a.copy()).view(torch.bfloat16).to(DEV)
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _cases(npz):
    out = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        out.setdefault(case, {})[field] = npz[key]
    return out


# ---------------------------------------------------------------------------- K1 store
def test_store_kv_golden_bit_exact(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "store_kv.npz"))
    for ci in range(2):
        kc, vc = _t(z[f"c{ci}_kc_in"]), _t(z[f"c{ci}_vc_in"])
        ops.store_cache(_t(z[f"c{ci}_k"]), _t(z[f"c{ci}_v"]), kc, vc, _t(z[f"c{ci}_loc"]))
        assert np.array_equal(_np(kc), z[f"c{ci}_kc_out"])
        assert np.array_equal(_np(vc), z[f"c{ci}_vc_out"])


@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
@pytest.mark.parametrize("row", [1024, 96, 6])  # 2048-B rows (Llama TP1), 192-B, 12-B (4-byte path)
def test_store_kv_vs_oracle(ops, idt, row):
    g = torch.Generator().manual_seed(1)
    n, rows = 300, 1000
    k = torch.randn(n, row, generator=g).to(torch.bfloat16)
    v = torch.randn(n, row, generator=g).to(torch.bfloat16)
    kc = torch.randn(rows, row, generator=g).to(torch.bfloat16)
    vc = torch.randn(rows, row, generator=g).to(torch.bfloat16)
    loc = torch.randperm(rows, generator=g)[:n]
    loc[5] = 0  # the reserved slot is skipped
    kc_ref, vc_ref = _np(kc).copy(), _np(vc).copy()
    orc.store_kv(_np(k), _np(v), kc_ref, vc_ref, loc.numpy())
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    # strided source rows (as q/k/v slices of a fused qkv projection are)
    big = torch.zeros(n, 3 * row, dtype=torch.bfloat16, device=DEV)
    big[:, row:2 * row] = k.to(DEV)
    ops.store_cache(big[:, row:2 * row], v.to(DEV), kcd, vcd, loc.to(idt).to(DEV))
    assert np.array_equal(_np(kcd), kc_ref)
    assert np.array_equal(_np(vcd), vc_ref)


def test_store_kv_oob_is_dropped_and_flagged(ops):
    k = torch.ones(4, 64, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(8, 64, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros(8, 64, dtype=torch.bfloat16, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    loc = torch.tensor([1, 9, -3, 2], device=DEV)
    ops.store_cache(k, k, kc, vc, loc, err_flag=flag)
    torch.cuda.synchronize()
    assert flag.item() == 1
    assert kc[1].float().sum().item() == 64 and kc[2].float().sum().item() == 64
    assert kc[[0, 3, 4, 5, 6, 7]].float().abs().sum().item() == 0


# ---------------------------------------------------------------------------- K2 kv indices
def test_kv_indices_golden_bit_exact(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "kv_indices.npz"))
    for ci in range(3):
        for use_start in (0, 1):
            t = f"c{ci}_{use_start}_"
            lens = _t(z[t + "lens"])
            bs = lens.shape[0]
            kv_indptr = torch.full((bs + 1,), -1, dtype=torch.int32, device=DEV)
            for odt in (torch.int64, torch.int32):
                kv_indices = torch.full((int(z[t + "kv_indptr"][-1]),), -1, dtype=odt, device=DEV)
                ops.build_kv_indices(_t(z[t + "req_to_token"]), _t(z[t + "req_pool_indices"]), lens,
                                     kv_indptr, kv_indices,
                                     _t(z[t + "start"]) if use_start else None)
                assert np.array_equal(_np(kv_indptr), z[t + "kv_indptr"])
                assert np.array_equal(_np(kv_indices).astype(np.int64), z[t + "kv_indices"])


def test_kv_indices_reference_test_shapes(ops):
    """test_create_kvindices.py:72-77: BATCH 1/37/1786 of MAX_BATCH 4096 x CTX 4096."""
    rng = np.random.default_rng(0)
    max_batch = ctx = 4096
    r2t = torch.arange(max_batch * ctx, dtype=torch.int32, device=DEV).reshape(max_batch, ctx)
    for batch in (1, 37, 1786):
        rpi = rng.choice(max_batch, size=batch, replace=False).astype(np.int32)
        lens = rng.choice(ctx, size=batch, replace=False).astype(np.int32)
        kv_indptr = torch.zeros(batch + 1, dtype=torch.int32, device=DEV)
        kv_indices = torch.empty(int(lens.sum()), dtype=torch.int64, device=DEV)
        ops.build_kv_indices(r2t, _t(rpi), _t(lens), kv_indptr, kv_indices)
        want = np.concatenate([np.arange(r * ctx, r * ctx + n) for r, n in zip(rpi, lens)])
        assert np.array_equal(_np(kv_indices), want)
        assert np.array_equal(_np(kv_indptr)[1:], np.cumsum(lens))


# ---------------------------------------------------------------------------- K3 splits
def test_num_kv_splits_golden_bit_exact(ops, golden_dir):
    with open(os.path.join(golden_dir, "kv_splits.json")) as f:
        rows = json.load(f)
    for r in rows:
        seq = torch.tensor(r["seq_lens"], dtype=torch.int32, device=DEV)
        out = torch.zeros(len(r["seq_lens"]) * r["num_group"], dtype=torch.int32, device=DEV)
        ops.get_num_kv_splits(out, seq, r["hq"], r["hkv"], r["max_splits"], r["cores"])
        assert _np(out).tolist() == r["out"], {k: r[k] for k in ("hq", "hkv", "max_splits")}


# ---------------------------------------------------------------------------- K9 allocation
def test_alloc_kernels_golden_bit_exact(ops, golden_dir):
    with open(os.path.join(golden_dir, "alloc_sequences.json")) as f:
        cases = json.load(f)
    n_ext = n_dec = 0
    for case in cases:
        ps = case["page_size"]
        if ps == 1 or case["need_sort"]:
            continue  # need_sort may merge release pages inside the call; kernel inputs differ
        prev_free = list(range(1, case["size"] // ps + 1))  # the kernel reads the list BEFORE the call
        for ent in case["log"]:
            if ent["op"] == "alloc_extend" and ent["out"] is not None:
                fp = torch.tensor(prev_free, dtype=torch.int64, device=DEV)
                out = torch.full((len(ent["out"]),), -1, dtype=torch.int64, device=DEV)
                ops.alloc_extend(torch.tensor(ent["prefix_lens"], device=DEV),
                                 torch.tensor(ent["seq_lens"], device=DEV),
                                 torch.tensor(ent["last_loc"], device=DEV), fp, out, ps)
                assert _np(out).tolist() == ent["out"]
                n_ext += 1
            elif ent["op"] == "alloc_decode" and ent["out"] is not None:
                fp = torch.tensor(prev_free, dtype=torch.int64, device=DEV)
                out = torch.full((len(ent["out"]),), -1, dtype=torch.int64, device=DEV)
                ops.alloc_decode(torch.tensor(ent["seq_lens"], device=DEV),
                                 torch.tensor(ent["last_loc"], device=DEV), fp, out, ps)
                assert _np(out).tolist() == ent["out"]
                n_dec += 1
            prev_free = ent["free"][0]
    assert n_ext >= 8 and n_dec >= 8, (n_ext, n_dec)


# ---------------------------------------------------------------------------- K4-K6 decode
def _run_decode(ops, c, mode, dtype_t):
    q, kb, vb = _t(c["q"]), _t(c["kb"]), _t(c["vb"])
    bs, hq, _ = q.shape
    dv = vb.shape[-1]
    o = torch.zeros(bs, hq, dv, dtype=q.dtype, device=DEV)
    kv_indptr = _t(c["kv_indptr"])
    cap = float(c["logit_cap"]) if "logit_cap" in c else 0.0
    if mode == "split":
        S = int(c["max_splits"])
        al = torch.zeros(bs, hq, S, dv, dtype=torch.float32, device=DEV)
        lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
        ops.decode_attention_fwd(q, kb, vb, o, kv_indptr, _t(c["kv_indices"]), al, lse,
                                 _t(c["nsplit"]), S, float(c["sm_scale"]), 1.0, 1.0, logit_cap=cap)
        # the stage-1 partials for inspection: a request with ONE split writes its output straight from stage 1 in the
        # two-stage call (no partial row), so they come from an explicit stage-1-only call
        al.zero_(); lse.zero_()
        ops.decode_attention_fwd(q, kb, vb, torch.empty_like(o), kv_indptr, _t(c["kv_indices"]), al, lse,
                                 _t(c["nsplit"]), S, float(c["sm_scale"]), 1.0, 1.0, logit_cap=cap, stages=1)
        return o, al, lse
    ops.decode_attention_fwd(q, kb, vb, o, kv_indptr, _t(c["kv_indices"]).to(torch.int32), None, None,
                             None, 1, float(c["sm_scale"]), 1.0, 1.0, logit_cap=cap)
    return o, None, None


def test_decode_golden_fp16(ops, golden_dir):
    cases = _cases(np.load(os.path.join(golden_dir, "decode.npz")))
    for name, c in cases.items():
        cap = float(c["logit_cap"]) if "logit_cap" in c else 0.0
        want = orc.decode_attention(c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"],
                                    float(c["sm_scale"]), logit_cap=cap)
        for mode in ("split", "single"):
            o, al, lse = _run_decode(ops, c, mode, torch.float16)
            got = _np(o).astype(np.float64)
            # vs the reference's own output, its own tolerance
            np.testing.assert_allclose(got, c["o"].astype(np.float64), atol=1e-2, rtol=1e-2,
                                       err_msg=f"{name}/{mode} vs triton golden")
            # vs the fp64 oracle, north-star bar
            parity.check_out(got, want, torch.float16, (name, mode))
            if mode == "split":
                logits, lse_s, _ = orc.decode_attention_split(
                    c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"], c["nsplit"],
                    int(c["max_splits"]), float(c["sm_scale"]), logit_cap=cap)
                w = ~np.isnan(lse_s)
                np.testing.assert_allclose(_np(lse)[w], lse_s[w], atol=2e-3, rtol=1e-3, err_msg=name)
                np.testing.assert_allclose(_np(al)[w], logits[w], atol=3e-3, rtol=1e-2, err_msg=name)


def _make_paged_case(rng, bs, hq, hkv, d, lens, page_size, dtype, layout="shuffled"):
    """Pools + req_to_token with page-granular slot layouts (kit/dense_attention.py:593-712)."""
    max_ctx = int(max(lens)) + page_size
    pages_per_req = [(int(n) + page_size - 1) // page_size for n in lens]
    n_pages = sum(pages_per_req) + 3
    page_ids = np.arange(1, n_pages)  # page 0 reserved
    if layout == "shuffled":
        page_ids = rng.permutation(page_ids)
    elif layout == "interleaved":
        page_ids = np.concatenate([page_ids[0::2], page_ids[1::2]])
    r2t = np.zeros((bs + 1, max_ctx), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        slots = []
        for _ in range(pages_per_req[i]):
            p = page_ids[pi]; pi += 1
            slots.extend(range(p * page_size, (p + 1) * page_size))
        r2t[i + 1, : int(n)] = slots[: int(n)]
    pool = n_pages * page_size
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    return q, kb, vb, r2t, rpi


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (8, 1, 128), (4, 4, 64), (12, 12, 64), (16, 1, 128), (40, 2, 128)])
@pytest.mark.parametrize("page_size,layout", [(1, "shuffled"), (16, "shuffled"), (16, "interleaved"), (32, "contiguous")])
def test_decode_paged_vs_oracle(ops, dtype, hq, hkv, d, page_size, layout):
    rng = np.random.default_rng(hq * 131 + d + page_size)
    lens = np.array([1, 31, 32, 33, 257, 500, 64], dtype=np.int64)
    bs = len(lens)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, bs, hq, hkv, d, lens, page_size, dtype, layout)
    sm = 1.0 / d ** 0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want = orc.decode_attention(_np(q), _np(kb), _np(vb), kv_indptr, kv_indices, sm)
    qd, kbd, vbd = q.to(DEV), kb.to(DEV), vb.to(DEV)
    r2td, rpid, lensd = _t(r2t), _t(rpi), _t(lens)
    # (b) native mode: in-kernel req_to_token walk, single pass
    o = torch.zeros(bs, hq, d, dtype=dtype, device=DEV)
    ops.decode_attention_fwd_paged(qd, kbd, vbd, o, r2td, rpid, lensd, None, None, None, 1, sm,
                                   page_size=page_size)
    parity.check_out(_np(o.float()), want, dtype, "paged/single")
    # (a) reference contract: kv_indices + K3-chosen splits + stage 2
    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, lensd.to(torch.int32), hq, hkv, S, 256)
    kvi = torch.empty(int(lens.sum()), dtype=torch.int64, device=DEV)
    kvp = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    ops.build_kv_indices(r2td, rpid, lensd, kvp, kvi)
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o2 = torch.zeros_like(o)
    ops.decode_attention_fwd(qd, kbd, vbd, o2, kvp, kvi, al, lse, nsplit, S, sm, 1.0, 1.0,
                             page_size=page_size)
    parity.check_out(_np(o2.float()), want, dtype, "indices/split")
    # stage 2 inside the stage-1 kernel (merge_counters): the last workgroup of a head block merges -- the SAME
    # arithmetic, so the same bits; three calls in a row prove the counters come back to zero
    cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
    for rep in range(3):
        o3 = torch.full_like(o, float("nan"))
        ops.decode_attention_fwd(qd, kbd, vbd, o3, kvp, kvi, al, lse, nsplit, S, sm, 1.0, 1.0,
                                 page_size=page_size, merge_counters=cnt)
        assert torch.equal(o3, o2), (rep, (o3.float() - o2.float()).abs().max().item())
        assert int(cnt.abs().sum()) == 0
    o4 = torch.full_like(o, float("nan"))
    ops.decode_attention_fwd_paged(qd, kbd, vbd, o4, r2td, rpid, lensd, al, lse, nsplit, S, sm,
                                   page_size=page_size, merge_counters=cnt)
    assert torch.equal(o4, o2) and int(cnt.abs().sum()) == 0


def test_decode_hnd_layout_and_sinks(ops):
    rng = np.random.default_rng(3)
    hq, hkv, d, ps = 8, 2, 128, 16
    lens = np.array([40, 17, 129], dtype=np.int64)
    bs = len(lens)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, bs, hq, hkv, d, lens, ps, torch.float16)
    sinks = torch.randn(hq)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want, absw = parity.want_and_absw(orc.decode_attention, (_np(q), _np(kb), _np(vb), kv_indptr, kv_indices, 0.1), (2,),
                                      sinks=sinks.numpy())
    pages = kb.shape[0] // ps
    k_hnd = kb.view(pages, ps, hkv, d).permute(0, 2, 1, 3).contiguous().to(DEV)
    v_hnd = vb.view(pages, ps, hkv, d).permute(0, 2, 1, 3).contiguous().to(DEV)
    lay = ops.kv_layout_hnd(k_hnd, v_hnd)
    o = torch.zeros(bs, hq, d, dtype=torch.float16, device=DEV)
    ops.decode_attention_fwd_paged(q.to(DEV), k_hnd, v_hnd, o, _t(r2t), _t(rpi), _t(lens), None, None,
                                   None, 1, 0.1, sinks=sinks.to(DEV), kv_layout=lay)
    parity.check_out(_np(o.float()), want, torch.float16, "hnd + sinks", ulps=1, absw=absw)


@pytest.mark.parametrize("hq,hkv,dk,dv", [(4, 4, 80, 80), (4, 4, 13, 13), (16, 1, 96, 96), (16, 1, 576, 512)])
def test_decode_generic_head_dims(ops, hq, hkv, dk, dv):
    """Odd head dims of the reference's tests (test_triton_attention_kernels.py:563-573, :663-676)."""
    g = torch.Generator().manual_seed(42)
    B, S = 2, 100
    q = torch.randn(B, hq, dk, generator=g).to(torch.bfloat16)
    kb = torch.randn(B * S, hkv, dk, generator=g).to(torch.bfloat16)
    vb = torch.randn(B * S, hkv, dv, generator=g).to(torch.bfloat16)
    kv_indptr = np.array([0, S, 2 * S], dtype=np.int32)
    kv_indices = np.arange(B * S)
    sm = 1.0 / dk ** 0.5
    want, absw = parity.want_and_absw(orc.decode_attention, (_np(q), _np(kb), _np(vb), kv_indptr, kv_indices, sm), (2,))
    o = torch.zeros(B, hq, dv, dtype=torch.bfloat16, device=DEV)
    nsplit = torch.full((B,), 4, dtype=torch.int32, device=DEV)
    al = torch.zeros(B, hq, 8, dv, dtype=torch.float32, device=DEV)
    lse = torch.zeros(B, hq, 8, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o, _t(kv_indptr), _t(kv_indices), al, lse,
                             nsplit, 8, sm, 1.0, 1.0)
    parity.check_out(_np(o.float()), want, torch.bfloat16, ("generic decode", dk, dv), ulps=1, absw=absw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("mode", ["indices_splits", "paged_single", "paged_hnd_splits"])
@pytest.mark.parametrize("d", [256, 96])
def test_decode_head_dim_256(ops, dtype, mode, d):
    """D = 256 (Gemma-class heads) and D = 96 (Phi-3-class) on the MFMA split-KV decode kernel (the reference's kernel takes any Lk; here
    head dims other than 64 / 128 ran the scalar generic kernel until round 2): ragged lengths incl. 1 and tile
    crossings, GQA 8 / 2 and 16 q heads per kv head, kv splits, paged HND pool -- vs the fp64 oracle."""
    rng = np.random.default_rng(256)
    hq, hkv, ps = 16, 2, 16
    lens = np.array([1, 31, 32, 33, 200, 517], dtype=np.int64)
    bs = len(lens)
    n_pages = int(sum(-(-int(n) // ps) for n in lens)) + 2
    pool = n_pages * ps
    ids = rng.permutation(np.arange(1, n_pages))
    ctx = int(lens.max()) + ps
    r2t = np.zeros((bs + 1, ctx), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        k_ = -(-int(n) // ps)
        row = (ids[pi: pi + k_, None] * ps + np.arange(ps)[None]).reshape(-1)[: int(n)]
        pi += k_
        r2t[i + 1, : len(row)] = row
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    g = torch.Generator().manual_seed(6)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    sm = d ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want = orc.decode_attention(_np(q), _np(kb), _np(vb), kv_indptr, kv_indices, sm)
    o = torch.zeros(bs, hq, d, dtype=dtype, device=DEV)
    S = 8
    nsplit = torch.tensor([1, 1, 2, 2, 4, 8], dtype=torch.int32, device=DEV)
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    if mode == "indices_splits":
        ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o, _t(kv_indptr), _t(kv_indices), al, lse, nsplit, S,
                                 sm, 1.0, 1.0, page_size=1)
    elif mode == "paged_single":
        ops.decode_attention_fwd_paged(q.to(DEV), kb.to(DEV), vb.to(DEV), o, _t(r2t), _t(rpi), _t(lens), None, None, None,
                                       1, sm, page_size=ps)
    else:
        k_hnd = kb.to(DEV).view(n_pages, ps, hkv, d).permute(0, 2, 1, 3).contiguous()
        v_hnd = vb.to(DEV).view(n_pages, ps, hkv, d).permute(0, 2, 1, 3).contiguous()
        ops.decode_attention_fwd_paged(q.to(DEV), k_hnd, v_hnd, o, _t(r2t), _t(rpi), _t(lens), al, lse, nsplit, S, sm,
                                       page_size=ps, kv_layout=ops.kv_layout_hnd(k_hnd, v_hnd))
    parity.check_out(_np(o.float()), want, dtype, ("decode_d256", d, mode))
    if mode == "indices_splits":   # stage 2 inside the stage-1 kernel: the same bits, counters back at zero
        cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
        for rep in range(2):
            o2 = torch.full_like(o, float("nan"))
            ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o2, _t(kv_indptr), _t(kv_indices), al, lse, nsplit,
                                     S, sm, 1.0, 1.0, page_size=1, merge_counters=cnt)
            assert torch.equal(o2, o), (rep, (o2.float() - o.float()).abs().max().item())
            assert int(cnt.abs().sum()) == 0


# ---------------------------------------------------------------------------- K7 extend
def _run_extend(ops, c, with_lse=True):
    q, ke, ve, kb, vb = (_t(c[k]) for k in ("q", "k_ext", "v_ext", "kb", "vb"))
    o = torch.zeros_like(q)
    lse = torch.zeros(q.shape[0], q.shape[1], dtype=torch.float32, device=DEV) if with_lse else None
    qo = _t(c["qo_indptr"])
    max_len = int(np.diff(c["qo_indptr"]).max())
    ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, _t(c["kv_indptr"]), _t(c["kv_indices"]), None,
                             bool(c["causal"]), None, max_len, 1.0, 1.0, sm_scale=float(c["sm_scale"]),
                             logit_cap=float(c["logit_cap"]), lse_extend=lse)
    return o, lse


def test_extend_golden_fp16(ops, golden_dir):
    cases = _cases(np.load(os.path.join(golden_dir, "extend.npz")))
    for name, c in cases.items():
        o, lse = _run_extend(ops, c)
        got = _np(o).astype(np.float64)
        np.testing.assert_allclose(got, c["o"].astype(np.float64), atol=1e-2, rtol=1e-2,
                                   err_msg=f"{name} vs triton golden")
        want, want_lse = orc.extend_attention(
            c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"],
            c["kv_indices"], is_causal=bool(c["causal"]), sm_scale=float(c["sm_scale"]),
            logit_cap=float(c["logit_cap"]), return_lse=True)
        parity.check_out(got, want, torch.float16, name)
        np.testing.assert_allclose(_np(lse), want_lse, atol=2e-3, rtol=1e-3, err_msg=name)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,hkv,d", [(8, 2, 128), (4, 4, 64), (4, 1, 128)])
@pytest.mark.parametrize("window", [-1, 7])
def test_extend_ragged_vs_oracle(ops, dtype, hq, hkv, d, window):
    """zero-prefix / exact-page / cross-page / ragged cases (kit/dense_attention.py:102-215)."""
    rng = np.random.default_rng(d + hq + (window > 0))
    pre = np.array([0, 16, 33, 130, 5], dtype=np.int32)
    ext = np.array([1, 32, 50, 140, 129], dtype=np.int32)
    bs, T = len(pre), int(ext.sum())
    total = int((pre + ext).sum())
    pool = total + 5
    slots = rng.permutation(pool - 1)[:total] + 1
    g = torch.Generator().manual_seed(7)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(T, hq, d, generator=g).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.empty(int(pre.sum()), dtype=np.int64)
    ext_slots = np.empty(T, dtype=np.int64)
    so = 0
    for i in range(bs):
        s = slots[so: so + pre[i] + ext[i]]; so += pre[i] + ext[i]
        kv_indices[kv_indptr[i]: kv_indptr[i + 1]] = s[: pre[i]]
        ext_slots[qo[i]: qo[i + 1]] = s[pre[i]:]
    ke, ve = kb[ext_slots], vb[ext_slots]
    want, want_lse = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr,
                                          kv_indices, sm_scale=1.0 / d ** 0.5,
                                          sliding_window_size=window, return_lse=True)
    o = torch.zeros(T, hq, d, dtype=dtype, device=DEV)
    lse = torch.zeros(T, hq, dtype=torch.float32, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo),
                             _t(kv_indptr), _t(kv_indices), None, True, None, int(ext.max()), 1.0, 1.0,
                             sliding_window_size=window, lse_extend=lse)
    absw = orc.extend_attention(_np(q), _np(ke), parity.abs_values(_np(ve)), _np(kb), parity.abs_values(_np(vb)), qo,
                                kv_indptr, kv_indices, sm_scale=1.0 / d ** 0.5, sliding_window_size=window)
    parity.check_out(_np(o.float()), want, dtype, (hq, hkv, d, window), absw=absw)
    np.testing.assert_allclose(_np(lse), want_lse, atol=5e-3, rtol=2e-3)


@pytest.mark.parametrize("d", [80, 13, 96])
def test_extend_generic_head_dims(ops, d):
    rng = np.random.default_rng(d)
    hq, hkv = 4, 2
    pre = np.array([9, 0], dtype=np.int32); ext = np.array([20, 70], dtype=np.int32)
    T, total = int(ext.sum()), int((pre + ext).sum())
    g = torch.Generator().manual_seed(d)
    kb = torch.randn(total + 1, hkv, d, generator=g).to(torch.bfloat16)
    vb = torch.randn(total + 1, hkv, d, generator=g).to(torch.bfloat16)
    q = torch.randn(T, hq, d, generator=g).to(torch.bfloat16)
    kv_indptr = np.array([0, 9, 9], dtype=np.int32); qo = np.array([0, 20, 90], dtype=np.int64)
    kv_indices = np.arange(1, 10, dtype=np.int64)
    ext_slots = np.concatenate([np.arange(10, 30), np.arange(30, 100)])
    ke, ve = kb[ext_slots], vb[ext_slots]
    want = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices,
                                sm_scale=1.0 / d ** 0.5)
    o = torch.zeros(T, hq, d, dtype=torch.bfloat16, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo),
                             _t(kv_indptr), _t(kv_indices), None, True, None, 70, 1.0, 1.0)
    np.testing.assert_allclose(_np(o.float()), want, atol=1.5e-2, rtol=1e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("dk,dv", [(256, 256), (192, 128), (192, 192), (96, 96)])
@pytest.mark.parametrize("variant", ["plain", "window", "window_100_noncausal", "cap", "cap_sinks_noncausal", "hnd_pool",
                                     "short_extends", "max_jumps", "mha"])
def test_extend_mfma_other_head_dims(ops, dtype, dk, dv, variant):
    """rx::extend_nd_kernel (MFMA 16x16x32 for head dims 256 / 192+128 / 192 / 96 -- the shapes the reference retunes
    for gfx950, extend_attention.py:66-77, and the MLA prefill shape) vs the fp64 oracle: ragged batch with zero /
    tile-crossing prefixes and extends, GQA, LSE; sliding window; logit cap + sinks + non-causal; a paged HND pool."""
    rng = np.random.default_rng(dk + dv)
    # ("mha": one q head per kv head -- Gemma-7B-class -- takes the row -> token map of the 256-row kernels through its
    # G = 1 case, which has no 32-bit multiply-high magic; found broken in round 3)
    hq, hkv = (4, 4) if variant == "mha" else (8, 2)
    pre = np.array([0, 16, 33, 200, 5, 64], dtype=np.int32)
    # the longest extend picks the kernel form: > 64 (Dv > 128) / > 128 rows -> eight waves and 64-token tiles for
    # Dk > 128, else the four-wave form ("short_extends" runs every head dim through that one)
    ext = np.array([1, 32, 50, 140, 129, 64] if variant != "short_extends" else [1, 32, 50, 40, 29, 64], dtype=np.int32)
    if variant == "mha":
        ext = np.array([1, 32, 50, 300, 129, 257], dtype=np.int32)  # (> 128 rows per kv head: the 256-row kernel)
    bs, T = len(pre), int(ext.sum())
    total = int((pre + ext).sum())
    ps = 16
    n_pages = (total + ps - 1) // ps + 3
    pool = n_pages * ps
    slots = rng.permutation(pool - ps)[:total] + ps          # page 0 reserved
    g = torch.Generator().manual_seed(7)
    kb = torch.randn(pool, hkv, dk, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, dv, generator=g).to(dtype)
    q = torch.randn(T, hq, dk, generator=g).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.empty(int(pre.sum()), dtype=np.int64)
    ext_slots = np.empty(T, dtype=np.int64)
    so = 0
    for i in range(bs):
        s_ = slots[so: so + pre[i] + ext[i]]; so += pre[i] + ext[i]
        kv_indices[kv_indptr[i]: kv_indptr[i + 1]] = s_[: pre[i]]
        ext_slots[qo[i]: qo[i + 1]] = s_[pre[i]:]
        if variant == "max_jumps":   # keys that grow along the sequence: the running max jumps by far more than 2^8 several
            n = len(s_)              # times per row (the kernels with AGPR accumulators rescale only on such jumps)
            kb[torch.from_numpy(s_.copy())] *= torch.linspace(0.05, 3.0, n).pow(3).view(n, 1, 1).to(dtype)
    ke, ve = kb[ext_slots], vb[ext_slots]
    kw, okw = {}, {}
    causal = True
    if variant == "window":
        kw = okw = dict(sliding_window_size=9)
    elif variant == "window_100_noncausal":
        kw = okw = dict(sliding_window_size=100)
        causal = False
    elif variant == "cap":
        kw = okw = dict(logit_cap=20.0)
    elif variant == "cap_sinks_noncausal":
        sinks = torch.randn(hq, generator=g)
        causal = False
        kw = dict(logit_cap=20.0, sinks=sinks.to(DEV))
        okw = dict(logit_cap=20.0, sinks=sinks.numpy())
    want, want_lse = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices,
                                          is_causal=causal, sm_scale=1.0 / dk ** 0.5, return_lse=True, **okw)
    o = torch.zeros(T, hq, dv, dtype=dtype, device=DEV)
    lse = torch.zeros(T, hq, dtype=torch.float32, device=DEV)
    kbd, vbd, lay = kb.to(DEV), vb.to(DEV), None
    if variant == "hnd_pool":     # [pages, Hkv, page, D]: same slots, page / offset addressing
        kbd = kbd.view(n_pages, ps, hkv, dk).permute(0, 2, 1, 3).contiguous()
        vbd = vbd.view(n_pages, ps, hkv, dv).permute(0, 2, 1, 3).contiguous()
        lay = ops.kv_layout_hnd(kbd, vbd)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kbd, vbd, _t(qo), _t(kv_indptr), _t(kv_indices),
                             None, causal, None, int(ext.max()), 1.0, 1.0, lse_extend=lse,
                             page_size=ps if variant == "hnd_pool" else 1, kv_layout=lay, **kw)
    absw = orc.extend_attention(_np(q), _np(ke), parity.abs_values(_np(ve)), _np(kb), parity.abs_values(_np(vb)), qo,
                                kv_indptr, kv_indices, is_causal=causal, sm_scale=1.0 / dk ** 0.5, **okw)
    parity.check_out(_np(o.float()), want, dtype, (dk, dv, variant), absw=absw)
    if variant != "cap_sinks_noncausal":  # (the LSE output leaves the sink out, as the reference's does)
        np.testing.assert_allclose(_np(lse), want_lse, atol=5e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("variant", ["aliased", "own_v", "own_v_new_tokens", "noncausal", "skip_prefix", "skip_extend",
                                     "many_requests_int32", "paged_pool", "max_jumps", "max_jumps_own_v"])
def test_extend_mla_latent_shape(ops, dtype, variant):
    """rx::extend_mla_kernel (q 576 against one latent kv head, v = the first 512 columns: the absorbed-MLA extend of
    triton_backend.py:1290-1437 / extend_attention.py:241-661 at Lq 576, Lv 512) vs the fp64 oracle.  Ragged batch:
    zero / one-tile / slot-block-crossing prefixes (256-token id blocks, a ring of four tiles), extends from 1 token
    (16 packed rows) to several workgroups, an empty request.  'aliased' is how the pool always looks (v_buffer a view
    of the latent rows) with the new tokens' v a view of their k; the own_v variants give v tensors of their own (the
    kernel's two-image form), with DIFFERENT values so that a kernel reading k for v would fail."""
    rng = np.random.default_rng(576)
    hq, dk, dv = 16, 576, 512
    if variant == "many_requests_int32":   # >= 8 requests: requests bound to XCDs, a partial last group
        pre = np.array([0, 16, 33, 290, 5, 64, 31, 32, 100, 7, 0], dtype=np.int32)
        ext = np.array([1, 32, 5, 40, 9, 17, 0, 3, 64, 2, 8], dtype=np.int32)
    else:
        pre = np.array([0, 16, 33, 700, 5, 300], dtype=np.int32)
        ext = np.array([1, 32, 50, 140, 0, 64], dtype=np.int32)
    bs, T = len(pre), int(ext.sum())
    total = int((pre + ext).sum())
    ps = 16 if variant == "paged_pool" else 1
    n_pages = (total + ps - 1) // ps + 3
    pool = n_pages * ps
    slots = rng.permutation(pool - ps)[:total] + ps
    g = torch.Generator().manual_seed(11)
    kb = (torch.randn(pool, 1, dk, generator=g) * 0.5).to(dtype)
    if variant.startswith("max_jumps"):  # keys that grow along the pool: the running max jumps by far more than 2^8
        kb *= torch.linspace(0.05, 2.5, pool).pow(3).view(pool, 1, 1).to(dtype)
        slots = np.sort(slots[: int((pre + ext).sum())])  # ascending slots = ascending key norms along every sequence
    own_pool_v = variant == "own_v"
    vb = (torch.randn(pool, 1, dv, generator=g).to(dtype)) if own_pool_v else kb[..., :dv]
    q = torch.randn(T, hq, dk, generator=g).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.empty(int(pre.sum()), dtype=np.int64)
    ext_slots = np.empty(T, dtype=np.int64)
    so = 0
    for i in range(bs):
        s_ = slots[so: so + pre[i] + ext[i]]; so += pre[i] + ext[i]
        kv_indices[kv_indptr[i]: kv_indptr[i + 1]] = s_[: pre[i]]
        ext_slots[qo[i]: qo[i + 1]] = s_[pre[i]:]
    ke = kb[ext_slots].contiguous()
    if variant in ("own_v", "own_v_new_tokens", "max_jumps_own_v"):
        ve = torch.randn(T, 1, dv, generator=g).to(dtype)
    else:
        ve = ke[..., :dv]
    causal = variant != "noncausal"
    okw = dict(skip_prefix=True) if variant == "skip_prefix" else dict(skip_extend=True) if variant == "skip_extend" else {}
    sm = 1.0 / (128 + 64) ** 0.5
    want, want_lse = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices,
                                          is_causal=causal, sm_scale=sm, return_lse=True, **okw)
    kbd = kb.to(DEV)
    vbd = vb.to(DEV) if own_pool_v else kbd[..., :dv]
    ked = ke.to(DEV)
    ved = ve.to(DEV) if variant in ("own_v", "own_v_new_tokens", "max_jumps_own_v") else ked[..., :dv]
    lay = None
    if variant == "paged_pool":    # [pages, 1, page, 576] with a padded page stride: page / offset addressing
        big = torch.zeros(n_pages, ps * dk + 64, dtype=dtype, device=DEV)
        big[:, : ps * dk] = kbd.view(n_pages, ps * dk)
        kbd = big[:, : ps * dk].view(n_pages, 1, ps, dk)
        vbd = kbd[..., :dv]
        lay = ops.kv_layout_hnd(kbd, vbd)
    o = torch.zeros(T, hq, dv, dtype=dtype, device=DEV)
    lse = torch.zeros(T, hq, dtype=torch.float32, device=DEV)
    idx_dt = torch.int32 if variant == "many_requests_int32" else torch.int64
    ops.extend_attention_fwd(q.to(DEV), ked, ved, o, kbd, vbd, _t(qo), _t(kv_indptr), _t(kv_indices).to(idx_dt),
                             None, causal, None, int(ext.max()), 1.0, 1.0, lse_extend=lse, sm_scale=sm,
                             page_size=ps, kv_layout=lay, **okw)
    absw = orc.extend_attention(_np(q), _np(ke), parity.abs_values(_np(ve)), _np(kb), parity.abs_values(_np(vb)), qo,
                                kv_indptr, kv_indices, is_causal=causal, sm_scale=sm, **okw)
    live = np.repeat(np.arange(bs), ext)
    rows = np.ones(T, dtype=bool)
    if variant == "skip_prefix":
        pass
    if variant == "skip_extend":   # requests without a prefix see nothing: the reference leaves 0 / -inf-like rows
        rows = pre[live] > 0
    parity.check_out(_np(o.float())[rows], want[rows], dtype, ("mla_extend", variant), absw=absw[rows])
    np.testing.assert_allclose(_np(lse)[rows], want_lse[rows], atol=5e-3, rtol=2e-3)


# ---------------------------------------------------------------------------- K10 / K11
def test_move_kv_and_write_req_to_token(ops):
    rng = np.random.default_rng(0)
    bufs = [torch.randn(64, 256).to(torch.bfloat16).to(DEV) for _ in range(6)]
    ref = [_np(b).copy() for b in bufs]
    src = rng.permutation(63)[:10] + 1
    tgt = np.setdiff1d(np.arange(1, 64), src)[:10]
    orc.move_kv(ref, tgt, src)
    ptrs = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device=DEV)
    rb = torch.tensor([b.stride(0) * 2 for b in bufs], dtype=torch.int64, device=DEV)
    ops.move_kv(ptrs, rb, _t(tgt.astype(np.int64)), _t(src.astype(np.int64)))
    for b, r in zip(bufs, ref):
        assert np.array_equal(_np(b), r)
    # write_req_to_token
    r2t = torch.zeros(5, 64, dtype=torch.int32, device=DEV)
    pre = np.array([3, 0, 10], dtype=np.int64); seq = np.array([8, 4, 30], dtype=np.int64)
    pfx = [torch.arange(100, 103, device=DEV), None, torch.arange(200, 210, device=DEV)]
    ptr_t = torch.tensor([0 if t is None else t.data_ptr() for t in pfx], dtype=torch.int64, device=DEV)
    out_loc = torch.arange(1000, 1000 + int((seq - pre).sum()), device=DEV)
    ops.write_req_to_token(r2t, torch.tensor([4, 1, 2], device=DEV), ptr_t, _t(pre), _t(seq),
                           _t(seq - pre), out_loc)
    got = _np(r2t)
    assert got[4, :8].tolist() == [100, 101, 102, 1000, 1001, 1002, 1003, 1004]
    assert got[1, :4].tolist() == [1005, 1006, 1007, 1008]
    assert got[2, :30].tolist() == list(range(200, 210)) + list(range(1009, 1029))


# ---------------------------------------------------------------------------- MLA decode kernel
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,page_size", [(16, 1), (128, 16), (5, 64)])
def test_decode_mla_kernel(ops, dtype, hq, page_size):
    """rx::decode_mla_kernel (Dk 576 / Dv 512, V = first 512 columns of the latent rows): single pass
    and split-KV + stage 2, ragged lengths incl. tile-boundary cases, vs the fp64 oracle.
    Shapes of the reference's MLA test (test_triton_attention_kernels.py:674: (2,128,1,576,512))."""
    rng = np.random.default_rng(hq + page_size)
    lens = np.array([1, 31, 32, 33, 700, 64, 2049], dtype=np.int64)
    bs = len(lens)
    g = torch.Generator().manual_seed(hq)
    npages = int(sum((n + page_size - 1) // page_size for n in lens)) + 2
    pool = npages * page_size
    kv = torch.randn(pool, 1, 576, generator=g).to(dtype)
    q = torch.randn(bs, hq, 576, generator=g).to(dtype)
    pages = rng.permutation(np.arange(1, npages))
    r2t = np.zeros((bs + 1, 2100), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        k = (n + page_size - 1) // page_size
        sl = np.concatenate([np.arange(p * page_size, (p + 1) * page_size) for p in pages[pi: pi + k]]); pi += k
        r2t[i + 1, :n] = sl[:n]
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    sm = (128 + 64) ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    kvn = _np(kv)
    want = orc.decode_attention(_np(q), kvn, kvn[..., :512], kv_indptr, kv_indices, sm)
    absw = orc.decode_attention(_np(q), kvn, parity.abs_values(kvn[..., :512]), kv_indptr, kv_indices, sm)
    kvd, qd = kv.to(DEV), q.to(DEV)
    o = torch.zeros(bs, hq, 512, dtype=dtype, device=DEV)
    ops.decode_attention_fwd_paged(qd, kvd, kvd[..., :512], o, _t(r2t), _t(rpi), _t(lens), None, None, None, 1, sm,
                                   page_size=page_size)
    parity.check_out(_np(o.float()), want, dtype, "mla / single", absw=absw)   # the north star's element-wise bound
    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, _t(lens).int(), hq, 1, S, 256)
    al = torch.zeros(bs, hq, S, 512, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o2 = torch.zeros_like(o)
    ops.decode_attention_fwd(qd, kvd, kvd[..., :512], o2, _t(kv_indptr), _t(kv_indices), al, lse, nsplit, S, sm,
                             1.0, 1.0, page_size=page_size)
    parity.check_out(_np(o2.float()), want, dtype, "mla / split", absw=absw)
    # in-kernel stage 2 (merge_counters): same bits as the two-launch form, counters back at zero, call after call
    cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
    for rep in range(3):
        o3 = torch.full_like(o, float("nan"))
        ops.decode_attention_fwd(qd, kvd, kvd[..., :512], o3, _t(kv_indptr), _t(kv_indices), al, lse, nsplit, S, sm,
                                 1.0, 1.0, page_size=page_size, merge_counters=cnt)
        assert torch.equal(o3, o2), (rep, (o3.float() - o2.float()).abs().max().item())
        assert int(cnt.abs().sum()) == 0
    kv8 = kvd.to(torch.float8_e4m3fn)                     # the LDS-DMA kernel of fp8 latent rows
    o5, o6 = torch.zeros_like(o), torch.full_like(o, float("nan"))
    ops.decode_attention_fwd(qd, kv8, kv8[..., :512], o5, _t(kv_indptr), _t(kv_indices), al, lse, nsplit, S, sm,
                             1.0, 1.0, page_size=page_size)
    ops.decode_attention_fwd(qd, kv8, kv8[..., :512], o6, _t(kv_indptr), _t(kv_indices), al, lse, nsplit, S, sm,
                             1.0, 1.0, page_size=page_size, merge_counters=cnt)
    assert torch.equal(o5, o6) and int(cnt.abs().sum()) == 0


# ---------------------------------------------------------------------------- edge cases
def test_decode_edge_cases(ops):
    """bs=1 with a 40k-token context (forces real split-KV), an empty request inside a batch,
    page_size 64, int32 request indices / seq lens, q taken as a strided slice of a fused qkv buffer."""
    rng = np.random.default_rng(99)
    hq, hkv, d, ps = 32, 8, 128, 64
    lens = np.array([40000, 0, 65, 1], dtype=np.int64)
    bs = len(lens)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, bs, hq, hkv, d, lens, ps, torch.bfloat16)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want, absw = parity.want_and_absw(orc.decode_attention, (_np(q), _np(kb), _np(vb), kv_indptr, kv_indices, d ** -0.5), (2,))
    kbd, vbd = kb.to(DEV), vb.to(DEV)
    qkv = torch.zeros(bs, (hq + 2 * hkv) * d, dtype=torch.bfloat16, device=DEV)
    qkv[:, : hq * d] = q.view(bs, -1).to(DEV)
    q_view = qkv[:, : hq * d].view(bs, hq, d)  # row stride (hq+2hkv)*d: not contiguous
    assert not q_view.is_contiguous()
    S = 16
    lens32 = _t(lens).to(torch.int32)
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, torch.clamp(lens32, min=1), hq, hkv, S, 256)
    assert int(nsplit.max()) > 1
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    o = torch.full((bs, hq, d), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_fwd_paged(q_view, kbd, vbd, o, _t(r2t), _t(rpi).to(torch.int32), lens32, al, lse, nsplit,
                                   S, d ** -0.5, page_size=ps)
    got = _np(o.float()).astype(np.float64)
    live = lens > 0
    parity.check_out(got[live], want[live], torch.bfloat16, "decode edge cases / split", ulps=1, absw=absw[live])
    # single pass: the empty request gets a defined (zero) output
    o1 = torch.full((bs, hq, d), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_fwd_paged(q_view, kbd, vbd, o1, _t(r2t), _t(rpi), _t(lens), None, None, None, 1,
                                   d ** -0.5, page_size=ps)
    got1 = _np(o1.float()).astype(np.float64)
    parity.check_out(got1[live], want[live], torch.bfloat16, "decode edge cases / single pass", ulps=1, absw=absw[live])
    assert np.all(got1[~live] == 0)


def test_extend_edge_cases(ops):
    """one request longer than the 128-query workgroup tile with no prefix, one single-token extend on a
    long prefix, an empty extend inside the batch (qo_indptr repeats), fp16, int32 kv_indices."""
    rng = np.random.default_rng(5)
    hq, hkv, d = 8, 2, 128
    pre = np.array([0, 700, 33], dtype=np.int32)
    ext = np.array([300, 1, 0], dtype=np.int32)
    T, total = int(ext.sum()), int((pre + ext).sum())
    pool = total + 3
    slots = rng.permutation(pool - 1)[:total] + 1
    g = torch.Generator().manual_seed(2)
    kb = torch.randn(pool, hkv, d, generator=g).half()
    vb = torch.randn(pool, hkv, d, generator=g).half()
    q = torch.randn(T, hq, d, generator=g).half()
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.empty(int(pre.sum()), dtype=np.int64)
    ext_slots = np.empty(T, dtype=np.int64)
    so = 0
    for i in range(len(pre)):
        s = slots[so: so + pre[i] + ext[i]]; so += pre[i] + ext[i]
        kv_indices[kv_indptr[i]: kv_indptr[i + 1]] = s[: pre[i]]
        ext_slots[qo[i]: qo[i + 1]] = s[pre[i]:]
    ke, ve = kb[ext_slots], vb[ext_slots]
    want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices),
                                      (2, 4), sm_scale=d ** -0.5)
    o = torch.zeros(T, hq, d, dtype=torch.float16, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo).to(torch.int32),
                             _t(kv_indptr), _t(kv_indices).to(torch.int32), None, True, None, int(ext.max()),
                             1.0, 1.0)
    parity.check_out(_np(o.float()), want, torch.float16, "extend edge cases", ulps=1, absw=absw)


def test_k_and_v_scales(ops):
    """fp8-style per-tensor k/v scales multiply the logits / the prefix values (extend_attention.py:458,
    :508; decode_attention.py:1001-1003)."""
    rng = np.random.default_rng(8)
    hq, hkv, d = 4, 2, 64
    lens = np.array([50, 9], dtype=np.int64)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, 2, hq, hkv, d, lens, 1, torch.float16)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want, absw = parity.want_and_absw(orc.decode_attention, (_np(q), _np(kb), _np(vb), kv_indptr, kv_indices, 0.2), (2,),
                                      k_scale=0.5, v_scale=2.0)
    o = torch.zeros(2, hq, d, dtype=torch.float16, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o, _t(kv_indptr), _t(kv_indices), None, None, None,
                             1, 0.2, 0.5, 2.0)
    parity.check_out(_np(o.float()), want, torch.float16, "k / v scales, decode", ulps=1, absw=absw)
    # extend: scales apply to the cached prefix only
    pre, ext = np.array([20], dtype=np.int32), np.array([40], dtype=np.int32)
    g = torch.Generator().manual_seed(4)
    kbe = torch.randn(64, hkv, d, generator=g).half(); vbe = torch.randn(64, hkv, d, generator=g).half()
    qe = torch.randn(40, hq, d, generator=g).half()
    ke = torch.randn(40, hkv, d, generator=g).half(); ve = torch.randn(40, hkv, d, generator=g).half()
    kvi = np.arange(1, 21, dtype=np.int64)
    want_e, absw_e = parity.want_and_absw(orc.extend_attention, (_np(qe), _np(ke), _np(ve), _np(kbe), _np(vbe), np.array([0, 40]),
                                                                 np.array([0, 20], dtype=np.int32), kvi), (2, 4),
                                          sm_scale=0.2, k_scale=0.5, v_scale=2.0)
    oe = torch.zeros(40, hq, d, dtype=torch.float16, device=DEV)
    ops.extend_attention_fwd(qe.to(DEV), ke.to(DEV), ve.to(DEV), oe, kbe.to(DEV), vbe.to(DEV), _t(np.array([0, 40])),
                             _t(np.array([0, 20], dtype=np.int32)), _t(kvi), None, True, None, 40, 0.5, 2.0, sm_scale=0.2)
    parity.check_out(_np(oe.float()), want_e, torch.float16, "k / v scales, extend", ulps=1, absw=absw_e)


def test_torch_custom_ops_match_direct_calls_and_capture(ops, golden_dir):
    """torch.ops.radix_hip.* is the same code path as sglang_amd.ops, and replays under a HIP graph."""
    from sglang_amd import custom_ops  # noqa: F401  (registers the ops)

    cases = _cases(np.load(os.path.join(golden_dir, "decode.npz")))
    name, c = next(iter(cases.items()))
    o_ref, al_ref, lse_ref = _run_decode(ops, c, "split", None)
    q, kb, vb = _t(c["q"]), _t(c["kb"]), _t(c["vb"])
    S = int(c["max_splits"])
    o = torch.zeros_like(o_ref)
    al, lse = torch.zeros_like(al_ref), torch.zeros_like(lse_ref)
    a = (q, kb, vb, o, _t(c["kv_indptr"]), _t(c["kv_indices"]), al, lse, _t(c["nsplit"]), S,
         float(c["sm_scale"]))
    torch.ops.radix_hip.decode_attention(*a)
    assert torch.equal(o, o_ref)
    # HIP graph capture through the dispatcher
    o.zero_()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.ops.radix_hip.decode_attention(*a)
        s.synchronize()
        o.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            torch.ops.radix_hip.decode_attention(*a)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(o, o_ref)

    ecases = _cases(np.load(os.path.join(golden_dir, "extend.npz")))
    ename, e = next(iter(ecases.items()))
    oe_ref, _ = _run_extend(ops, e, with_lse=False)
    qe, ke, ve, kbe, vbe = (_t(e[k]) for k in ("q", "k_ext", "v_ext", "kb", "vb"))
    oe = torch.zeros_like(qe)
    torch.ops.radix_hip.extend_attention(qe, ke, ve, oe, kbe, vbe, _t(e["qo_indptr"]), _t(e["kv_indptr"]),
                                         _t(e["kv_indices"]), bool(e["causal"]),
                                         int(np.diff(e["qo_indptr"]).max()), 1.0, 1.0,
                                         float(e["sm_scale"]), float(e["logit_cap"]))
    assert torch.equal(oe, oe_ref)


def test_extend_tree_mask_window_xai_golden(ops, golden_dir):
    """F9 (reference Triton kernel, fp16): speculative tree masks with / without the prefix part,
    sliding window + window_kv_offsets, xai temperature.  D = 128 runs the MFMA kernel (fast prefix
    tiles, masked boundary path), D = 64 the generic kernel."""
    cases = _cases(np.load(os.path.join(golden_dir, "extend_mask.npz")))
    for name, c in cases.items():
        q, ke, ve, kb, vb = (_t(c[k]) for k in ("q", "k_ext", "v_ext", "kb", "vb"))
        o = torch.zeros_like(q)
        cm = _t(c["custom_mask"]) if "custom_mask" in c else None
        mi = _t(c["mask_indptr"]) if "mask_indptr" in c else None
        wo = _t(c["window_kv_offsets"]) if "window_kv_offsets" in c else None
        skipm = int(c["skip_prefix_mask"])
        ops.extend_attention_fwd(q, ke, ve, o, kb, vb, _t(c["qo_indptr"]), _t(c["kv_indptr"]), _t(c["kv_indices"]),
                                 cm, True, mi, int(np.diff(c["qo_indptr"]).max()), 1.0, 1.0,
                                 sm_scale=float(c["sm_scale"]), skip_prefix_custom_mask=(skipm != 0),
                                 sliding_window_size=int(c["window"]), window_kv_offsets=wo,
                                 xai_temperature_len=int(c["xai"]))
        got = _np(o).astype(np.float64)
        want = c["o"].astype(np.float64)
        ok = np.isfinite(want).all(axis=-1)  # rows that see nothing: 0/0 in the reference
        ref, absw = parity.want_and_absw(orc.extend_attention, (
            c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"]), (2, 4),
            is_causal=True, sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]),
            custom_mask=c.get("custom_mask"), mask_indptr=c.get("mask_indptr"),
            skip_prefix_custom_mask=(skipm != 0), window_kv_offsets=c.get("window_kv_offsets"),
            xai_temperature_len=int(c["xai"]))
        parity.check_out(got[ok], want[ok], torch.float16, (name, "vs triton golden"), ulps=2, absw=2 * absw[ok])  # (vs the reference kernel's own fp16 output: both sides round the result AND their P operand to 16 bits -- 2 ulp, twice the P term)
        parity.check_out(got[ok], ref[ok], torch.float16, (name, "vs oracle"), ulps=1, absw=absw[ok])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_extend_tree_mask_long_prefix_vs_oracle(ops, dtype):
    """TARGET_VERIFY shape: long cached prefixes (fast unmasked tiles) + a few draft tokens under a tree
    mask, bs 5, GQA, page 16."""
    rng = np.random.default_rng(23)
    hq, hkv, d, nd = 8, 2, 128, 7
    prefix = np.array([700, 64, 1, 333, 128], dtype=np.int64)
    bs = len(prefix)
    pool = int(prefix.sum()) + 40
    g = torch.Generator().manual_seed(9)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    T_ = bs * nd
    q = torch.randn(T_, hq, d, generator=g).to(dtype)
    ke = torch.randn(T_, hkv, d, generator=g).to(dtype)
    ve = torch.randn(T_, hkv, d, generator=g).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)
    kv_indices = (rng.permutation(pool - 1)[: int(prefix.sum())] + 1).astype(np.int64)
    qo = (np.arange(bs + 1) * nd).astype(np.int64)
    rows = []
    for i in range(bs):
        m = np.ones((nd, int(prefix[i]) + nd), dtype=bool)
        tri = np.tril(rng.random((nd, nd)) < 0.6)
        np.fill_diagonal(tri, True)
        m[:, int(prefix[i]):] = tri
        rows.append(m.reshape(-1))
    cm = np.concatenate(rows).astype(np.uint8)
    mi = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)
    sm = d ** -0.5
    want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices),
                                      (2, 4), is_causal=True, sm_scale=sm, custom_mask=cm, mask_indptr=mi)
    o = torch.zeros_like(q, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo), _t(kv_indptr),
                             _t(kv_indices), _t(cm).bool(), True, _t(mi), nd, 1.0, 1.0, sm_scale=sm)
    parity.check_out(_np(o.float()), want, dtype, "tree mask, long prefix", ulps=1, absw=absw)


def test_decode_xai_temperature_golden(ops, golden_dir):
    cases = _cases(np.load(os.path.join(golden_dir, "decode_xai.npz")))
    for name, c in cases.items():
        q, kb, vb = _t(c["q"]), _t(c["kb"]), _t(c["vb"])
        bs, hq, d = q.shape
        S = int(c["max_splits"])
        al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
        lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
        o = torch.zeros_like(q)
        ops.decode_attention_fwd(q, kb, vb, o, _t(c["kv_indptr"]), _t(c["kv_indices"]), al, lse, _t(c["nsplit"]), S,
                                 float(c["sm_scale"]), 1.0, 1.0, xai_temperature_len=int(c["xai"]))
        want, absw = parity.want_and_absw(orc.decode_attention, (c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"],
                                                                 float(c["sm_scale"])), (2,), xai_temperature_len=int(c["xai"]))
        parity.check_out(_np(o), c["o"], torch.float16, (name, "vs triton golden"), ulps=2, absw=2 * absw)  # (vs the reference kernel's own fp16 output: both sides round the result AND their P operand to 16 bits -- 2 ulp, twice the P term)
        parity.check_out(_np(o), want, torch.float16, (name, "vs oracle"), ulps=1, absw=absw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_merge_state_and_prefix_cascade(ops, dtype):
    """rx_merge_state vs the oracle, then the cascade it exists for: extend over (prefix only) and
    (new tokens only) merged by LSE equals the one-pass extend (merge_state.py docstring use)."""
    g = torch.Generator().manual_seed(2)
    T, H, D = 37, 6, 128
    a = torch.randn(T, H, D, generator=g).to(dtype).to(DEV)
    b = torch.randn(T, H, D, generator=g).to(dtype).to(DEV)
    la = torch.randn(T, H, generator=g).to(DEV) * 3
    lb = torch.randn(T, H, generator=g).to(DEV) * 3
    la[0, 0], lb[1, 1] = float("inf"), float("-inf")
    out, lse = ops.merge_state(a, la, b, lb)
    want, want_lse = orc.merge_state(_np(a), la.cpu().numpy(), _np(b), lb.cpu().numpy())
    fin = np.isfinite(want)  # (the +inf / -inf LSE rows: one side takes all)
    parity.check_out(orc.to_f64(_np(out))[fin], want[fin], dtype, "merge_state", ulps=1)  # one rounding of an exact blend
    np.testing.assert_allclose(lse.cpu().numpy(), want_lse, atol=1e-5, rtol=1e-5)

    # cascade
    rng = np.random.default_rng(1)
    hq, hkv, d = 8, 2, 128
    prefix = np.array([300, 17, 64], dtype=np.int64)
    ext = np.array([33, 64, 5], dtype=np.int64)
    pool = int(prefix.sum()) + 9
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype).to(DEV)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype).to(DEV)
    Tq = int(ext.sum())
    q = torch.randn(Tq, hq, d, generator=g).to(dtype).to(DEV)
    ke = torch.randn(Tq, hkv, d, generator=g).to(dtype).to(DEV)
    ve = torch.randn(Tq, hkv, d, generator=g).to(dtype).to(DEV)
    kv_indptr = _t(np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32))
    kv_indices = _t((rng.permutation(pool - 1)[: int(prefix.sum())] + 1).astype(np.int64))
    qo = _t(np.concatenate([[0], np.cumsum(ext)]).astype(np.int64))

    def run(**kw):
        o = torch.zeros_like(q)
        l = torch.zeros(Tq, hq, dtype=torch.float32, device=DEV)
        ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kv_indptr, kv_indices, None, True, None, int(ext.max()),
                                 1.0, 1.0, sm_scale=d ** -0.5, lse_extend=l, **kw)
        return o, l

    o_full, l_full = run()
    o_p, l_p = run(skip_extend=True)
    o_e, l_e = run(skip_prefix=True)
    o_m, l_m = ops.merge_state(o_p, l_p, o_e, l_e)
    assert (o_m.float() - o_full.float()).abs().max().item() <= (4e-3 if dtype == torch.float16 else 3e-2)
    np.testing.assert_allclose(l_m.cpu().numpy(), l_full.cpu().numpy(), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("pool", ["none", "nhd", "hnd_fp8"])
def test_rope_store_kv_golden(ops, golden_dir, pool):
    """Fused RoPE (+ KV store): q/k rotated in place match the reference's apply_rotary_emb to one
    16-bit rounding; the pool receives exactly the rotated k (and v), or their fp8 quantisation."""
    cases = _cases(np.load(os.path.join(golden_dir, "rope.npz")))
    for name, c in cases.items():
        for dtype in (torch.float16, torch.bfloat16):
            q = torch.from_numpy(c["q"]).to(dtype).to(DEV)
            k = torch.from_numpy(c["k"]).to(dtype).to(DEV)
            n, hkv, d = k.shape
            v = torch.randn(n, hkv, d, generator=torch.Generator().manual_seed(1)).to(dtype).to(DEV)
            pos = _t(c["positions"])
            cache = _t(c["cos_sin_cache"])
            neox, rot = bool(c["is_neox"]), int(c["rotary_dim"])
            want_q = orc.rope(_np(q) if dtype == torch.float16 else _np(q), c["positions"], c["cos_sin_cache"], neox, rot)
            want_k = orc.rope(_np(k), c["positions"], c["cos_sin_cache"], neox, rot)
            kw = {}
            page, slots = 16, 64
            loc = torch.tensor([3, 17, 18, 0, 40, 41, 63, 5, 9], device=DEV)
            if pool == "nhd":
                kb = torch.zeros(slots, hkv, d, dtype=dtype, device=DEV)
                vb = torch.zeros_like(kb)
                kw = dict(layout=ops._kv_layout(kb, vb, page), loc=loc, size_limit=slots)
            elif pool == "hnd_fp8":
                kb = torch.zeros(slots // page, hkv, page, d, dtype=torch.uint8, device=DEV)
                vb = torch.zeros_like(kb)
                kw = dict(layout=ops.kv_layout_hnd(kb, vb), loc=loc, size_limit=slots, k_scale=0.5, v_scale=2.0)
            ops.rope_store_kv(q, k, v if pool != "none" else None, pos, cache, neox, rotary_dim=rot, **kw)
            ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
            for got, want in ((q, want_q), (k, want_k)):
                g = orc.to_f64(_np(got))
                assert (np.abs(g - want) <= ulp * np.maximum(np.abs(want), 1.0) * 1.01).all(), (name, dtype)
            if pool == "none":
                continue
            keep = (loc != 0).cpu().numpy()
            locn = loc.cpu().numpy()[keep]
            if pool == "nhd":
                assert torch.equal(kb[loc[loc != 0]], k[torch.from_numpy(keep).to(DEV)])
                assert torch.equal(vb[loc[loc != 0]], v[torch.from_numpy(keep).to(DEV)])
                assert int(kb[0].abs().sum()) == 0  # reserved slot untouched
            else:
                gk = kb.cpu().numpy().transpose(0, 2, 1, 3).reshape(slots, hkv, d)[locn]
                gv = vb.cpu().numpy().transpose(0, 2, 1, 3).reshape(slots, hkv, d)[locn]
                is_bf = dtype == torch.bfloat16
                assert np.array_equal(gk, orc.quantize_kv_fp8(k.float().cpu().numpy()[keep], 0.5, is_bf))
                assert np.array_equal(gv, orc.quantize_kv_fp8(v.float().cpu().numpy()[keep], 2.0, is_bf))


def test_extend_unified_golden(ops, golden_dir):
    """K8 (deterministic one-stage extend): D = 128 on the MFMA kernel, D = 64 on the generic kernel."""
    cases = _cases(np.load(os.path.join(golden_dir, "extend_unified.npz")))
    for name, c in cases.items():
        q, kb, vb = _t(c["q"]), _t(c["kb"]), _t(c["vb"])
        o = torch.zeros_like(q)
        cm = _t(c["custom_mask"]) if "custom_mask" in c else None
        mi = _t(c["mask_indptr"]) if "mask_indptr" in c else None
        ops.extend_attention_fwd_unified(q, o, kb, vb, 1.0, 1.0, _t(c["qo_indptr"]), _t(c["kv_indptr"]),
                                         _t(c["kv_indices"]), _t(c["prefix_lens"]),
                                         int(np.diff(c["qo_indptr"]).max()), custom_mask=cm, mask_indptr=mi,
                                         sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]),
                                         xai_temperature_len=int(c["xai"]))
        got = _np(o).astype(np.float64)
        want = c["o"].astype(np.float64)
        ok = np.isfinite(want).all(axis=-1)
        ref, absw = parity.want_and_absw(orc.extend_attention_unified, (
            c["q"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"], c["prefix_lens"]), (2,),
            sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]), custom_mask=c.get("custom_mask"),
            mask_indptr=c.get("mask_indptr"), xai_temperature_len=int(c["xai"]))
        parity.check_out(got[ok], want[ok], torch.float16, (name, "vs triton golden"), ulps=2, absw=2 * absw[ok])  # (vs the reference kernel's own fp16 output: both sides round the result AND their P operand to 16 bits -- 2 ulp, twice the P term)
        parity.check_out(got[ok], ref[ok], torch.float16, (name, "vs oracle"), ulps=1, absw=absw[ok])


def test_extend_unified_equals_two_stage_on_long_batch(ops):
    """The unified form over (prefix + stored new tokens) equals the two-stage extend, incl. long prefixes
    that take the fast unmasked tiles and the causal boundary inside the kv list."""
    rng = np.random.default_rng(8)
    hq, hkv, d = 8, 2, 128
    prefix = np.array([300, 0, 129, 64], dtype=np.int64)
    ext = np.array([70, 260, 33, 1], dtype=np.int64)
    bs = len(ext)
    tot = prefix + ext
    pool = int(tot.sum()) + 9
    g = torch.Generator().manual_seed(4)
    kb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    vb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    T = int(ext.sum())
    q = torch.randn(T, hq, d, generator=g).to(torch.bfloat16).to(DEV)
    slots = rng.permutation(pool - 1)[: int(tot.sum())] + 1
    u_indptr = np.concatenate([[0], np.cumsum(tot)]).astype(np.int32)
    p_idx, e_idx = [], []
    for i in range(bs):
        s = slots[u_indptr[i]: u_indptr[i + 1]]
        p_idx.append(s[: prefix[i]]); e_idx.append(s[prefix[i]:])
    e_all = torch.from_numpy(np.concatenate(e_idx)).to(DEV)
    ke, ve = kb[e_all].contiguous(), vb[e_all].contiguous()
    qo = _t(np.concatenate([[0], np.cumsum(ext)]).astype(np.int64))
    o2 = torch.zeros_like(q)
    ops.extend_attention_fwd(q, ke, ve, o2, kb, vb, qo, _t(np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)),
                             _t(np.concatenate(p_idx).astype(np.int64)), None, True, None, int(ext.max()), 1.0, 1.0,
                             sm_scale=d ** -0.5)
    o1 = torch.zeros_like(q)
    ops.extend_attention_fwd_unified(q, o1, kb, vb, 1.0, 1.0, qo, _t(u_indptr), _t(slots.astype(np.int64)),
                                     _t(prefix.astype(np.int32)), int(ext.max()), sm_scale=d ** -0.5)
    assert (o1.float() - o2.float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("pattern", ["ramp", "spikes", "plateau"])
def test_extend_thresholded_max_adversarial_scores(ops, dtype, pattern):
    """The D = 128 MFMA kernel takes exp2 against the STANDING reference max of a row and redoes a 32-token block the
    classic way only when a lane's partial row sum exceeds 4096; the redo moves the reference max only when the block's
    max exceeds it by > 2^8 (kMaxSlack; rx_extend32_kernel.inc, sm_slice).  Score sequences built to stress that rule --
    a slow ramp that stays inside the slack tile after tile, rare huge spikes, a long flat plateau after one early
    peak -- must still match the fp64 oracle (output and LSE).  This is the unpacked four-wave instance; every other
    kernel that carries the sum check, and two more patterns, are in tests/test_gpu_adversarial_scores.py."""
    hq, hkv, d, P, E = 2, 1, 128, 1024 + 37, 192
    rng = np.random.default_rng({"ramp": 1, "spikes": 2, "plateau": 3}[pattern])
    g = torch.Generator().manual_seed(3)
    # q = e0 * sqrt(d) so that score(position n) = k[n][0]; the other coordinates add gaussian noise
    q = torch.randn(E, hq, d, generator=g) * 0.05
    q[:, :, 0] = d ** 0.5
    n = P + E
    if pattern == "ramp":        # +0.02 per token: ~1.8 log2 units per 64-token tile, never > 8 in one step
        base = torch.arange(n, dtype=torch.float32) * 0.02
    elif pattern == "spikes":    # flat, with a +40 spike every ~300 tokens, each higher than the last
        base = torch.zeros(n)
        for j, pos in enumerate(range(50, n, 300)):
            base[pos] = 40.0 + 7.0 * j
    else:                        # early peak at token 3, then a plateau 5.5 below it (inside the slack)
        base = torch.full((n,), 4.0)
        base[3] = 9.5
    kfull = torch.randn(n, hkv, d, generator=g) * 0.3
    kfull[:, :, 0] = base[:, None]
    vfull = torch.randn(n, hkv, d, generator=g)
    q, kfull, vfull = q.to(dtype), kfull.to(dtype), vfull.to(dtype)
    pool = n + 3
    slots = rng.permutation(pool - 1)[:n] + 1
    kb = torch.zeros(pool, hkv, d, dtype=dtype)
    vb = torch.zeros(pool, hkv, d, dtype=dtype)
    kb[slots] = kfull
    vb[slots] = vfull
    kv_indptr = np.array([0, P], dtype=np.int32)
    kv_indices = slots[:P].astype(np.int64)
    qo = np.array([0, E], dtype=np.int64)
    ke, ve = kfull[P:].contiguous(), vfull[P:].contiguous()
    want, want_lse = orc.extend_attention(_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices,
                                          sm_scale=1.0 / d ** 0.5, return_lse=True)
    absw = orc.extend_attention(_np(q), _np(ke), parity.abs_values(_np(ve)), _np(kb), parity.abs_values(_np(vb)), qo,
                                kv_indptr, kv_indices, sm_scale=1.0 / d ** 0.5)
    o = torch.zeros(E, hq, d, dtype=dtype, device=DEV)
    lse = torch.zeros(E, hq, dtype=torch.float32, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo), _t(kv_indptr),
                             _t(kv_indices), None, True, None, E, 1.0, 1.0, lse_extend=lse)
    parity.check_out(_np(o.float()), want, dtype, ("thresholded max", pattern), ulps=1, absw=absw)
    np.testing.assert_allclose(_np(lse), want_lse, atol=5e-3, rtol=2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,hkv", [(8, 2), (16, 2), (4, 1)])
@pytest.mark.parametrize("mode", ["tree", "causal", "window"])
def test_extend_gqa_packed_rows_match_oracle(ops, dtype, hq, hkv, mode):
    """q_pack (GQA-packed query rows, the speculative-verify shape: few new tokens per request over a long
    prefix): same results as the oracle for tree masks, plain causal extends and a sliding window."""
    d, page = 128, 16
    rng = np.random.default_rng(hq * 10 + hkv + len(mode))
    pre = np.array([300, 0, 1023, 64, 17], dtype=np.int32)
    ext = np.array([8, 5, 16, 1, 33], dtype=np.int32)
    bs, T = len(pre), int(ext.sum())
    total = int((pre + ext).sum())
    pool = total + 7
    slots = rng.permutation(pool - 1)[:total] + 1
    g = torch.Generator().manual_seed(hq + len(mode))
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(T, hq, d, generator=g).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.empty(int(pre.sum()), dtype=np.int64)
    ext_slots = np.empty(T, dtype=np.int64)
    so = 0
    for i in range(bs):
        s = slots[so: so + pre[i] + ext[i]]; so += pre[i] + ext[i]
        kv_indices[kv_indptr[i]: kv_indptr[i + 1]] = s[: pre[i]]
        ext_slots[qo[i]: qo[i + 1]] = s[pre[i]:]
    ke, ve = kb[ext_slots], vb[ext_slots]
    mask = mask_indptr = None
    kw = {}
    if mode == "tree":  # random draft tree: every new token sees the whole prefix, itself, and a random subset
        rows = []
        for i in range(bs):
            mm = np.ones((ext[i], pre[i] + ext[i]), dtype=np.uint8)
            tri = np.tril(rng.integers(0, 2, size=(ext[i], ext[i]))).astype(np.uint8)  # ancestors only
            np.fill_diagonal(tri, 1)
            mm[:, pre[i]:] = tri
            rows.append(mm.reshape(-1))
        mask = np.concatenate(rows)
        mask_indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    window = 40 if mode == "window" else -1
    want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices),
                                      (2, 4), sm_scale=1.0 / d ** 0.5, custom_mask=mask, mask_indptr=mask_indptr,
                                      sliding_window_size=window)
    o = torch.zeros(T, hq, d, dtype=dtype, device=DEV)
    ops.extend_attention_fwd_gqa_packed(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo),
                                        _t(kv_indptr), _t(kv_indices), None if mask is None else _t(mask), True,
                                        None if mask is None else _t(mask_indptr), int(ext.max()), 1.0, 1.0,
                                        sliding_window_size=window)
    got = _np(o.float()).astype(np.float64)
    ok = np.isfinite(want).all(axis=(1, 2))  # a window / mask can hide everything from a row (0/0 in the reference)
    parity.check_out(got[ok], want[ok], dtype, ("gqa packed rows", mode), ulps=1, absw=absw[ok])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("chunks", [1, 3, 8])
def test_verify_attention_splitkv_vs_oracle(ops, dtype, chunks):
    """Small-batch speculative verify, cached part split into chunks (the reference's verify_splitkv case):
    GQA-packed chunk launch + draft-block launch + rx_merge_chunks == the oracle's masked extend."""
    rng = np.random.default_rng(31 + chunks)
    hq, hkv, d, nd = 8, 2, 128, 6
    prefix = np.array([900, 64, 1, 333, 4097], dtype=np.int64)   # incl. fewer tokens than chunks * 64
    bs = len(prefix)
    pool = int(prefix.sum()) + 40
    g = torch.Generator().manual_seed(5)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    T_ = bs * nd
    q = torch.randn(T_, hq, d, generator=g).to(dtype)
    ke = torch.randn(T_, hkv, d, generator=g).to(dtype)
    ve = torch.randn(T_, hkv, d, generator=g).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)
    kv_indices = (rng.permutation(pool - 1)[: int(prefix.sum())] + 1).astype(np.int64)
    qo = (np.arange(bs + 1) * nd).astype(np.int64)
    rows = []
    for i in range(bs):
        m = np.ones((nd, int(prefix[i]) + nd), dtype=bool)
        tri = np.tril(rng.random((nd, nd)) < 0.6)
        np.fill_diagonal(tri, True)
        m[:, int(prefix[i]):] = tri
        rows.append(m.reshape(-1))
    cm = np.concatenate(rows).astype(np.uint8)
    mi = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)
    sm = d ** -0.5
    want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo, kv_indptr, kv_indices),
                                      (2, 4), is_causal=True, sm_scale=sm, custom_mask=cm, mask_indptr=mi)
    o = torch.zeros_like(q, device=DEV)
    ops.verify_attention_splitkv(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb.to(DEV), vb.to(DEV), _t(qo), _t(kv_indptr),
                                 _t(kv_indices), _t(cm), _t(mi), nd, chunks, 1.0, 1.0, sm_scale=sm)
    parity.check_out(_np(o.float()), want, dtype, ("verify split-KV", chunks), ulps=2, absw=absw)  # (16-bit chunk partials merged by LSE: two roundings, 2 ulp)


@pytest.mark.parametrize("masked", [False, True], ids=["causal", "tree_mask"])
@pytest.mark.parametrize("chunks", [1, 5])
def test_verify_attention_splitkv_latent_mla(ops, masked, chunks):
    """The split-KV form at the latent MLA shape (q 576 / v 512 over one kv head): the cached part cut into chunks
    that run as pseudo-requests of ONE rx::extend_mla_kernel launch, the new tokens' own block as a second launch
    (causal rule: the same kernel; a draft tree's mask: the generic kernel), rx_merge_chunks at Dv = 512 -- vs the
    fp64 oracle's masked extend."""
    rng = np.random.default_rng(77 + chunks)
    dtype = torch.bfloat16
    hq, dk, dv, nd = 16, 576, 512, 6
    prefix = np.array([700, 64, 1, 1300], dtype=np.int64)
    bs = len(prefix)
    pool = int(prefix.sum()) + 40
    g = torch.Generator().manual_seed(9)
    kb = (torch.randn(pool, 1, dk, generator=g) * 0.5).to(dtype)
    T_ = bs * nd
    q = torch.randn(T_, hq, dk, generator=g).to(dtype)
    ke = (torch.randn(T_, 1, dk, generator=g) * 0.5).to(dtype)
    kv_indptr = np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)
    kv_indices = (rng.permutation(pool - 1)[: int(prefix.sum())] + 1).astype(np.int64)
    qo = (np.arange(bs + 1) * nd).astype(np.int64)
    cm = mi = None
    if masked:
        rows = []
        for i in range(bs):
            m = np.ones((nd, int(prefix[i]) + nd), dtype=bool)
            tri = np.tril(rng.random((nd, nd)) < 0.6)
            np.fill_diagonal(tri, True)
            m[:, int(prefix[i]):] = tri
            rows.append(m.reshape(-1))
        cm = np.concatenate(rows).astype(np.uint8)
        mi = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)
    sm = 192 ** -0.5
    okw = dict(custom_mask=cm, mask_indptr=mi) if masked else {}
    want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ke[..., :dv].contiguous()), _np(kb),
                                                             _np(kb[..., :dv].contiguous()), qo, kv_indptr, kv_indices), (2, 4),
                                      is_causal=True, sm_scale=sm, **okw)
    o = torch.zeros(T_, hq, dv, dtype=dtype, device=DEV)
    kbd, ked = kb.to(DEV), ke.to(DEV)
    ops.verify_attention_splitkv(q.to(DEV), ked, ked[..., :dv], o, kbd, kbd[..., :dv], _t(qo), _t(kv_indptr),
                                 _t(kv_indices), _t(cm) if masked else None, _t(mi) if masked else None, nd, chunks,
                                 1.0, 1.0, sm_scale=sm)
    parity.check_out(_np(o.float()), want, dtype, ("latent split-KV", masked, chunks), ulps=2, absw=absw)  # (16-bit chunk partials merged by LSE: two roundings, 2 ulp)


def test_verify_splitkv_replays_under_hip_graph(ops):
    """plan() and the layer call use device data only: a captured {plan, call} graph follows changed cached lengths
    (different chunk boundaries) and a changed tree mask on replay."""
    hq, hkv, d, nd, bs = 8, 2, 128, 4, 2
    rng = np.random.default_rng(12)
    cap = 3000
    pool = bs * cap + 64
    g = torch.Generator().manual_seed(2)
    kb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    vb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    q = torch.randn(bs * nd, hq, d, generator=g).to(torch.bfloat16).to(DEV)
    ke = torch.randn(bs * nd, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    ve = torch.randn(bs * nd, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    kv_indices = torch.from_numpy((rng.permutation(pool - 1)[: bs * cap] + 1).astype(np.int64)).to(DEV)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    qo = torch.arange(0, (bs + 1) * nd, nd, dtype=torch.int64, device=DEV)
    mask = torch.ones(bs * nd * (cap + nd), dtype=torch.uint8, device=DEV)
    mi = torch.zeros(bs + 1, dtype=torch.int64, device=DEV)
    o = torch.zeros_like(q)
    vs = ops.VerifySplitKV(hq, hkv, torch.bfloat16, DEV)

    def set_case(prefix, seed):
        prefix = np.asarray(prefix, dtype=np.int64)
        kv_indptr.copy_(torch.from_numpy(np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)))
        r = np.random.default_rng(seed)
        rows = []
        for p in prefix:
            m = np.ones((nd, int(p) + nd), dtype=np.uint8)
            tri = np.tril(r.integers(0, 2, size=(nd, nd))) | np.eye(nd, dtype=np.int64)
            m[:, int(p):] = tri
            rows.append(m.reshape(-1))
        cm = np.concatenate(rows)
        mask[: cm.size] = torch.from_numpy(cm).to(DEV)
        mi.copy_(torch.from_numpy(np.concatenate([[0], np.cumsum([x.size for x in rows])]).astype(np.int64)))
        return prefix, cm

    def step():
        vs.plan(qo, kv_indptr, kv_indices, mask, mi, nd)
        vs(q, ke, ve, o, kb, vb, 1.0, 1.0)

    set_case([1000, 2500], 1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for prefix, seed in (([2999, 65], 7), ([700, 1500], 9)):
        prefix, cm = set_case(prefix, seed)
        graph.replay()
        torch.cuda.synchronize()
        ip = np.concatenate([[0], np.cumsum(prefix)]).astype(np.int32)
        want, absw = parity.want_and_absw(orc.extend_attention, (_np(q), _np(ke), _np(ve), _np(kb), _np(vb), qo.cpu().numpy(), ip,
                                                                 kv_indices.cpu().numpy()[: int(prefix.sum())]), (2, 4),
                                          is_causal=True, sm_scale=d ** -0.5, custom_mask=cm, mask_indptr=mi.cpu().numpy())
        parity.check_out(_np(o.float()), want, o.dtype, ("split-KV graph replay", seed), ulps=2, absw=absw)  # (16-bit chunk partials merged by LSE: two roundings, 2 ulp)


def test_decode_in_kernel_merge_edges_and_graph_replay(ops):
    """rx_decode_params.merge_counters: a zero-length request (no workgroup ever arrives), requests shorter than one
    split (one live split), sinks and a V scale in the merge, fp8 pool; then the call captured in a HIP graph and
    replayed with NEW lengths -- the counters are reset by the kernels themselves."""
    rng = np.random.default_rng(77)
    hq, hkv, d, ps, S = 16, 2, 128, 16, 8
    lens = np.array([0, 1, 40, 300, 1000, 17, 256], dtype=np.int64)
    bs = len(lens)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, bs, hq, hkv, d, np.maximum(lens, 1200), ps, torch.bfloat16, "shuffled")
    qd, kbd, vbd, r2td, rpid = q.to(DEV), kb.to(DEV), vb.to(DEV), _t(r2t), _t(rpi)
    sinks = torch.randn(hq, device=DEV)
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
    lens_d = _t(lens)
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)

    def both(kbuf, vbuf, lens_t, **kw):
        ops.get_num_kv_splits(nsplit, lens_t.to(torch.int32), hq, hkv, S, 256)
        a = torch.full((bs, hq, d), float("nan"), dtype=torch.bfloat16, device=DEV)
        b = torch.full_like(a, float("nan"))
        ops.decode_attention_fwd_paged(qd, kbuf, vbuf, a, r2td, rpid, lens_t, al, lse, nsplit, S, d ** -0.5, page_size=ps, **kw)
        ops.decode_attention_fwd_paged(qd, kbuf, vbuf, b, r2td, rpid, lens_t, al, lse, nsplit, S, d ** -0.5, page_size=ps,
                                       merge_counters=cnt, **kw)
        torch.cuda.synchronize()
        same = (a == b) | (torch.isnan(a) & torch.isnan(b))      # the empty request is 0/0 in both forms
        assert bool(same.all()) and int(cnt.abs().sum()) == 0
        return b

    both(kbd, vbd, lens_d)
    both(kbd, vbd, lens_d, sinks=sinks, v_scale=0.5, logit_cap=30.0)
    both(kbd.to(torch.float8_e4m3fn), vbd.to(torch.float8_e4m3fn), lens_d, k_scale=1.0, v_scale=1.0)
    # stale-line hunt: the SAME buffers, fresh queries and lengths every call, merged by whichever workgroup arrives
    # last (any XCD) -- a partial read from a stale cache line would differ from the two-launch result
    g = torch.Generator(device=DEV).manual_seed(9)
    for it in range(40):
        qd.copy_(torch.randn(qd.shape, device=DEV, generator=g).to(qd.dtype))
        lens_i = torch.randint(1, 1200, (bs,), device=DEV, generator=g)
        both(kbd, vbd, lens_i)
    # graph capture + replay with other lengths
    lens_g = lens_d.clone()
    ops.get_num_kv_splits(nsplit, lens_g.to(torch.int32), hq, hkv, S, 256)
    og = torch.zeros(bs, hq, d, dtype=torch.bfloat16, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.decode_attention_fwd_paged(qd, kbd, vbd, og, r2td, rpid, lens_g, al, lse, nsplit, S, d ** -0.5, page_size=ps,
                                       merge_counters=cnt)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ops.decode_attention_fwd_paged(qd, kbd, vbd, og, r2td, rpid, lens_g, al, lse, nsplit, S, d ** -0.5, page_size=ps,
                                       merge_counters=cnt)
    for new in ([5, 1200, 64, 31, 999, 2, 513], [1200] * bs, [33, 1, 1, 700, 8, 1100, 90]):
        new_t = torch.tensor(new, dtype=torch.int64, device=DEV)
        lens_g.copy_(new_t)
        ops.get_num_kv_splits(nsplit, lens_g.to(torch.int32), hq, hkv, S, 256)
        gr.replay()
        ref = torch.zeros_like(og)
        ops.decode_attention_fwd_paged(qd, kbd, vbd, ref, r2td, rpid, new_t, al, lse, nsplit, S, d ** -0.5, page_size=ps)
        torch.cuda.synchronize()
        assert torch.equal(og, ref) and int(cnt.abs().sum()) == 0, new


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (4, 1, 128), (16, 1, 128), (8, 8, 64), (12, 4, 64)])
@pytest.mark.parametrize("page_size,hnd", [(1, False), (16, False), (16, True), (64, True)])
def test_decode_fused_store_of_the_new_token(ops, dtype, hq, hkv, d, page_size, hnd):
    """rx_decode_params.k_new / v_new: the step's KV store inside the decode launch.  The pool holds every token BUT the
    newest of each request; the fused call must (a) give the bits of store-then-decode and (b) leave the new rows in
    their slots -- lengths put the new token on every position class of a 32-token tile (first / 16th / last row, a
    tile of its own), in the single-pass and the split form, both lookup modes."""
    rng = np.random.default_rng(hq * 7 + d + page_size)
    lens = np.array([1, 2, 16, 17, 32, 33, 48, 64, 65, 500, 1000], dtype=np.int64)
    bs = len(lens)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, bs, hq, hkv, d, lens, page_size, dtype, "shuffled")
    g = torch.Generator().manual_seed(3)
    k_new = torch.randn(bs, hkv, d, generator=g).to(dtype).to(DEV)
    v_new = torch.randn(bs, hkv, d, generator=g).to(dtype).to(DEV)
    new_slots = torch.tensor([int(r2t[i + 1, n - 1]) for i, n in enumerate(lens)], dtype=torch.int64, device=DEV)
    sm = d ** -0.5
    qd, r2td, rpid, lensd = q.to(DEV), _t(r2t), _t(rpi), _t(lens)

    def pools():  # fresh pools without the new tokens (their slots hold garbage)
        k0, v0 = kb.to(DEV).clone(), vb.to(DEV).clone()
        k0[new_slots] = 7.0
        v0[new_slots] = -7.0
        if not hnd:
            return k0, v0, None, (k0, v0)
        n_pages = k0.shape[0] // page_size
        kh = k0.view(n_pages, page_size, hkv, d).permute(0, 2, 1, 3).contiguous()
        vh = v0.view(n_pages, page_size, hkv, d).permute(0, 2, 1, 3).contiguous()
        return kh, vh, ops.kv_layout_hnd(kh, vh), (kh, vh)

    def rows(buf):  # the new tokens' rows of a pool, [bs, hkv, d]
        if not hnd:
            return buf[new_slots]
        return buf[new_slots // page_size, :, new_slots % page_size, :]

    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, lensd.to(torch.int32), hq, hkv, S, 256)
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    kvi = torch.empty(int(lens.sum()), dtype=torch.int64, device=DEV)
    kvp = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    ops.build_kv_indices(r2td, rpid, lensd, kvp, kvi)
    for mode in ("paged_single", "paged_split", "indices_split"):
        outs = []
        for fused in (False, True):
            kbuf, vbuf, lay, (kraw, vraw) = pools()
            if not fused:  # the store as its own step
                if hnd:
                    kraw[new_slots // page_size, :, new_slots % page_size, :] = k_new
                    vraw[new_slots // page_size, :, new_slots % page_size, :] = v_new
                else:
                    kraw[new_slots] = k_new
                    vraw[new_slots] = v_new
            kw = dict(k_new=k_new, v_new=v_new) if fused else {}
            o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=DEV)
            if mode == "paged_single":
                ops.decode_attention_fwd_paged(qd, kbuf, vbuf, o, r2td, rpid, lensd, None, None, None, 1, sm,
                                               page_size=page_size, kv_layout=lay, **kw)
            elif mode == "paged_split":
                ops.decode_attention_fwd_paged(qd, kbuf, vbuf, o, r2td, rpid, lensd, al, lse, nsplit, S, sm,
                                               page_size=page_size, kv_layout=lay, **kw)
            else:
                ops.decode_attention_fwd(qd, kbuf, vbuf, o, kvp, kvi, al, lse, nsplit, S, sm, 1.0, 1.0,
                                         page_size=page_size, kv_layout=lay, **kw)
            torch.cuda.synchronize()
            outs.append(o)
            if fused:
                assert torch.equal(rows(kraw), k_new) and torch.equal(rows(vraw), v_new), mode
        assert torch.equal(outs[0], outs[1]), (mode, (outs[0].float() - outs[1].float()).abs().max().item())


def test_decode_fused_store_rejects_what_it_cannot_do(ops):
    from sglang_amd.lib import RadixHipError

    bs, d = 2, 128
    q = torch.randn(bs, 32, d, device=DEV).to(torch.bfloat16)          # 32 q heads on ONE kv head: two q blocks
    kb = torch.randn(64, 1, d, device=DEV).to(torch.bfloat16)
    r2t = torch.arange(64, dtype=torch.int32, device=DEV).repeat(bs + 1, 1)
    lens = torch.tensor([5, 9], device=DEV)
    o = torch.empty_like(q)
    kn = torch.randn(bs, 1, d, device=DEV).to(torch.bfloat16)
    with pytest.raises(RadixHipError):
        ops.decode_attention_fwd_paged(q, kb, kb, o, r2t, torch.tensor([1, 2], device=DEV), lens, None, None, None, 1,
                                       d ** -0.5, k_new=kn, v_new=kn)
    with pytest.raises(ValueError):
        ops.decode_attention_fwd_paged(q[:, :4], kb, kb, o[:, :4], r2t, torch.tensor([1, 2], device=DEV), lens, None, None,
                                       None, 1, d ** -0.5, k_new=kn, v_new=None)


def test_decode_request_order_changes_nothing_but_the_launch_order(ops):
    """rx_decode_params.request_order (longest request first): same bits as the natural order, single pass and split."""
    rng = np.random.default_rng(12)
    hq, hkv, d, ps = 8, 2, 128, 16
    lens = rng.integers(1, 900, size=40).astype(np.int64)
    bs = len(lens)
    q, kb, vb, r2t, rpi = _make_paged_case(rng, bs, hq, hkv, d, lens, ps, torch.bfloat16, "shuffled")
    qd, kbd, vbd, r2td, rpid, lensd = q.to(DEV), kb.to(DEV), vb.to(DEV), _t(r2t), _t(rpi), _t(lens)
    order = torch.argsort(lensd, descending=True).to(torch.int32)
    assert sorted(order.tolist()) == list(range(bs))
    S = 8
    nsplit = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(nsplit, lensd.to(torch.int32), hq, hkv, S, 64)
    al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
    for split in (False, True):
        outs = []
        for ro in (None, order):
            o = torch.full((bs, hq, d), float("nan"), dtype=torch.bfloat16, device=DEV)
            if split:
                ops.decode_attention_fwd_paged(qd, kbd, vbd, o, r2td, rpid, lensd, al, lse, nsplit, S, d ** -0.5,
                                               page_size=ps, request_order=ro)
            else:
                ops.decode_attention_fwd_paged(qd, kbd, vbd, o, r2td, rpid, lensd, None, None, None, 1, d ** -0.5,
                                               page_size=ps, request_order=ro)
            outs.append(o)
        assert torch.equal(outs[0], outs[1])
