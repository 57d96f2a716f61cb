"""Pin the CPU oracle (oracle/radix_oracle.py) against the golden vectors produced
by the reference's own kernels/allocators (tests/golden/make_golden.py).
Integer paths must match bit-exactly; attention within the reference's own
tolerances (test_triton_attention_kernels.py:559 atol=rtol=1e-2 for decode,
:307 rtol=1e-2/atol=1e-3 for extend) -- the fp64 oracle vs the fp16 Triton output.
"""
import json
import os

import numpy as np
import pytest

from oracle import radix_oracle as orc


def _load_npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _cases(npz):
    out = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        out.setdefault(case, {})[field] = npz[key]
    return out


# ------------------------------------------------------------------ F1
def _replay_allocator(case, alloc):
    for step, ent in enumerate(case["log"]):
        op = ent["op"]
        if op == "alloc":
            out = alloc.alloc(ent["need"])
        elif op == "alloc_extend":
            out = alloc.alloc_extend(ent["prefix_lens"], ent["seq_lens"], ent["last_loc"])
        elif op == "alloc_decode":
            out = alloc.alloc_decode(ent["seq_lens"], ent["last_loc"])
        elif op == "free":
            alloc.free(np.array(ent["idx"], dtype=np.int64)); out = "skip"
        elif op == "free_segment":
            alloc.free_segment(np.array(ent["idx"], dtype=np.int64), ent["start_pos"]); out = "skip"
        elif op == "merge_and_sort_free":
            alloc.merge_and_sort_free(); out = "skip"
        elif op == "free_group":
            alloc.free_group_begin()
            for idx in ent["idx"]:
                alloc.free(np.array(idx, dtype=np.int64))
            alloc.free_group_end(); out = "skip"
        else:
            raise AssertionError(op)
        if not isinstance(out, str):
            want = ent["out"]
            if want is None:
                assert out is None, (step, op)
            else:
                assert out is not None and out.tolist() == want, (step, op)
        free, rel = ent["free"]
        assert alloc.free_pages.tolist() == free, (step, op)
        assert alloc.release_pages.tolist() == rel, (step, op)


def test_allocator_sequences_bit_exact(golden_dir):
    with open(os.path.join(golden_dir, "alloc_sequences.json")) as f:
        cases = json.load(f)
    assert len(cases) == 8
    for case in cases:
        ps = case["page_size"]
        if ps == 1:
            alloc = orc.TokenAllocatorOracle(case["size"], case["need_sort"])
        else:
            alloc = orc.PagedAllocatorOracle(case["size"], ps, case["need_sort"])
        _replay_allocator(case, alloc)


# ------------------------------------------------------------------ F2
def test_kv_indices_bit_exact(golden_dir):
    z = _load_npz(golden_dir, "kv_indices.npz")
    for ci in range(3):
        for use_start in (0, 1):
            t = f"c{ci}_{use_start}_"
            kv_indptr, kv_indices = orc.build_kv_indices(
                z[t + "req_to_token"], z[t + "req_pool_indices"], z[t + "lens"],
                z[t + "start"] if use_start else None)
            assert np.array_equal(kv_indptr, z[t + "kv_indptr"])
            assert kv_indptr.dtype == np.int32
            assert np.array_equal(kv_indices, z[t + "kv_indices"])


# ------------------------------------------------------------------ F3
def test_num_kv_splits_bit_exact(golden_dir):
    with open(os.path.join(golden_dir, "kv_splits.json")) as f:
        rows = json.load(f)
    assert len(rows) >= 400
    for r in rows:
        got = orc.num_kv_splits(r["seq_lens"], r["num_group"], r["hq"], r["hkv"],
                                r["max_splits"], r["cores"])
        assert got.tolist() == r["out"], {k: r[k] for k in ("hq", "hkv", "max_splits", "num_group")}


# ------------------------------------------------------------------ F4
def test_store_kv_bit_exact(golden_dir):
    z = _load_npz(golden_dir, "store_kv.npz")
    for ci in range(2):
        kc, vc = z[f"c{ci}_kc_in"].copy(), z[f"c{ci}_vc_in"].copy()
        orc.store_kv(z[f"c{ci}_k"], z[f"c{ci}_v"], kc, vc, z[f"c{ci}_loc"])
        assert np.array_equal(kc, z[f"c{ci}_kc_out"])
        assert np.array_equal(vc, z[f"c{ci}_vc_out"])


def test_store_kv_skips_reserved_slot_and_rejects_oob():
    k = np.arange(12, dtype=np.uint16).reshape(3, 4)
    kc = np.zeros((5, 4), dtype=np.uint16)
    vc = np.zeros((5, 4), dtype=np.uint16)
    orc.store_kv(k, k + 100, kc, vc, np.array([2, 0, 4]))
    assert np.array_equal(kc[2], k[0]) and np.array_equal(kc[4], k[2])
    assert not kc[0].any() and not vc[0].any()  # slot 0 is the padding sink: skipped
    with pytest.raises(IndexError):
        orc.store_kv(k, k, kc, vc, np.array([1, 5, 2]))


# ------------------------------------------------------------------ F5
def test_decode_attention_vs_triton_golden(golden_dir):
    cases = _cases(_load_npz(golden_dir, "decode.npz"))
    assert len(cases) >= 11
    for name, c in cases.items():
        cap = float(c["logit_cap"]) if "logit_cap" in c else 0.0
        o, lse = orc.decode_attention(c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"],
                                      float(c["sm_scale"]), logit_cap=cap, return_lse=True)
        want = c["o"].astype(np.float64)
        np.testing.assert_allclose(o, want, atol=1e-2, rtol=1e-2, err_msg=name)
        # split layout: which (b,h,s) entries exist and what they hold
        logits, lse_s, o2 = orc.decode_attention_split(
            c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"], c["nsplit"],
            int(c["max_splits"]), float(c["sm_scale"]), logit_cap=cap)
        np.testing.assert_allclose(o2, o, atol=1e-9, rtol=1e-9, err_msg=name)
        written = ~np.isnan(lse_s)
        np.testing.assert_allclose(lse_s[written], c["attn_lse"].astype(np.float64)[written],
                                   atol=2e-3, rtol=2e-3, err_msg=name)
        np.testing.assert_allclose(logits[written], c["attn_logits"].astype(np.float64)[written],
                                   atol=5e-3, rtol=1e-2, err_msg=name)


# ------------------------------------------------------------------ F6
def test_extend_attention_vs_triton_golden(golden_dir):
    cases = _cases(_load_npz(golden_dir, "extend.npz"))
    assert len(cases) == 5
    for name, c in cases.items():
        o, lse = orc.extend_attention(
            c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"],
            c["kv_indices"], is_causal=bool(c["causal"]), sm_scale=float(c["sm_scale"]),
            logit_cap=float(c["logit_cap"]), return_lse=True)
        np.testing.assert_allclose(o, c["o"].astype(np.float64), atol=2e-3, rtol=1e-2, err_msg=name)
        np.testing.assert_allclose(lse, c["lse"].astype(np.float64), atol=2e-3, rtol=1e-3,
                                   err_msg=name)


def test_sdpa_semantics_match_kernel_semantics():
    """torch_native (req_to_token gather + causal SDPA over the whole sequence) and the
    Triton contract (prefix from cache + contiguous new K/V) describe the same function."""
    rng = np.random.default_rng(0)
    hq, hkv, d = 4, 2, 16
    pre = np.array([3, 0]); ext = np.array([4, 5]); seq = pre + ext
    pool = 32
    kb = rng.standard_normal((pool, hkv, d)).astype(np.float32)
    vb = rng.standard_normal((pool, hkv, d)).astype(np.float32)
    r2t = np.zeros((3, 16), dtype=np.int32)
    slots = rng.permutation(pool - 1)[: seq.sum()] + 1
    r2t[1, : seq[0]] = slots[: seq[0]]; r2t[2, : seq[1]] = slots[seq[0]:]
    rpi = np.array([1, 2])
    q = rng.standard_normal((ext.sum(), hq, d)).astype(np.float32)
    a = orc.sdpa_extend_req_to_token(q, kb, vb, r2t, rpi, seq, pre, ext, 0.25)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, pre)
    qo = np.concatenate([[0], np.cumsum(ext)])
    k_ext = np.concatenate([kb[r2t[1, pre[0]:seq[0]]], kb[r2t[2, pre[1]:seq[1]]]])
    v_ext = np.concatenate([vb[r2t[1, pre[0]:seq[0]]], vb[r2t[2, pre[1]:seq[1]]]])
    b = orc.extend_attention(q, k_ext, v_ext, kb, vb, qo, kv_indptr, kv_indices, sm_scale=0.25)
    np.testing.assert_allclose(a, b, atol=1e-12)


def test_bf16_roundtrip():
    x = np.array([1.0, -2.5, 3.14159, 1e-3, 65504.0, np.nan], dtype=np.float32)
    b = orc.f32_to_bf16(x)
    y = orc.bf16_to_f32(b)
    assert np.isnan(y[-1])
    np.testing.assert_allclose(y[:-1], x[:-1], rtol=2**-8)
    assert orc.f32_to_bf16(np.array([1.00390625], dtype=np.float32))[0] == 0x3F80  # ties-to-even


# ------------------------------------------------------------------ F8 (bf16, reference C++ CPU kernels)
def test_cpu_native_golden_bf16(golden_dir):
    """decode_attention_cpu / extend_attention_cpu outputs (compiled from the reference's own
    aot/csrc/cpu sources) vs the numpy oracle AND the C oracle; tolerance of the reference's CPU
    tests (test_decode.py:266 atol 3e-2, test_extend.py:344 atol=rtol=1e-2) -- observed ~4e-3,
    one bf16 ulp of the output."""
    from oracle import c_oracle

    cases = _cases(_load_npz(golden_dir, "cpu_native.npz"))
    assert len(cases) == 5
    for name, c in cases.items():
        sm = float(c["sm_scale"])
        if name.startswith("dec"):
            # the kernel stores the new token's K/V at loc before attending (decode.cpp:940)
            kb, vb = c["kb_in"].copy(), c["vb_in"].copy()
            orc.store_kv(c["new_k"], c["new_v"], kb, vb, c["loc"], reserved_skip_index=-1)
            assert np.array_equal(kb, c["kb_out"]) and np.array_equal(vb, c["vb_out"])
            kv_indptr, kv_indices = orc.build_kv_indices(c["req_to_token"], c["req_pool_indices"],
                                                         c["seq_lens"])
            o = orc.decode_attention(c["q"], kb, vb, kv_indptr, kv_indices, sm)
            got_c = orc.bf16_to_f32(c_oracle.decode_bf16(c["q"], kb, vb, c["req_to_token"],
                                                         c["req_pool_indices"], c["seq_lens"], sm))
        else:
            pre, ext = c["extend_prefix_lens"], c["extend_seq_lens"]
            kv_indptr, kv_indices = orc.build_kv_indices(c["req_to_token"], c["req_pool_indices"], pre)
            qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
            o = orc.extend_attention(c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], qo, kv_indptr,
                                     kv_indices, sm_scale=sm)
            got_c = orc.bf16_to_f32(c_oracle.extend_bf16(c["q"], c["k_ext"], c["v_ext"], c["kb"],
                                                         c["vb"], qo, kv_indptr, kv_indices, sm))
            # and the torch_native formulation (whole sequence from the cache) agrees
            o2 = orc.sdpa_extend_req_to_token(c["q"], c["kb"], c["vb"], c["req_to_token"],
                                              c["req_pool_indices"], c["seq_lens"], pre, ext, sm)
            np.testing.assert_allclose(o2, o, atol=1e-9)
        want = orc.bf16_to_f32(c["o"]).astype(np.float64)
        np.testing.assert_allclose(o, want, atol=1e-2, rtol=1e-2, err_msg=name)
        np.testing.assert_allclose(got_c, o, atol=1e-2, rtol=1e-2, err_msg=name + " (C oracle)")


def test_fp8_e4m3fn_codec_matches_torch_cast():
    """The reference's fp8 KV pools are torch casts (memory_pool.py:2334-2343, :4022-4066): pin the
    oracle's e4m3fn decode / encode (RNE, ties to even, NaN above 464) against torch's CPU cast."""
    import torch

    u = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(u).view(torch.float8_e4m3fn).float().numpy()
    got = orc.fp8_e4m3fn_decode(u)
    assert np.array_equal(np.isnan(want), np.isnan(got))
    assert np.array_equal(np.nan_to_num(want), np.nan_to_num(got))
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * np.exp(rng.uniform(-12, 7, 100000))).astype(np.float32)
    mid = ((orc._FP8_POS[:-1] + orc._FP8_POS[1:]) / 2).astype(np.float32)
    x = np.concatenate([x, mid, -mid, np.array([0.0, -0.0, 448, 464, 464.5, 1e6, -1e6], np.float32)])
    for cast in (torch.bfloat16, torch.float16):
        xs = torch.from_numpy(x).to(cast)
        want = xs.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
        got = orc.fp8_e4m3fn_encode(xs.float().numpy())
        assert np.array_equal(want, got), cast
    # quant-on-write with a scale: in-place div_ on the 16-bit tensor, then the cast
    xb = torch.from_numpy(x[:50000]).bfloat16()
    want = xb.clone().div_(0.37).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = orc.quantize_kv_fp8(xb.float().numpy(), 0.37, True)
    assert np.array_equal(want, got)


def test_extend_mask_window_xai_golden(golden_dir):
    """F9: tree (custom) masks with / without the prefix part, sliding window + window_kv_offsets and
    the xai temperature, produced by the reference's Triton extend kernel (fp16, interpreter)."""
    z = np.load(os.path.join(golden_dir, "extend_mask.npz"))
    cases = {}
    for key in z.files:
        c, f = key.split(".", 1)
        cases.setdefault(c, {})[f] = z[key]
    assert set(cases) == {"tree", "tree_prefix", "tree_swa", "xai", "swa"}
    for name, c in cases.items():
        skipm = int(c["skip_prefix_mask"])
        got = orc.extend_attention(
            c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"],
            is_causal=True, sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]),
            custom_mask=c.get("custom_mask"), mask_indptr=c.get("mask_indptr"),
            skip_prefix_custom_mask=(skipm != 0), window_kv_offsets=c.get("window_kv_offsets"),
            xai_temperature_len=int(c["xai"]))
        want = c["o"].astype(np.float64)
        ok = np.isfinite(want).all(axis=-1)  # rows that see nothing are 0/0 in the reference
        assert ok.mean() > 0.8, name
        assert np.abs(got[ok] - want[ok]).max() <= 2e-3, (name, np.abs(got[ok] - want[ok]).max())


def test_decode_xai_temperature_golden(golden_dir):
    """F10: Grok's xai temperature in the reference's decode kernels (grouped + MHA, fp16)."""
    z = np.load(os.path.join(golden_dir, "decode_xai.npz"))
    cases = {}
    for key in z.files:
        c, f = key.split(".", 1)
        cases.setdefault(c, {})[f] = z[key]
    for name, c in cases.items():
        got = orc.decode_attention(c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"], float(c["sm_scale"]),
                                   xai_temperature_len=int(c["xai"]))
        assert np.abs(got - c["o"].astype(np.float64)).max() <= 2e-3, name
        off = orc.decode_attention(c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"], float(c["sm_scale"]))
        assert np.abs(off - c["o"].astype(np.float64)).max() > 1e-2, "the case must exercise the factor"


def test_rope_golden(golden_dir):
    """F11: the oracle's rope() vs the reference's torch-native apply_rotary_emb (fp32)."""
    z = np.load(os.path.join(golden_dir, "rope.npz"))
    cases = {}
    for key in z.files:
        c, f = key.split(".", 1)
        cases.setdefault(c, {})[f] = z[key]
    assert set(cases) == {"neox128", "gptj64", "partial"}
    for name, c in cases.items():
        for x, want in (("q", "q_out"), ("k", "k_out")):
            got = orc.rope(c[x], c["positions"], c["cos_sin_cache"], bool(c["is_neox"]), int(c["rotary_dim"]))
            assert np.abs(got - c[want].astype(np.float64)).max() <= 2e-6, (name, x)


def _npz_cases(path):
    z = np.load(path)
    cases = {}
    for key in z.files:
        c, f = key.split(".", 1)
        cases.setdefault(c, {})[f] = z[key]
    return cases


def test_extend_unified_golden(golden_dir):
    """F12: the one-stage unified extend of deterministic inference (reference Triton kernel, fp16)."""
    cases = _npz_cases(os.path.join(golden_dir, "extend_unified.npz"))
    assert set(cases) == {"causal", "mha64", "swa", "tree", "xai"}
    for name, c in cases.items():
        got = orc.extend_attention_unified(
            c["q"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"], c["prefix_lens"],
            sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]), custom_mask=c.get("custom_mask"),
            mask_indptr=c.get("mask_indptr"), xai_temperature_len=int(c["xai"]))
        want = c["o"].astype(np.float64)
        ok = np.isfinite(want).all(axis=-1)
        assert ok.mean() > 0.8, name
        assert np.abs(got[ok] - want[ok]).max() <= 2e-3, (name, np.abs(got[ok] - want[ok]).max())


def test_score_bias_golden(golden_dir):
    """F19: score_mod = relative_bias_score_mod with aux_tensors = [rel_logits] (score_mod.py:44-56) through the reference's
    extend (both stages; sliding window + logit cap), unified-extend and decode Triton kernels (fp16, interpreter)."""
    cases = _npz_cases(os.path.join(golden_dir, "score_bias.npz"))
    assert set(cases) == {"ext_gqa", "ext_long", "ext_mha64", "ext_swa_cap", "uni_gqa", "uni_mha64", "dec_gqa", "dec_mha64", "dec_wide"}
    for name, c in cases.items():
        aux = c["aux"].astype(np.float64)
        assert (c["aux"].dtype == np.float32) == bool(c["aux_f32"])
        if name.startswith("ext_"):
            args = (c["q"], c["k_ext"], c["v_ext"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"])
            kw = dict(is_causal=True, sm_scale=float(c["sm_scale"]), sliding_window_size=int(c["window"]), logit_cap=float(c["cap"]))
            fn = orc.extend_attention
        elif name.startswith("uni_"):
            args = (c["q"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"], c["prefix_lens"])
            kw = dict(sm_scale=float(c["sm_scale"]))
            fn = orc.extend_attention_unified
        else:
            args = (c["q"], c["kb"], c["vb"], c["kv_indptr"], c["kv_indices"], float(c["sm_scale"]))
            kw = {}
            fn = orc.decode_attention
        got = fn(*args, score_bias=aux, **kw)
        want = c["o"].astype(np.float64)
        ok = np.isfinite(want).all(axis=-1)
        assert ok.mean() > 0.8, name
        assert np.abs(got[ok] - want[ok]).max() <= 2e-3, (name, np.abs(got[ok] - want[ok]).max())
        off = fn(*args, **kw)
        assert np.abs(off[ok] - want[ok]).max() > 1e-2, (name, "the case must exercise the bias")


def test_unified_kv_indices_golden(golden_dir):
    """F20: build_unified_kv_indices (the reference's Triton copy kernel + torch cumsum), bit-exact."""
    cases = _npz_cases(os.path.join(golden_dir, "unified_kv_indices.npz"))
    assert set(cases) == {"ragged", "no_prefix", "one", "many"}
    for name, c in cases.items():
        bs = len(c["prefix_lens"])
        indptr, idx, pl = orc.build_unified_kv_indices(c["prefix_kv_indptr"], c["prefix_kv_indices"], c["extend_start_loc"],
                                                       c["extend_seq_lens"], c["extend_kv_indices"], bs)
        assert np.array_equal(indptr, c["unified_kv_indptr"]) and indptr.dtype == np.int32, name
        assert np.array_equal(idx, c["unified_kv_indices"]) and idx.dtype == np.int64, name
        assert np.array_equal(pl, c["prefix_lens"]), name


def test_row_vectorised_extend_oracle_equals_its_row_at_a_time_form():
    """The extend oracles work on all query rows of a (request, head) at once (round 5: the GPU suite's wall time was
    mostly their per-row Python loops); the row-at-a-time forms they were pinned in stay in the module and must agree to
    fp64 summation order on every feature: causal / not, window, cap + temperature, tree masks (prefix masked or not),
    relative bias + sinks, skip_prefix / skip_extend, scales -- and on which rows see nothing."""
    rng = np.random.default_rng(1)
    hq, hkv, d = 8, 2, 32
    pre, ext = [40, 0, 7], [6, 20, 9]
    T, pool = sum(ext), sum(pre) + sum(ext) + 9
    f16 = lambda *sh: rng.standard_normal(sh).astype(np.float16)  # noqa: E731
    q, ke, ve, kb, vb = f16(T, hq, d), f16(T, hkv, d), f16(T, hkv, d), f16(pool, hkv, d), f16(pool, hkv, d)
    kvp = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    perm = rng.permutation(pool - 1) + 1
    kvi = perm[: sum(pre)]
    qo = np.concatenate([[0], np.cumsum(ext)])
    masks, mi = [], [0]
    for p_, e_ in zip(pre, ext):
        m = rng.random((e_, p_ + e_)) < 0.6
        for r in range(e_):
            m[r, p_ + r] = True
            m[r, p_ + r + 1:] = False
        m[0, :p_] = False  # (with the prefix part masked, row 0 sees only itself)
        masks.append(m.reshape(-1)); mi.append(mi[-1] + m.size)
    cm, mi = np.concatenate(masks), np.array(mi)
    aux = rng.standard_normal((T, hq, 11))
    variants = [dict(), dict(is_causal=False), dict(sliding_window_size=5), dict(logit_cap=20.0, xai_temperature_len=8),
                dict(custom_mask=cm, mask_indptr=mi, skip_prefix_custom_mask=False), dict(custom_mask=cm, mask_indptr=mi),
                dict(score_bias=aux, sinks=rng.standard_normal(hq)), dict(skip_prefix=True), dict(skip_extend=True, is_causal=False),
                dict(k_scale=0.7, v_scale=1.3, score_bias=aux, sliding_window_size=9)]
    for kw in variants:
        a = orc.extend_attention(q, ke, ve, kb, vb, qo, kvp, kvi, return_lse=True, return_absw=True, **kw)
        b = orc._extend_attention_rows(q, ke, ve, kb, vb, qo, kvp, kvi, return_lse=True, return_absw=True, **kw)
        for x, y in zip(a, b):
            fin = np.isfinite(y)
            assert np.array_equal(np.isfinite(x), fin), kw
            assert np.abs(x[fin] - y[fin]).max() <= 1e-12, kw
    # unified form: the kv list holds prefix + new tokens
    tot = np.array(pre) + np.array(ext)
    ukvp = np.concatenate([[0], np.cumsum(tot)]).astype(np.int32)
    ukvi = perm[: int(tot.sum())]
    for kw in [dict(), dict(sliding_window_size=5), dict(logit_cap=20.0, xai_temperature_len=8), dict(custom_mask=cm, mask_indptr=mi),
               dict(score_bias=aux, sinks=rng.standard_normal(hq), k_scale=0.7, v_scale=1.3), dict(is_causal=False)]:
        a = orc.extend_attention_unified(q, kb, vb, qo, ukvp, ukvi, np.array(pre), return_absw=True, **kw)
        b = orc._extend_attention_unified_rows(q, kb, vb, qo, ukvp, ukvi, np.array(pre), return_absw=True, **kw)
        for x, y in zip(a, b):
            assert np.abs(x - y).max() <= 1e-12, kw


def test_cpu_baseline_container_fixture(golden_dir):
    """SURVEY 8(d) CPU-baseline item (1): the reference's compiled CPU kernel and the C restatement timed in
    the build container on identical inputs (oracle/time_cpu_container.py) -- fixture present and consistent."""
    import json
    import os

    d = json.load(open(os.path.join(golden_dir, "cpu_baseline_container.json")))
    assert d["reference"]["kind"] == "reference" and d["port"]["kind"] == "port"
    for k in ("reference", "port"):
        assert d[k]["ms_per_layer"] > 0 and abs(d[k]["value"] - 256 / (32 * d[k]["ms_per_layer"] * 1e-3)) < 1e-6
    assert d["parity_max_abs_reference_vs_port_small_case"] <= 2e-3  # bf16 outputs of two summation orders
    # round 3: the extend half of the metric (extend_attention_cpu, extend.cpp:425) beside it
    for k in ("extend_reference", "extend_port"):
        e = d[k]
        assert e["unit"] == "TFLOP/s" and e["kind"] == k.split("_")[1]
        assert abs(e["value"] - e["flops_per_chunk"] / (e["ms_per_chunk"] * 1e-3) / 1e12) < 1e-6
    assert d["extend_parity_max_abs_reference_vs_port_small_case"] <= 8e-3  # one bf16 ulp at |o| in [1, 2)


def test_a14_torch_native_semantics_pinned_to_the_reference(golden_dir):
    """a14, pinned DIRECTLY: outputs of the reference's own TorchNativeAttnBackend._run_sdpa_forward_extend /
    _run_sdpa_forward_decode (torch_native_backend.py:61-277; executed from the reference file by
    tests/golden/make_golden.py::f14 on fp32 tensors) vs the oracle's restatement -- ragged extend over cached
    prefixes, decode, GQA / MHA / MQA, and both under a sliding window (_make_sliding_window_mask :36-48)."""
    npz = np.load(os.path.join(golden_dir, "torch_native.npz"))
    cases = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        cases.setdefault(case, {})[field] = npz[key]
    assert set(cases) == {"gqa", "mha", "mqa_window", "gqa_window"}
    for name, c in cases.items():
        d = c["q"].shape[-1]
        w = int(c["window"])
        got = orc.sdpa_extend_req_to_token(c["q"], c["kc"], c["vc"], c["r2t"], c["rpi"], c["seq"], c["prefix"], c["ext"],
                                           d ** -0.5, causal=True, sliding_window_size=w)
        assert np.abs(got - c["o_extend"].astype(np.float64)).max() < 5e-6, name
        gotd = orc.sdpa_decode_req_to_token(c["qd"], c["kc"], c["vc"], c["r2t"], c["rpi"], c["seq"], d ** -0.5,
                                            sliding_window_size=w)
        assert np.abs(gotd - c["o_decode"].astype(np.float64)).max() < 5e-6, name


def _dcp_golden(golden_dir):
    npz = np.load(os.path.join(golden_dir, "dcp.npz"))
    cases = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        cases.setdefault(case, {})[field] = npz[key]
    return cases


def test_dcp_lens_and_local_kv_indices_bit_exact(golden_dir):
    """8e (decode context parallel), F15: the reference's get_dcp_lens (srt/layers/dcp/layout.py:23-41) and its
    create_triton_kv_indices_for_dcp_triton (kernels/ops/attention/dcp_kernels.py:34-76, run under the Triton
    interpreter) for dcp 2 / 3 / 8, every rank, with and without a start offset, lengths 0 / 1 / dcp-1 / dcp / dcp+1."""
    cases = _dcp_golden(golden_dir)
    n = int(cases["idx"]["count"])
    assert n == 26
    for i in range(n):
        c = cases[f"idx{i}"]
        start = c["start"] if int(c["use_start"]) else None
        dl = orc.dcp_lens(c["lens"], int(c["dcp"]), int(c["rank"]), start)
        assert np.array_equal(dl, c["dcp_lens"]), i
        indptr, idx, dl2 = orc.dcp_kv_indices(c["req_to_token"], c["req_pool_indices"], c["lens"], int(c["dcp"]),
                                              int(c["rank"]), start)
        assert np.array_equal(indptr, c["kv_indptr"]) and np.array_equal(idx, c["kv_indices"]), i
        assert np.array_equal(dl2, c["dcp_lens"])
    # the shares of all ranks partition the request
    c = cases["idx0"]
    tot = sum(orc.dcp_lens(c["lens"], 3, r) for r in range(3))
    assert np.array_equal(tot, c["lens"])


def test_dcp_lse_merge_matches_the_reference(golden_dir):
    """F15: cp_lse_ag_out_rs_mha (srt/layers/dcp/comm.py:82-108) executed from the reference file rank by rank:
    the scaled contribution of every rank, the head slice of the sum each rank keeps and the global LSE -- with a
    rank that saw no token for one request (LSE -inf, NaN output row) and a row that is empty on every rank."""
    cases = _dcp_golden(golden_dir)
    for mi in range(int(cases["merge"]["count"])):
        c = cases[f"merge{mi}"]
        world, T, H, D = c["outs"].shape
        scaled, summed, g = orc.dcp_merge(c["outs"], c["lses"])
        assert np.abs(scaled - c["scaled"]).max() < 2e-6
        hl = H // world
        for r in range(world):
            assert np.abs(summed[:, r * hl:(r + 1) * hl] - c["final"][r]).max() < 5e-6
            want_l = c["global_lse"][r].astype(np.float64)
            got_l = g[:, r * hl:(r + 1) * hl]
            both_inf = np.isneginf(want_l) & np.isneginf(got_l)
            with np.errstate(invalid="ignore"):
                assert np.all(both_inf | (np.abs(got_l - want_l) < 5e-6))


def test_dcp_store_loc_restatement():
    loc = np.array([10, 11, 12, 13, 25, 7], dtype=np.int64)
    pos = np.array([0, 1, 2, 3, 9, 4], dtype=np.int64)
    assert orc.dcp_store_loc(loc, pos, 2, 1).tolist() == [0, 5, 0, 6, 12, 0]
    assert orc.dcp_store_loc(loc, pos, 2, 0, skip_index=-1).tolist() == [5, -1, 6, -1, -1, 3]


def test_dcp_host_lens_match_the_reference(golden_dir):
    """sglang_amd.attention.dcp.get_dcp_lens (the host-side form planning code uses) against F15's get_dcp_lens rows."""
    import torch

    from sglang_amd.attention.dcp import get_dcp_lens

    cases = _dcp_golden(golden_dir)
    for i in range(int(cases["idx"]["count"])):
        c = cases[f"idx{i}"]
        start = torch.from_numpy(c["start"]) if int(c["use_start"]) else None
        got = get_dcp_lens(torch.from_numpy(c["lens"]), int(c["dcp"]), int(c["rank"]), start)
        assert np.array_equal(got.numpy(), c["dcp_lens"]), i


# ------------------------------------------------------------------ F16
def test_mla_decode_fused_rope_oracle_matches_reference_kernel(golden_dir):
    """oracle.decode_attention_grouped_rope against the reference's own stage-1 kernel run under the Triton interpreter
    (rocm_mla_decode_rope.py:45-315; partials merged by LSE in make_golden.f16): the attention output within the
    reference's own decode tolerance, the rotated k_pe of the newest tokens to one fp16 rounding."""
    for name, c in _cases(_load_npz(golden_dir, "mla_rope.npz")).items():
        o, kpe = orc.decode_attention_grouped_rope(c["q"], c["kb"], c["kv_indptr"], c["kv_indices"], c["cos_sin"],
                                                   c["positions"], float(c["sm_scale"]), is_neox=bool(c["neox"]))
        np.testing.assert_allclose(o, c["o"].astype(np.float64), atol=2e-3, rtol=2e-3, err_msg=name)
        want_k = c["k_pe_out"].astype(np.float64).reshape(kpe.shape)
        assert np.abs(kpe - want_k).max() <= 2.0 ** -10 * max(1.0, np.abs(want_k).max()), name


# ------------------------------------------------------------------ F17
def test_fused_qk_norm_rope_oracle_matches_reference_norm_and_rope(golden_dir):
    """oracle.fused_qk_norm_rope against RMSNorm.forward_native + apply_rotary_emb run in fp32 (make_golden.f17: the pair the
    reference's own test holds its fused kernel to, kernels/aot/tests/test_fused_qk_norm_rope.py:31-128): with the
    frequencies computed from `base` (the fused kernel's way, fused_qknorm_rope.cuh:42-63) and with the golden's cos / sin
    rows as a cache, both to fp32 rounding of values up to |w| * sqrt(D) ~ 100; YaRN off (factor 1) reduces to the same."""
    cases = _npz_cases(os.path.join(golden_dir, "qknorm_rope.npz"))
    assert set(cases) == {"neox128", "gptj128", "partial64of128", "gptj64", "neox256", "partial_gptj32of64"}
    for name, c in cases.items():
        hq, hkv, d = int(c["hq"]), int(c["hkv"]), int(c["head_dim"])
        n = c["qkv"].shape[0]
        q = c["qkv"][:, : hq * d].reshape(n, hq, d)
        k = c["qkv"][:, hq * d: (hq + hkv) * d].reshape(n, hkv, d)
        want_q, want_k = c["q_out"].astype(np.float64), c["k_out"].astype(np.float64)
        scale = max(np.abs(want_q).max(), np.abs(want_k).max())
        for kw in (dict(), dict(cos_sin_cache=c["cos_sin"], positions=np.arange(n))):
            args = dict(positions=c["positions"], eps=float(c["eps"]), base=float(c["base"]), is_neox=bool(c["is_neox"]),
                        rotary_dim=int(c["rotary_dim"]))
            args.update(kw)
            gq, gk = orc.fused_qk_norm_rope(q, k, c["q_weight"], c["k_weight"], **args)
            # fp32 pipeline vs float64: the angle pos * freq carries fp32 rounding of both factors (pos <= 4095)
            tol = scale * (3e-4 if not kw else 4e-6)
            assert np.abs(gq - want_q).max() <= tol and np.abs(gk - want_k).max() <= tol, (name, bool(kw), np.abs(gq - want_q).max(), tol)
    # YaRN: the ramp's two ends (fused_qknorm_rope.cuh:46-60): below `low` pure extrapolation, above `high` freq / factor
    f = orc.qknorm_rope_freqs(64, 10000.0, factor=4.0, low=8.0, high=24.0)
    f0 = orc.qknorm_rope_freqs(64, 10000.0)
    assert np.allclose(f[:9], f0[:9]) and np.allclose(f[24:], f0[24:] / 4.0) and (f[9:24] < f0[9:24]).all() and (f[9:24] > f0[9:24] / 4).all()
    g = orc.qknorm_rope_freqs(64, 10000.0, factor=2.0, low=5.0, high=5.0)      # low == high: high + 0.001
    assert np.allclose(g[:6], f0[:6]) and np.allclose(g[6:], f0[6:] / 2.0)


def test_draft_decode_kv_indices_golden(golden_dir):
    """F18: the oracle's restatement of generate_draft_decode_kv_indices against the reference's Triton kernel (run under
    the interpreter by tests/golden/make_golden.py f18), every element of both outputs, bit-exact."""
    z = np.load(os.path.join(golden_dir, "draft_kv_indices.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) == 6
    for n in names:
        g = {k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(n + ".")}
        kvi, kvp = orc.draft_decode_kv_indices(g["req_to_token"], g["req_pool_indices"], g["seq_lens"], g["positions"],
                                               int(g["topk"]), int(g["num_steps"]), int(g["page_size"]),
                                               g["kv_indices"].shape[1], g["kv_indptr"].shape[1])
        assert np.array_equal(kvi, g["kv_indices"]), n
        assert np.array_equal(kvp, g["kv_indptr"]), n


def test_fused_fp8_qkv_quantisation_golden(golden_dir):
    """F21: the oracle's restatement of fused_fp8_qkv_kv_cache's arithmetic (x * (1.0f / scale), satfinite RNE to e4m3fn)
    against the reference test's expected bytes -- every case of the fixture, K / V with their scales, q with scale 1, and
    the edge values (+-448, saturating values, ties, subnormals, -0.0); bit-exact."""
    z = np.load(os.path.join(golden_dir, "fused_fp8_qkv.npz"))
    n_cases = int(z["n_cases"][0])
    assert n_cases >= 60
    for c in range(n_cases):
        hq, hkv, hd, n, slots, has_scale, is_bf16 = (int(x) for x in z[f"c{c}.meta"])
        qkv = z[f"c{c}.qkv"]
        x = orc.bf16_to_f32(qkv) if is_bf16 else qkv.astype(np.float32)
        q_dim, kv_dim = hq * hd, hkv * hd
        ks, vs = (float(s) for s in z[f"c{c}.scale"])
        assert np.array_equal(orc.quantize_fused_fp8(x[:, :q_dim]), z[f"c{c}.q_fp8"]), c
        assert np.array_equal(orc.quantize_fused_fp8(x[:, q_dim: q_dim + kv_dim], ks), z[f"c{c}.k_fp8"]), c
        assert np.array_equal(orc.quantize_fused_fp8(x[:, q_dim + kv_dim:], vs), z[f"c{c}.v_fp8"]), c
    for dn in ("bf16", "fp16"):
        for sc in (1.0, 0.5, 3.0, 0.3):
            xe = z[f"edge_{dn}_{sc}.x"]
            xf = orc.bf16_to_f32(xe) if dn == "bf16" else xe.astype(np.float32)
            assert np.array_equal(orc.quantize_fused_fp8(xf, sc), z[f"edge_{dn}_{sc}.fp8"]), (dn, sc)
