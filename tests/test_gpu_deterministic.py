"""Deterministic-inference mode of HipRadixAttnBackend (round 5): server_args.enable_deterministic_inference as in
TritonAttnBackend (srt/layers/attention/triton_backend.py:247-264 fixed split tile, :325-333 ceil(len / tile) kv splits,
:1339-1350 + :1572-1712 the one-stage extend over the unified kv list built by build_unified_kv_indices,
kernels/ops/attention/extend_attention.py:193-238).

* rx_build_unified_kv_indices is bit-exact against the reference's own output (tests/golden/unified_kv_indices.npz, F20).
* The mode's point, as PROPERTIES held to the last bit: a request's decode output does not depend on the batch it rides
  in; an extend row's output does not depend on where the radix cache cut the prompt (prefix / extend split) nor on the
  other requests of the batch."""
import os

import numpy as np
import pytest

import parity_util as parity
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_build_unified_kv_indices_golden(golden_dir):
    from sglang_amd import ops

    z = np.load(os.path.join(golden_dir, "unified_kv_indices.npz"))
    cases = {}
    for key in z.files:
        c, f = key.split(".", 1)
        cases.setdefault(c, {})[f] = z[key]
    for name, c in cases.items():
        bs = len(c["prefix_lens"])
        for idt in (torch.int64, torch.int32):  # (the pool's slot lists come in either width)
            pre = _t(c["prefix_kv_indices"]).to(idt) if c["prefix_kv_indices"].size else None
            indptr, idx, pl = ops.build_unified_kv_indices(_t(c["prefix_kv_indptr"]), pre, _t(c["extend_start_loc"]),
                                                           _t(c["extend_seq_lens"]), _t(c["extend_kv_indices"]).to(idt), bs)
            total = int(c["unified_kv_indptr"][-1])
            assert indptr.dtype == torch.int32 and idx.dtype == torch.int64 and pl.dtype == torch.int32
            assert np.array_equal(indptr.cpu().numpy(), c["unified_kv_indptr"]), name
            assert np.array_equal(idx.cpu().numpy()[:total], c["unified_kv_indices"]), name
            assert np.array_equal(pl.cpu().numpy(), c["prefix_lens"]), name
    # empty batch
    indptr, idx, pl = ops.build_unified_kv_indices(torch.zeros(1, dtype=torch.int32, device=DEV), None,
                                                   torch.zeros(0, dtype=torch.int32, device=DEV),
                                                   torch.zeros(0, dtype=torch.int32, device=DEV),
                                                   torch.zeros(0, dtype=torch.int64, device=DEV), 0)
    assert indptr.tolist() == [0] and pl.numel() == 0


def _harness(dtype, d=128, hq=8, hkv=2, index_mode="paged"):
    from test_gpu_backend import _Harness

    return _Harness(16, hq, hkv, d, dtype, "shuffled_pages", index_mode, max_ctx=4200, size=16384,
                    server_args_extra={"enable_deterministic_inference": True})


def _extend(hs, rows, prefix_lens, extend_lens, q, k, v):
    from sglang_amd.forward_batch import ForwardBatch

    seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
    loc = hs.alloc_extend(rows, list(prefix_lens), seq_lens)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    fb = ForwardBatch.for_extend(rpi, torch.tensor(seq_lens, device=DEV), loc, list(prefix_lens), list(extend_lens))
    hs.backend.init_forward_metadata(fb)
    return hs.layer(q, k, v, fb, hs.backend)


@pytest.mark.parametrize("d", [128, 64], ids=["d128_mfma", "d64_generic"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_extend_rows_do_not_depend_on_the_prefix_split_or_the_batch(dtype, d):
    """One 700-token prompt: (a) prefilled whole, alone; (b) its first 300 tokens cached, the other 400 extended in a
    batch with two unrelated requests; (c) cut at 333 (not a tile or page multiple).  Rows 300.. / 333.. of (a) must be
    the rows of (b) / (c) to the last bit, and (a) matches the oracle at the bar."""
    from sglang_amd import lib as rxlib

    hq, hkv, L = 8, 2, 700
    g = torch.Generator().manual_seed(3)
    q = torch.randn(L, hq * d, generator=g).to(dtype).to(DEV)
    k = torch.randn(L, hkv * d, generator=g).to(dtype).to(DEV)
    v = torch.randn(L, hkv * d, generator=g).to(dtype).to(DEV)
    hs = _harness(dtype, d)
    assert hs.backend.enable_deterministic and hs.backend.split_tile_size == 256
    rows = hs.r2t.alloc(1)
    o_whole = _extend(hs, rows, [0], [L], q, k, v)
    assert rxlib.last_dispatch().startswith("extend_mfma32_uni_kernel" if d == 128 else "extend_generic_kernel"), rxlib.last_dispatch()
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = hs.r2t.req_to_token.cpu().numpy()
    idx = r2t[rows[0], :L].astype(np.int64)
    bits = (lambda t: t.detach().cpu().contiguous().view(torch.uint16).numpy()) if dtype == torch.bfloat16 else (lambda t: t.detach().cpu().numpy())
    want, absw = parity.want_and_absw(orc.extend_attention_unified, (bits(q.view(L, hq, d)), bits(kb), bits(vb), np.array([0, L]),
                                                                   np.array([0, L], dtype=np.int32), idx, np.array([0])), (2,), sm_scale=d ** -0.5)
    parity.check_out(o_whole.view(L, hq, d).float().cpu().numpy(), want, dtype, ("deterministic extend", d), absw=absw)
    for cut in (300, 333):
        hs2 = _harness(dtype, d)
        rows2 = hs2.r2t.alloc(3)
        # the cached part of OUR request holds the same K / V values (other slots); two strangers share the batch
        loc = hs2.alloc_extend([rows2[1]], [0], [cut])
        hs2.pool.set_kv_buffer(hs2.layer, loc, k[:cut].view(cut, hkv, d), v[:cut].view(cut, hkv, d))
        hs2.fill_prefix([rows2[0], rows2[2]], [77, 400])
        ext = [50, L - cut, 129]
        q2 = torch.cat([hs2.rand(ext[0], hq * d), q[cut:], hs2.rand(ext[2], hq * d)])
        k2 = torch.cat([hs2.rand(ext[0], hkv * d), k[cut:], hs2.rand(ext[2], hkv * d)])
        v2 = torch.cat([hs2.rand(ext[0], hkv * d), v[cut:], hs2.rand(ext[2], hkv * d)])
        o2 = _extend(hs2, rows2, [77, cut, 400], ext, q2, k2, v2)
        mine = o2[ext[0]: ext[0] + L - cut]
        assert torch.equal(mine.view(torch.int16), o_whole[cut:].view(torch.int16)), (cut, (mine.float() - o_whole[cut:].float()).abs().max().item())


@pytest.mark.parametrize("index_mode", ["paged", "indices"])
def test_decode_rows_do_not_depend_on_the_batch(index_mode):
    """ceil(len / 256) kv splits per request whatever the batch: a 700-token and a 3000-token request decode to the same
    bits alone and among five others; the schedule's counts are the reference's formula; the result meets the oracle."""
    from sglang_amd.forward_batch import ForwardBatch

    dtype, hq, hkv, d = torch.bfloat16, 8, 2, 128
    g = torch.Generator().manual_seed(5)
    lens_all = [700, 3000, 31, 257, 1024, 4000, 2]
    kv = {n: (torch.randn(n, hkv, d, generator=g).to(dtype).to(DEV), torch.randn(n, hkv, d, generator=g).to(dtype).to(DEV)) for n in lens_all}
    qs = {n: torch.randn(1, hq * d, generator=g).to(dtype).to(DEV) for n in lens_all}

    def run(lens):
        hs = _harness(dtype, d, hq, hkv, index_mode)
        rows = hs.r2t.alloc(len(lens))
        for r, n in zip(rows, lens):
            loc = hs.alloc_extend([r], [0], [n])
            hs.pool.set_kv_buffer(hs.layer, loc, kv[n][0], kv[n][1])
        rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
        seq_t = torch.tensor(lens, dtype=torch.int64)
        fb = ForwardBatch.for_decode(rpi, seq_t.to(DEV), torch.zeros(len(lens), dtype=torch.int64, device=DEV), seq_t)
        hs.backend.init_forward_metadata(fb)
        md = hs.backend.forward_metadata
        assert md.num_kv_splits.tolist() == [(n + 255) // 256 for n in lens] and md.max_kv_splits == (4200 + 255) // 256
        o = hs.layer(torch.cat([qs[n] for n in lens]), None, None, fb, hs.backend, save_kv_cache=False)
        return o, hs, rows

    o_all, hs, rows = run(lens_all)
    for i, n in enumerate(lens_all[:2]):
        o_one, _, _ = run([n])
        assert torch.equal(o_one[0].view(torch.int16), o_all[i].view(torch.int16)), n
    o_perm, _, _ = run(lens_all[::-1])
    assert torch.equal(o_perm.flip(0).view(torch.int16), o_all.view(torch.int16))
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = hs.r2t.req_to_token.cpu().numpy()
    kvi = np.concatenate([r2t[r, :n] for r, n in zip(rows, lens_all)]).astype(np.int64)
    kvp = np.concatenate([[0], np.cumsum(lens_all)]).astype(np.int32)
    bits = lambda t: t.detach().cpu().contiguous().view(torch.uint16).numpy()  # noqa: E731
    qa = torch.cat([qs[n] for n in lens_all]).view(len(lens_all), hq, d)
    want, absw = parity.want_and_absw(orc.decode_attention, (bits(qa), bits(kb), bits(vb), kvp, kvi, d ** -0.5), (2,))
    parity.check_out(o_all.view(len(lens_all), hq, d).float().cpu().numpy(), want, dtype, ("deterministic decode", index_mode), absw=absw)


def test_static_kv_splits_env(monkeypatch):
    """SGLANG_TRITON_DECODE_ATTN_STATIC_KV_SPLITS (triton_backend.py:215-217, :321-325): every request gets the cap."""
    from test_gpu_backend import _Harness

    from sglang_amd.forward_batch import ForwardBatch

    monkeypatch.setenv("SGLANG_TRITON_DECODE_ATTN_STATIC_KV_SPLITS", "true")
    hs = _Harness(16, 8, 2, 128, torch.float16, "contiguous", "indices")
    assert hs.backend.static_kv_splits and hs.backend.split_policy == "reference"
    lens = [40, 700, 3]
    rows = hs.r2t.alloc(3)
    hs.fill_prefix(rows, lens)
    seq_t = torch.tensor(lens, dtype=torch.int64)
    fb = ForwardBatch.for_decode(torch.tensor(rows, dtype=torch.int64, device=DEV), seq_t.to(DEV),
                                 torch.zeros(3, dtype=torch.int64, device=DEV), seq_t)
    hs.backend.init_forward_metadata(fb)
    md = hs.backend.forward_metadata
    assert md.num_kv_splits.tolist() == [8, 8, 8] and md.max_kv_splits == 8
    q = hs.rand(3, 8 * 128)
    o = hs.layer(q, None, None, fb, hs.backend, save_kv_cache=False)
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = hs.r2t.req_to_token.cpu().numpy()
    kvi = np.concatenate([r2t[r, :n] for r, n in zip(rows, lens)]).astype(np.int64)
    kvp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    want, absw = parity.want_and_absw(orc.decode_attention, (q.view(3, 8, 128).cpu().numpy(), kb.cpu().numpy(), vb.cpu().numpy(),
                                                             kvp, kvi, 128 ** -0.5), (2,))
    parity.check_out(o.view(3, 8, 128).float().cpu().numpy(), want, torch.float16, "static kv splits", absw=absw)


def test_target_verify_under_deterministic_mode():
    """TARGET_VERIFY through _forward_extend_unified (triton_backend.py:1632-1647: extend lens from spec_info.draft_token_num,
    start locs by cumsum; the tree mask rows are kv_len wide in the unified form): the draft rows against the oracle's
    two-stage semantics with the same mask."""
    from sglang_amd.forward_batch import ForwardBatch, ForwardMode

    hq, hkv, d, nd = 8, 2, 128, 5
    hs = _harness(torch.float16, d, hq, hkv)
    seq_lens = [300, 17, 129]
    bs = len(seq_lens)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, seq_lens)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    total = [s + nd for s in seq_lens]
    loc = hs.alloc_extend(rows, list(seq_lens), total)
    rng = np.random.default_rng(4)
    masks = []
    for s in seq_lens:
        m = np.ones((nd, s + nd), dtype=bool)
        tri = np.tril(rng.random((nd, nd)) < 0.5)
        np.fill_diagonal(tri, True)
        m[:, s:] = tri
        masks.append(m.reshape(-1))
    cm = np.concatenate(masks)

    class Spec:
        draft_token_num = nd
        custom_mask = torch.from_numpy(cm).to(DEV)

    T = bs * nd
    q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)
    seq_t = torch.tensor(seq_lens, dtype=torch.int64)
    fb = ForwardBatch(forward_mode=ForwardMode.TARGET_VERIFY, batch_size=bs, req_pool_indices=rpi,
                      seq_lens=seq_t.to(DEV), out_cache_loc=loc, seq_lens_sum=int(seq_t.sum()), seq_lens_cpu=seq_t,
                      spec_info=Spec)
    hs.backend.init_forward_metadata(fb)
    o = hs.layer(q, k, v, fb, hs.backend)
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = hs.r2t.req_to_token.cpu().numpy()
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, np.array(rows), np.array(seq_lens))
    qo = (np.arange(bs + 1) * nd).astype(np.int64)
    kbn, vbn = kb.cpu().numpy(), vb.cpu().numpy()
    ke = np.concatenate([kbn[r2t[rows[i], seq_lens[i]: total[i]]] for i in range(bs)])
    ve = np.concatenate([vbn[r2t[rows[i], seq_lens[i]: total[i]]] for i in range(bs)])
    mi = np.concatenate([[0], np.cumsum([m.size for m in masks])]).astype(np.int64)
    want, absw = parity.want_and_absw(orc.extend_attention, (q.view(T, hq, d).cpu().numpy(), ke, ve, kbn, vbn, qo, kv_indptr, kv_indices),
                                      (2, 4), is_causal=True, sm_scale=d ** -0.5, custom_mask=cm, mask_indptr=mi, skip_prefix_custom_mask=False)
    parity.check_out(o.view(T, hq, d).float().cpu().numpy(), want, torch.float16, "deterministic target verify", absw=absw)


def test_window_start_pos_shifts_both_sides_of_the_window_test_and_cancels(golden_dir):
    """extend_attention_fwd_unified's window_start_pos (extend_attention.py:1010-1027): the reference adds it to the query's
    AND the key's absolute position before the window test, so any value gives the same mask; here the argument is accepted
    and has no effect -- checked against the reference's own sliding-window output (F12 `swa`, produced with zeros) while
    passing non-zero starts."""
    from sglang_amd import ops

    z = np.load(os.path.join(golden_dir, "extend_unified.npz"))
    c = {k.split(".", 1)[1]: z[k] for k in z.files if k.startswith("swa.")}
    q, kb, vb = _t(c["q"]), _t(c["kb"]), _t(c["vb"])
    outs = []
    for wsp in (None, torch.tensor([7, 123], dtype=torch.int32, device=DEV)):
        o = torch.zeros_like(q)
        ops.extend_attention_fwd_unified(q, o, kb, vb, 1.0, 1.0, _t(c["qo_indptr"]), _t(c["kv_indptr"]), _t(c["kv_indices"]),
                                         _t(c["prefix_lens"]), int(np.diff(c["qo_indptr"]).max()), sm_scale=float(c["sm_scale"]),
                                         sliding_window_size=int(c["window"]), window_start_pos=wsp)
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    want = c["o"].astype(np.float64)
    ok = np.isfinite(want).all(axis=-1)
    ref, absw = parity.want_and_absw(orc.extend_attention_unified, (c["q"], c["kb"], c["vb"], c["qo_indptr"], c["kv_indptr"], c["kv_indices"],
                                                                  c["prefix_lens"]), (2,), sm_scale=float(c["sm_scale"]),
                                     sliding_window_size=int(c["window"]))
    parity.check_out(outs[1].float().cpu().numpy()[ok], want[ok], torch.float16, "window_start_pos vs triton golden", ulps=2, absw=2 * absw[ok])
