"""rx_split_items + rx_decode_params.split_items (round 3): the decode kernel's grid as the compacted list of live
(request, split) pairs, longest requests first, instead of bs x max_kv_splits split slots.  The table is checked
against its definition, the decode outputs must be BIT-identical to the slot form (same workgroups, same arithmetic,
another dispatch order), and the whole thing against the oracle."""
import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_split_items_table_matches_its_definition():
    from sglang_amd import ops

    rng = np.random.default_rng(0)
    for bs in (1, 7, 64, 1500, 3000):
        splits = rng.integers(0, 6, size=bs).astype(np.int32)   # 0 counts as 1 (a request always has its workgroup)
        order = rng.permutation(bs).astype(np.int32)
        for use_order in (False, True):
            want = [(int(b), s) for b in (order if use_order else range(bs)) for s in range(max(int(splits[b]), 1))]
            for cap in (len(want), len(want) + 5, max(1, len(want) // 2)):
                si = ops.SplitItems(max(cap, 1), DEV)
                si.items.fill_(-7)
                si.build(torch.from_numpy(splits).to(DEV), torch.from_numpy(order).to(DEV) if use_order else None, cap=cap)
                torch.cuda.synchronize()
                assert int(si.count.item()) == len(want)              # the count says how many there WOULD be
                got = si.items.cpu().numpy().reshape(-1, 2)
                n = min(cap, len(want))
                assert got[:n].tolist() == [list(x) for x in want[:n]]
                assert (got[n:] == -7).all()                           # nothing written past cap


def test_guarded_build_replaces_a_schedule_that_outgrows_its_cap():
    """rx_split_items_guarded: within cap = the plain table, counts untouched; beyond cap the SCHEDULE is
    replaced (every count 1, bs whole-request pairs in launch order, count = bs, overflow set) -- never dropped pairs."""
    from sglang_amd import ops

    rng = np.random.default_rng(1)
    for bs in (1, 9, 257, 2500):
        splits = rng.integers(1, 7, size=bs).astype(np.int32)
        order = rng.permutation(bs).astype(np.int32)
        total = int(splits.sum())
        for use_order in (False, True):
            od = torch.from_numpy(order).to(DEV) if use_order else None
            want = [[int(b), s] for b in (order if use_order else range(bs)) for s in range(int(splits[b]))]
            for cap in (total, total + 3, max(bs, total - 1), bs):
                si = ops.SplitItems(cap, DEV)
                si.items.fill_(-7)
                sp = torch.from_numpy(splits).to(DEV)
                si.build(sp, od, cap=cap, guarded=True)
                torch.cuda.synchronize()
                got = si.items.cpu().numpy().reshape(-1, 2)
                if total <= cap:
                    assert int(si.count.item()) == total and int(si.overflow.item()) == 0
                    assert got[:total].tolist() == want and sp.cpu().numpy().tolist() == splits.tolist()
                else:
                    assert int(si.count.item()) == bs and int(si.overflow.item()) == 1
                    assert sp.cpu().numpy().tolist() == [1] * bs
                    assert got[:bs].tolist() == [[int(b), 0] for b in (order if use_order else range(bs))]
    with pytest.raises(Exception):  # cap < bs cannot hold even the whole-request schedule
        ops.SplitItems(4, DEV).build(torch.ones(8, dtype=torch.int32, device=DEV), None, cap=4, guarded=True)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lookup", ["paged", "indices"])
def test_decode_with_split_items_is_bit_identical_to_split_slots(dtype, lookup):
    from sglang_amd import ops

    hq, hkv, d, ps = 32, 8, 128, 16
    lens = np.array([9000, 1, 0, 700, 2049, 33, 4096, 128, 5000, 17], dtype=np.int64)
    bs = len(lens)
    rng = np.random.default_rng(3)
    pages = [-(-int(n) // ps) for n in lens]
    perm = rng.permutation(np.arange(1, sum(pages) + 1))
    r2t = np.zeros((bs + 1, int(lens.max()) + ps), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        r2t[i + 1, :n] = (perm[pi: pi + pages[i], None] * ps + np.arange(ps)[None]).reshape(-1)[:n]
        pi += pages[i]
    pool = (sum(pages) + 1) * ps
    g = torch.Generator().manual_seed(1)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    S = 16
    splits = torch.zeros(bs, dtype=torch.int32, device=DEV)
    lens_d = torch.from_numpy(lens).to(DEV)
    ops.get_num_kv_splits_balanced(splits, lens_d, hq, hkv, S, 512, 128)
    order = torch.argsort(lens_d, descending=True).to(torch.int32)
    kbd, vbd, qd = kb.to(DEV), vb.to(DEV), q.to(DEV)
    ip, ii = orc.build_kv_indices(r2t, rpi, lens)
    outs = []
    for items in (None, "exact", "upper", "three_per_cu"):
        o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=DEV)
        al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
        ls = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
        si = None
        if items:
            n_live = int(splits.clamp_min(1).sum())
            # "three_per_cu": the kernel's OCC3 instance (rx_decode_params.split_items_wgs_per_cu): same arithmetic
            si = ops.SplitItems(n_live if items != "upper" else bs * S, DEV).build(
                splits, order, wgs_per_cu=3 if items == "three_per_cu" else 0)
        if lookup == "paged":
            ops.decode_attention_fwd_paged(qd, kbd, vbd, o, torch.from_numpy(r2t).to(DEV), torch.from_numpy(rpi).to(DEV),
                                           lens_d, al, ls, splits, S, d ** -0.5, page_size=ps, merge_counters=cnt,
                                           request_order=order, split_items=si)
        else:
            ops.decode_attention_fwd(qd, kbd, vbd, o, torch.from_numpy(ip).to(DEV), torch.from_numpy(ii).to(DEV), al, ls, splits,
                                     S, d ** -0.5, 1.0, 1.0, page_size=ps, merge_counters=cnt, request_order=order,
                                     split_items=si)
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0            # the in-kernel merge left its counters at zero
        outs.append(o)
    a, b, c, e = (x.view(torch.int16).cpu().numpy() for x in outs)
    assert (a == b).all() and (a == c).all() and (a == e).all()
    live = lens > 0
    want = orc.decode_attention(q.view(torch.uint16).numpy() if dtype == torch.bfloat16 else q.numpy(),
                                kb.view(torch.uint16).numpy() if dtype == torch.bfloat16 else kb.numpy(),
                                vb.view(torch.uint16).numpy() if dtype == torch.bfloat16 else vb.numpy(), ip, ii, d ** -0.5)
    parity.check_out(outs[1].float().cpu().numpy()[live], want[live], dtype, ("split items", lookup), ulps=1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_store_with_split_items_at_two_and_three_workgroups_per_cu(dtype):
    """The step's KV store inside the decode launch (rx_decode_params.k_new) on the live-pairs grid: the usual kernel and
    its three-per-CU instance (which re-reads the newest token's slot instead of carrying it through the loop) must write
    the same pool rows and outputs as store-then-decode on the split-slot grid, bit for bit."""
    from sglang_amd import ops

    hq, hkv, d, ps = 32, 8, 128, 16
    lens = np.array([6000, 1, 300, 2049, 33, 4100, 17, 5000], dtype=np.int64)  # lengths INCLUDING the new token
    bs = len(lens)
    rng = np.random.default_rng(9)
    pages = [-(-int(n) // ps) for n in lens]
    perm = rng.permutation(np.arange(1, sum(pages) + 1))
    r2t = np.zeros((bs + 1, int(lens.max()) + ps), dtype=np.int32)
    pi = 0
    for i, n in enumerate(lens):
        r2t[i + 1, :n] = (perm[pi: pi + pages[i], None] * ps + np.arange(ps)[None]).reshape(-1)[:n]
        pi += pages[i]
    pool = (sum(pages) + 1) * ps
    g = torch.Generator().manual_seed(4)
    kb0 = torch.randn(pool, hkv, d, generator=g).to(dtype).to(DEV)
    vb0 = torch.randn(pool, hkv, d, generator=g).to(dtype).to(DEV)
    q = torch.randn(bs, hq, d, generator=g).to(dtype).to(DEV)
    kn = torch.randn(bs, hkv, d, generator=g).to(dtype).to(DEV)
    vn = torch.randn(bs, hkv, d, generator=g).to(dtype).to(DEV)
    r2t_d = torch.from_numpy(r2t).to(DEV)
    rpi = torch.arange(1, bs + 1, device=DEV)
    lens_d = torch.from_numpy(lens).to(DEV)
    S = 16
    splits = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits_balanced(splits, lens_d, hq, hkv, S, 512, 128)
    order = torch.argsort(lens_d, descending=True).to(torch.int32)
    new_slots = r2t_d[rpi, (lens_d - 1)].long()

    def run(mode):
        kb, vb = kb0.clone(), vb0.clone()
        o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=DEV)
        al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
        ls = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
        if mode == "store_then_decode":
            kb[new_slots], vb[new_slots] = kn, vn
            ops.decode_attention_fwd_paged(q, kb, vb, o, r2t_d, rpi, lens_d, al, ls, splits, S, d ** -0.5, page_size=ps,
                                           merge_counters=cnt, request_order=order)
        else:
            si = ops.SplitItems(int(splits.clamp_min(1).sum()), DEV).build(splits, order, wgs_per_cu=3 if mode == "three" else 0)
            ops.decode_attention_fwd_paged(q, kb, vb, o, r2t_d, rpi, lens_d, al, ls, splits, S, d ** -0.5, page_size=ps,
                                           merge_counters=cnt, request_order=order, split_items=si, k_new=kn, v_new=vn)
        torch.cuda.synchronize()
        return o.view(torch.int16).cpu().numpy(), kb.view(torch.int16).cpu().numpy(), vb.view(torch.int16).cpu().numpy()

    ref = run("store_then_decode")
    for mode in ("two", "three"):
        got = run(mode)
        for a, b, what in zip(ref, got, ("o", "k pool", "v pool")):
            assert (a == b).all(), (mode, what)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fuse", [False, True], ids=["plain", "fused_store"])
def test_decode_units_tables_match_their_definition_and_change_no_bit(dtype, fuse):
    """rx_decode_units + rx_decode_params.unit_desc / unit_first_slots (round 6): the per-unit descriptor (request, split,
    length, split count, row offset, token range) and the first tiles' slot ids against their definition, and the decode
    outputs with the tables BIT-identical to the same launches without them -- unsplit (whole requests in launch order), a
    split schedule on the live-pairs grid (exact and upper-bound caps), with and without the fused store of the new token;
    ragged lengths incl. an empty request, one token, lengths below one tile per wave."""
    from sglang_amd import ops

    hq, hkv, d, ps = 32, 8, 128, 16
    lens = np.array([6000, 1, 0 if not fuse else 2, 300, 2049, 33, 4100, 17, 5000, 129], dtype=np.int64)
    bs = len(lens)
    rng = np.random.default_rng(11)
    pages = [-(-int(n) // ps) for n in lens]
    perm = rng.permutation(np.arange(1, sum(pages) + 1))
    r2t = np.zeros((bs + 3, int(lens.max()) + ps), dtype=np.int32)
    rows = rng.permutation(np.arange(1, bs + 3))[:bs]              # request rows in any order
    pi = 0
    for i, n in enumerate(lens):
        r2t[rows[i], :n] = (perm[pi: pi + pages[i], None] * ps + np.arange(ps)[None]).reshape(-1)[:n]
        pi += pages[i]
    pool = (sum(pages) + 1) * ps
    g = torch.Generator().manual_seed(2)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype).to(DEV)
    kn = torch.randn(bs, hkv, d, generator=g).to(dtype).to(DEV) if fuse else None
    vn = torch.randn(bs, hkv, d, generator=g).to(dtype).to(DEV) if fuse else None
    r2t_d, rpi = torch.from_numpy(r2t).to(DEV), torch.from_numpy(rows.astype(np.int64)).to(DEV)
    lens_d = torch.from_numpy(lens).to(DEV)
    S = 16
    splits = torch.zeros(bs, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits_balanced(splits, lens_d, hq, hkv, S, 512, 128)
    order = torch.argsort(lens_d, descending=True).to(torch.int32)
    sp = splits.cpu().numpy()

    def split_range(n, s, i):  # decode_attention.py:466-472
        per = (-(-n // s) + 31) // 32 * 32
        return per * i, min(per * i + per, n)

    # ---- the tables against their definition
    for mode in ("unsplit", "items_exact", "items_upper"):
        si = None
        if mode != "unsplit":
            n_live = int(splits.clamp_min(1).sum())
            si = ops.SplitItems(n_live if mode == "items_exact" else bs * S, DEV).build(splits, order)
        units = ops.DecodeUnits(bs if si is None else si.cap, DEV).build(r2t_d, rpi, lens_d, splits if si is not None else None,
                                                                        S if si is not None else 1, si, order)
        torch.cuda.synchronize()
        desc = units.desc.cpu().numpy().reshape(-1, 8)
        first = units.first.cpu().numpy().reshape(-1, 128)
        od = order.cpu().numpy()
        pairs = ([(int(b), 0, 1) for b in od] if si is None else
                 [(int(b), s, int(sp[b])) for b in od for s in range(max(int(sp[b]), 1))])
        for u, (b, s, nsp) in enumerate(pairs):
            n = int(lens[b])
            lo, hi = split_range(n, nsp, s) if nsp > 0 and s < nsp else (0, 0)
            row = int(rows[b]) * r2t.shape[1]
            assert desc[u].tolist() == [b, s, n, nsp, row & 0xffffffff, row >> 32, lo, hi], (mode, u)
            want = [int(r2t[rows[b], min(lo + j, hi - 1)]) if hi > lo else 0 for j in range(128)]
            assert first[u].tolist() == want, (mode, u)

    # ---- decode with and without the tables: the same bits (and the same stored rows)
    outs, pools = {}, {}
    for mode in ("unsplit", "split"):
        for with_units in (False, True):
            kbd, vbd = kb.to(DEV), vb.to(DEV)
            o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=DEV)
            kw = dict(page_size=ps, request_order=order, k_new=kn, v_new=vn)
            if mode == "unsplit":
                units = ops.DecodeUnits(bs, DEV).build(r2t_d, rpi, lens_d, None, 1, None, order) if with_units else None
                ops.decode_attention_fwd_paged(q, kbd, vbd, o, r2t_d, rpi, lens_d, None, None, None, 1, d ** -0.5, units=units, **kw)
            else:
                al = torch.zeros(bs, hq, S, d, dtype=torch.float32, device=DEV)
                ls = torch.zeros(bs, hq, S, dtype=torch.float32, device=DEV)
                cnt = torch.zeros(bs * hq, dtype=torch.int32, device=DEV)
                si = ops.SplitItems(bs * S, DEV).build(splits, order)
                units = ops.DecodeUnits(si.cap, DEV).build(r2t_d, rpi, lens_d, splits, S, si, order) if with_units else None
                ops.decode_attention_fwd_paged(q, kbd, vbd, o, r2t_d, rpi, lens_d, al, ls, splits, S, d ** -0.5, merge_counters=cnt,
                                               split_items=si, units=units, **kw)
                torch.cuda.synchronize()
                assert int(cnt.abs().sum()) == 0
            torch.cuda.synchronize()
            outs[(mode, with_units)] = o.view(torch.int16).cpu().numpy()
            pools[(mode, with_units)] = (kbd.view(torch.int16).cpu().numpy(), vbd.view(torch.int16).cpu().numpy())
    for mode in ("unsplit", "split"):
        assert (outs[(mode, False)] == outs[(mode, True)]).all(), mode
        assert (pools[(mode, False)][0] == pools[(mode, True)][0]).all() and (pools[(mode, False)][1] == pools[(mode, True)][1]).all(), mode
    # ... and they are the oracle's result (the fused store's rows included)
    kb2, vb2 = kb.clone(), vb.clone()
    if fuse:
        last = torch.from_numpy(np.array([r2t[rows[i], lens[i] - 1] for i in range(bs)], dtype=np.int64))
        kb2[last], vb2[last] = kn.cpu(), vn.cpu()
    ip, ii = orc.build_kv_indices(r2t, rows.astype(np.int64), lens)
    bits = (lambda t: t.view(torch.uint16).numpy()) if dtype == torch.bfloat16 else (lambda t: t.numpy())
    want, absw = orc.decode_attention(bits(q.cpu()), bits(kb2), bits(vb2), ip, ii, d ** -0.5, return_absw=True)
    live = lens > 0
    got = torch.from_numpy(outs[("split", True)]).view(dtype).float().numpy()
    parity.check_out(got[live], want[live], dtype, ("decode units", fuse), ulps=1, absw=absw[live])
