"""Cascade over SEVERAL shared prefixes (one per radix-tree node).

CPU: the planner (mem_cache.radix_cache.plan_shared_prefix_groups, from RadixCache.match_prefix nodes --
srt/mem_cache/radix_cache.py:352-430) and the host half of ops.CascadeGroups.plan.  GPU: a batch with three shared
prefixes of different depth plus requests that share nothing, through ops.CascadeGroups and through the backend, against
the fp64 oracle of plain decode attention (decode_attention_fwd, kernels/ops/attention/decode_attention.py:820-912)."""
import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

DEV = "cuda"


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _tree_batch(rng, spec):
    """spec: list of request token lists.  Inserts them into a RadixCache (slot = running counter) and returns the
    cache, each request's last node and each request's slot row as the tree hands it out (shared prefixes share slots)."""
    from sglang_amd.mem_cache.radix_cache import InsertParams, MatchPrefixParams, RadixCache, RadixKey

    rc = RadixCache.create_simulated()
    nodes, rows, slot = [], [], 1
    for toks in spec:
        m = rc.match_prefix(MatchPrefixParams(RadixKey(toks)))
        have = m.device_indices.cpu().numpy().astype(np.int64)
        own = np.arange(slot, slot + len(toks) - len(have), dtype=np.int64)
        slot += len(own)
        row = np.concatenate([have, own])
        rc.insert(InsertParams(RadixKey(toks), torch.from_numpy(row)))
        rows.append(row)
    for toks in spec:  # nodes after all inserts: splits have happened by now
        nodes.append(rc.match_prefix(MatchPrefixParams(RadixKey(toks))).last_device_node)
    return rc, nodes, rows, slot


def _spec(rng, sys_a=700, sys_b=400, doc=300):
    A = rng.integers(0, 5000, sys_a).tolist()
    B = rng.integers(0, 5000, sys_b).tolist()
    D1, D2 = rng.integers(0, 5000, doc).tolist(), rng.integers(0, 5000, doc).tolist()
    tail = lambda n: rng.integers(5000, 9000, n).tolist()  # noqa: E731
    reqs = [A + D1 + tail(20 + i) for i in range(5)]        # 0-4: system prompt A + document 1
    reqs += [A + D2 + tail(33 + i) for i in range(4)]       # 5-8: A + document 2
    reqs += [B + tail(50 + 7 * i) for i in range(6)]        # 9-14: system prompt B
    reqs += [tail(90), tail(260)]                           # 15, 16: nothing shared
    return reqs


def test_planner_picks_the_antichain_with_the_largest_saving():
    from sglang_amd.mem_cache.radix_cache import plan_shared_prefix_groups

    rng = np.random.default_rng(3)
    reqs = _spec(rng)
    _, nodes, rows, _ = _tree_batch(rng, reqs)
    lens = [len(r) for r in reqs]
    groups = plan_shared_prefix_groups(nodes, lens, min_shared=256, min_members=2)
    # A's children win: 4 * 1000 + 3 * 1000 = 7000 rows saved against 8 * 700 = 5600 for A itself
    assert groups == [(list(range(0, 5)), 1000), (list(range(5, 9)), 1000), (list(range(9, 15)), 400)]
    for members, L in groups:  # the shared slots really are the same
        for m in members:
            assert np.array_equal(rows[m][:L], rows[members[0]][:L])
    # a deeper threshold on members: document 2 (4 requests) no longer forms a group, A (9 requests) is better then
    g5 = plan_shared_prefix_groups(nodes, lens, min_shared=256, min_members=5)
    assert g5 == [(list(range(0, 9)), 700), (list(range(9, 15)), 400)]
    # min_shared above B: only the documents remain
    assert plan_shared_prefix_groups(nodes, lens, min_shared=512) == [(list(range(0, 5)), 1000), (list(range(5, 9)), 1000)]
    assert plan_shared_prefix_groups(nodes[15:], lens[15:], min_shared=1) == []
    assert plan_shared_prefix_groups([], None) == []


def test_layout_tables_cover_every_group_prefix_once():
    from sglang_amd.ops import CascadeGroups

    groups = [([3, 4, 9], 1000), ([0, 7], 130), ([1, 2, 5, 6, 8, 10], 64)]
    bs, hq = 12, 8
    for cu, max_chunks in ((256, 16), (8, 16), (256, 3)):
        lay = CascadeGroups.layout(groups, bs, hq, cu, max_chunks)
        C, G, M = lay["C"], lay["G"], lay["M"]
        assert G == 3 and M == 11 and 1 <= C <= max_chunks
        kv, qo = lay["kv_indptr"], lay["qo_indptr"]
        assert kv[0] == 0 and np.all(np.diff(kv) >= 0) and np.all(np.diff(qo) >= 0) and qo[-1] == C * M
        starts = [0, 3, 5]
        for g, (members, L) in enumerate(groups):
            covered = []
            for c in range(C):
                p = c * G + g
                n = kv[p + 1] - kv[p]
                covered += list(range(lay["src_col"][p], lay["src_col"][p] + n))
                assert qo[p] == c * M + starts[g] and qo[p + 1] - qo[p] == len(members)
                assert lay["src_member"][p] == members[0]
            assert covered == list(range(L))  # every shared token in exactly one chunk, in order
        assert lay["kv_start"].tolist() == [130, 64, 64, 1000, 1000, 64, 64, 130, 64, 1000, 64, 0]
        assert lay["extra_index"].tolist() == [3, 5, 6, 0, 1, 7, 8, 4, 9, 2, 10, -1]
        assert lay["gather"].tolist() == [3, 4, 9, 0, 7, 1, 2, 5, 6, 8, 10] * C
    with pytest.raises(ValueError):
        CascadeGroups.layout([([0, 1], 10), ([1, 2], 10)], 4, 8, 256, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("geom", [(8, 2, 128), (16, 1, 128), (4, 4, 64)], ids=["d128_gqa4", "d128_mqa16", "d64_mha"])
def test_three_prefix_batch_matches_plain_decode_oracle(dtype, geom):
    from sglang_amd import ops
    from sglang_amd.mem_cache.radix_cache import plan_shared_prefix_groups

    hq, hkv, d = geom
    rng = np.random.default_rng(11)
    reqs = _spec(rng)
    _, nodes, rows, nslots = _tree_batch(rng, reqs)
    bs = len(reqs)
    lens = np.asarray([len(r) for r in reqs], dtype=np.int64)
    ctx = int(lens.max()) + 8
    # slots -> pool rows: a random injective map (pages of `page` tokens keep their inner order, as a paged pool would)
    perm = rng.permutation(nslots + 3)
    r2t = np.zeros((bs + 2, ctx), dtype=np.int32)
    order = rng.permutation(bs)  # batch order != tree order: groups are scattered over the batch
    rpi = np.zeros(bs, dtype=np.int64)
    for pos, i in enumerate(order):
        r2t[pos + 1, : lens[i]] = perm[rows[i]]
        rpi[pos] = pos + 1
    lens_b = lens[order]
    nodes_b = [nodes[i] for i in order]
    groups = plan_shared_prefix_groups(nodes_b, lens_b.tolist(), min_shared=256, min_members=2)
    assert len(groups) == 3 and sum(len(m) for m, _ in groups) == 15
    pool = int(perm.max()) + 1
    g = torch.Generator().manual_seed(5)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    sm = d ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens_b)
    want, absw = parity.want_and_absw(orc.decode_attention, (_bits(q), _bits(kb), _bits(vb), kv_indptr, kv_indices, sm), (2,))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    cg = ops.CascadeGroups(bs + 3, hq, hkv, d, dtype, DEV, max_shared_total=4096)
    r2t_d, rpi_d, lens_d = T(r2t), T(rpi), T(lens_b)
    outs = []
    for gsel in (groups, groups[:1], []):  # three prefixes; one prefix that only part of the batch shares; none
        cg.plan(r2t_d, rpi_d, lens_d, gsel)
        o = torch.full((bs, hq, d), float("nan"), dtype=dtype, device=DEV)
        cg(q.to(DEV), kb.to(DEV), vb.to(DEV), o, sm, page_size=1)
        torch.cuda.synchronize()
        got = o.float().cpu().numpy().astype(np.float64)
        assert not np.isnan(got).any()
        parity.check_out(got, want, dtype, ("cascade groups", len(gsel), geom), ulps=2, absw=absw)  # two 16-bit roundings (the chunk partials cross 16-bit buffers before the merge): 2 ulp
        outs.append(got)
    # requests in no group never see a partial: their rows are those of the plain decode kernel, bit for bit
    loners = [pos for pos, i in enumerate(order) if i >= 15]
    assert np.array_equal(outs[0][loners], outs[2][loners])
    # the plan's tables
    total = sum(L for _, L in groups)
    cg.plan(r2t_d, rpi_d, lens_d, groups)
    assert int(cg.kv_indptr[-1]) == total
    ks = cg.kv_start.cpu().numpy()
    for members, L in groups:
        assert (ks[members] == L).all()
    assert (ks[loners] == 0).all()


@pytest.mark.gpu
def test_backend_cascades_per_radix_node_when_the_scheduler_hands_the_nodes_over():
    """HipRadixAttnBackend(cascade_decode=True) with forward_batch.radix_last_nodes: the same three-prefix batch, one
    decode step with the KV store through the layer, against the oracle's torch-native semantics."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.forward_batch import ForwardBatch
    from tests.test_gpu_backend import _Harness
    from tests.test_gpu_cascade import _runner_of

    hq, hkv, d = 8, 2, 128
    rng = np.random.default_rng(21)
    reqs = _spec(rng, sys_a=300, sys_b=200, doc=150)
    _, nodes, rows_tree, _ = _tree_batch(rng, reqs)
    bs = len(reqs)
    hs = _Harness(1, hq, hkv, d, torch.bfloat16, "contiguous", "paged", max_ctx=1200, max_reqs=bs + 2)
    hs.backend = HipRadixAttnBackend(_runner_of(hs), cascade_decode=True, cascade_min_bs=2, cascade_min_shared=128)
    hs.backend.cascade_min_members = 2
    rows = hs.r2t.alloc(bs)
    # one pool slot per tree slot: requests that share tree slots share pool rows
    n_tree = int(max(r.max() for r in rows_tree)) + 1
    slots = hs.alloc.alloc(n_tree)
    hs.pool.set_kv_buffer(hs.layer, slots, hs.rand(n_tree, hkv, d), hs.rand(n_tree, hkv, d))
    prefix_lens = [len(r) for r in rows_tree]
    for r, tr in zip(rows, rows_tree):
        hs.r2t.req_to_token[r, : len(tr)] = slots[torch.from_numpy(tr).to(DEV)].to(torch.int32)
    seq_lens = [p + 1 for p in prefix_lens]
    seq_t = torch.tensor(seq_lens, dtype=torch.int64)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    loc = hs.alloc.alloc(bs)
    hs.r2t.req_to_token[rpi, torch.tensor(prefix_lens, device=DEV)] = loc.to(torch.int32)
    q, k, v = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    fb = ForwardBatch.for_decode(rpi, seq_t.to(DEV), loc, seq_t)
    fb.radix_last_nodes = nodes
    hs.backend.init_forward_metadata(fb)
    o = hs.layer(q, k, v, fb, hs.backend)
    cg = hs.backend._cascade_groups
    assert cg is not None and cg.num_groups == 3 and cg.members == 15
    kb, vb = hs.pool.get_kv_buffer(0)
    want, absw = parity.want_and_absw(orc.sdpa_decode_req_to_token, (_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb),
                                                                     _bits(hs.r2t.req_to_token), np.array(rows),
                                                                     np.array(seq_lens), d ** -0.5), (2,))
    got = o.view(bs, hq, d).float().cpu().numpy().astype(np.float64)
    assert hs.pool.check_errors() == 0
    parity.check_out(got, want, o.dtype, "backend cascade groups", ulps=2, absw=absw)  # two 16-bit roundings (the chunk partials cross 16-bit buffers before the merge): 2 ulp


def test_rounds_rule_of_the_balanced_schedule_host_mirror():
    """ops.balanced_kv_splits_host with wg_target_mixed = -1 (the mirror of rx_num_kv_splits_balanced's rounds rule; the
    device kernel is held to it in tests/test_gpu_backend.py): pieces of R x the mean unsplit length when the first pass
    needs R >= 2 rounds of workgroups, the first pass itself otherwise, never finer than the first pass."""
    from sglang_amd import ops

    first = lambda lens, mt=1024: ops.balanced_kv_splits_host(lens, 32, 8, 64, 512, mt)  # noqa: E731
    rr = lambda lens, mt=1024: ops.balanced_kv_splits_host(lens, 32, 8, 64, 512, mt, -1)  # noqa: E731
    skew = [32768] + [1024] * 63
    assert first(skew)[0] == 22 and rr(skew)[0] == 16 and (rr(skew)[1:] == 1).all()      # 680 workgroups = 2 rounds: 2 x 1 k pieces
    one_round = [16384] * 2 + [2048] * 30
    assert (rr(one_round) == first(one_round)).all()                                    # 432 workgroups fit at once: unchanged
    uniform = [4096] * 16
    assert (rr(uniform) == first(uniform)).all() and (rr(uniform) == 4).all()           # not mixed: unchanged
    tiny = [30000] * 20 + [16]
    assert (rr(tiny, 128) <= first(tiny, 128)).all()                                    # a tiny unsplit request does not shred the rest
    ragged = [32768] + np.random.default_rng(2).integers(300, 2000, size=63).tolist()
    a, b = first(ragged), rr(ragged)
    assert b[0] <= a[0] and (b[1:] == a[1:]).all() and int(b.sum()) <= int(a.sum())
    assert (rr([]) == first([])).all() and rr([0, 0]).tolist() == [1, 1]


def test_fill_rule_of_the_balanced_schedule_host_mirror():
    """Round 4: near-uniform batches of 0.7 - 3 whole-request workgroups per CU (wg_target = 2 x 256 CUs) take one count
    for everybody -- nobody is cut up to one per CU, above it the smallest count whose pieces fill >= 85 % of whole rounds
    of CUs (the optima of tools/decode_sweep.py on the TP = 8 shard: 288 -> 4, 320 -> 3, 384 -> 2, 448 -> 1, 640 -> 2)."""
    from sglang_amd import ops

    tp8 = lambda lens, mixed=-1, cap=32: ops.balanced_kv_splits_host(lens, 4, 1, cap, 512, 1024, mixed)  # noqa: E731
    for bs, want in ((192, 1), (224, 1), (256, 1), (288, 4), (320, 3), (384, 2), (448, 1), (512, 1), (640, 2), (768, 1)):
        got = tp8([4096] * bs)
        assert (got == want).all(), (bs, got[:4])
    assert (tp8([4096] * 256, 0) == 2).all() and (tp8([4096] * 256, 768) == 1).all()    # off without a live-pairs schedule
    assert (tp8([4096] * 128) == tp8([4096] * 128, 0)).all()                             # below 0.7 per CU: the even share where it fills (see below)
    assert (tp8([4096] * 1024) == 1).all()                                               # 4 per CU: nobody is cut
    assert tp8([4096] * 288, cap=2).max() == 2                                           # the cap holds (best fill under it)
    short = tp8([300] * 320)
    assert (short == 1).all()                                                            # pieces of at least 256 tokens
    assert tp8([600] * 320).max() == 2
    skew = [32768] + [1024] * 319
    assert tp8(skew)[0] > 6 and (tp8(skew)[1:] == 1).all()                               # not near-uniform: the rounds rule's business
    assert (tp8([0] * 64 + [4096] * 256) == 1).all()                                     # empty requests do not count as blocks
    ragged = np.random.default_rng(3).integers(3000, 4097, size=320)
    assert (tp8(ragged) == 3).all()                                                      # 2 max <= 3 mean: still one count
    assert (ops.balanced_kv_splits_host([4096] * 40, 32, 8, 32, 512, 1024, -1) == 3).all()  # 320 blocks at Hkv = 8 as well
    # below 0.7 per CU everybody is cut; the even share's count stands when its workgroups fill whole rounds of CUs
    # (64 -> 8 = 512, 128 -> 4 = 512) and is replaced by the nearest count that does when they do not (tools/decode_sweep.py:
    # 176 -> 3 = 528 workgroups 92 us, 4 = 704: 77; 104 x 8 k -> 5: 108 us, 7: 92; 160 -> 4: 71, 3: 67; 100 -> 6: 49, 5: 45)
    low = lambda bs, ctx=4096, mixed=-1: ops.balanced_kv_splits_host([ctx] * bs, 4, 1, 32, 512, 128, mixed)  # noqa: E731
    for bs, ctx, want, old in ((64, 4096, 8, 8), (128, 4096, 4, 4), (176, 4096, 4, 3), (104, 8192, 7, 5), (160, 4096, 3, 4),
                               (100, 4096, 5, 6), (144, 4096, 5, 4), (72, 8192, 7, 8)):
        assert (low(bs, ctx) == want).all() and (low(bs, ctx, 0) == old).all(), (bs, ctx)
    assert (ops.balanced_kv_splits_host([4096] * 20, 32, 8, 32, 512, 1024, -1) == 3).all()   # 160 blocks at Hkv = 8, 1 k pieces
    assert (ops.balanced_kv_splits_host([4096] * 18, 32, 8, 32, 512, 1024, -1) == 4).all()   # 144: no better count of >= 1 k pieces
    # the whole-requests form (0 < wg_target_mixed <= wg_target: kernels without the live-pairs grid, the MLA pools): from
    # 0.8 requests per CU up nobody is cut, whatever the count (tools/probe/mla_split_sweep.py: 224 / 256 / 320 / 384
    # requests all fastest whole); below it the even share as before
    mla = lambda lens: ops.balanced_kv_splits_host(lens, 16, 1, 32, 512, 128, 512)  # noqa: E731
    for bs, want in ((64, 8), (128, 4), (192, 3), (224, 1), (256, 1), (320, 1), (384, 1), (1024, 1)):
        assert (mla([8192 if bs <= 192 else 4096] * bs) == want).all(), bs
    assert mla([32768] + [1024] * 255)[0] > 1                                           # not near-uniform: the long one is cut


def test_planner_caps_the_shared_length_below_every_member_and_survives_deep_chains():
    """L == seq_len would leave a member an empty suffix (its newest row never walked or stored), and the
    depth of a long radix chain must not recurse."""
    from sglang_amd.mem_cache.radix_cache import plan_shared_prefix_groups

    class N:  # the reference's TreeNode shape: .parent, .key
        def __init__(self, parent, n):
            self.parent, self.key = parent, [0] * n

    root = N(None, 0)
    shared = N(root, 300)
    a, b = N(shared, 5), N(shared, 9)
    # request 1 IS the shared prefix (seq_len 300): the group shares 299 tokens, not 300
    assert plan_shared_prefix_groups([a, shared, b], [305, 300, 309], min_shared=64) == [([0, 1, 2], 299)]
    assert plan_shared_prefix_groups([a, shared, b], None, min_shared=64) == [([0, 1, 2], 300)]
    n = root
    for _ in range(5000):  # deeper than Python's recursion limit
        n = N(n, 1)
    leaves = [N(n, 3), N(n, 4)]
    assert plan_shared_prefix_groups(leaves, [5003, 5004], min_shared=64) == [([0, 1], 5000)]


def test_cascade_groups_buffers_cover_every_reachable_layout():
    """the gathered-query / partial buffers hold rows_bound(max_bs) rows, not max_chunks * max_bs;
    every layout() of every member count stays inside, and the bound is ~16x smaller at serving sizes."""
    from sglang_amd.ops import CascadeGroups

    rng = np.random.default_rng(0)
    for max_bs, hq, cu in ((512, 32, 256), (4096, 32, 256), (300, 8, 256), (64, 64, 304)):
        bound = CascadeGroups.rows_bound(max_bs, hq, cu, 16)
        for _ in range(200):
            bs = int(rng.integers(1, max_bs + 1))
            rows = rng.permutation(bs)
            cuts = np.sort(rng.choice(np.arange(1, bs), size=min(bs - 1, int(rng.integers(0, 6))), replace=False)) if bs > 1 else []
            groups = [(m.tolist(), int(rng.integers(64, 4000))) for m in np.split(rows, cuts) if len(m)]
            if rng.random() < 0.5 and len(groups) > 1:
                groups = groups[:-1]  # some requests in no group
            lay = CascadeGroups.layout(groups, bs, hq, cu, 16)
            assert lay["C"] * lay["M"] <= bound, (max_bs, hq, cu, lay["C"], lay["M"], bound)
    assert CascadeGroups.rows_bound(4096, 32, 256, 16) <= 4096 + 128 * 8 + 128   # (was 16 * 4096)


def test_host_mirror_depends_on_the_cap_under_the_rounds_rule():
    """Why the eager metadata hands the device schedule the cap the host mirror ran with: a smaller cap
    ends the rounds-rule search earlier and yields MORE (request, split) pairs."""
    from sglang_amd import ops

    lens = [17702, 2856, 201, 518, 2486, 2851, 822, 1004, 2620, 1327, 892, 2500, 845, 1286, 1967, 1693, 348, 179, 2610,
            2285, 2529, 1660, 2470, 1056, 1412, 2386]
    h32 = ops.balanced_kv_splits_host(lens, 32, 8, 32, 512, 1024, -1)
    hS = ops.balanced_kv_splits_host(lens, 32, 8, int(h32.max()), 512, 1024, -1)
    assert int(h32.sum()) == 49 and int(hS.sum()) == 59
