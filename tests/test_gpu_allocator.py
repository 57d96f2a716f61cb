"""Device-resident allocators (sglang_amd/mem_cache/allocator.py over csrc/rx_pool.hip) against op logs RECORDED
FROM THE REFERENCE's TokenToKVPoolAllocator / PagedTokenToKVPoolAllocator (tests/golden/alloc_sequences.json, made by
tests/golden/make_golden.py::gen_allocator): every call's result and BOTH lists after every call, bit for bit --
page sizes 1 / 4 / 16 / 32, with and without need_sort, alloc / alloc_extend / alloc_decode / free / free_segment /
free_group / merge_and_sort_free, including the calls the reference answers with None."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(x):
    return torch.tensor(x, dtype=torch.int64, device=DEV)


def _make(case):
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator

    if case["page_size"] == 1:
        return TokenToKVPoolAllocator(case["size"], torch.bfloat16, DEV, None, case["need_sort"])
    return PagedTokenToKVPoolAllocator(case["size"], case["page_size"], torch.bfloat16, DEV, None, case["need_sort"])


def _replay(a, ent):
    op = ent["op"]
    if op == "alloc":
        return a.alloc(ent["need"])
    if op == "alloc_extend":
        pre, seq = torch.tensor(ent["prefix_lens"]), torch.tensor(ent["seq_lens"])
        return a.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, _t(ent["last_loc"]), int((seq - pre).sum()))
    if op == "alloc_decode":
        seq = torch.tensor(ent["seq_lens"])
        return a.alloc_decode(seq.to(DEV), seq, _t(ent["last_loc"]))
    if op == "free":
        a.free(_t(ent["idx"]))
    elif op == "free_segment":
        a.free_segment(_t(ent["idx"]), start_pos=ent["start_pos"])
    elif op == "merge_and_sort_free":
        a.merge_and_sort_free()
    elif op == "free_group":
        a.free_group_begin()
        for idx in ent["idx"]:
            a.free(_t(idx))
        a.free_group_end()
    else:
        raise AssertionError(op)
    return "void"


def test_reference_op_logs_replay_bit_exact():
    cases = json.load(open(os.path.join(GOLD, "alloc_sequences.json")))
    n_ops, n_none = 0, 0
    for case in cases:
        a = _make(case)
        for step, ent in enumerate(case["log"]):
            got = _replay(a, ent)
            tag = (case["page_size"], case["need_sort"], step, ent["op"])
            if got != "void":
                assert (got is None) == (ent["out"] is None), tag
                if got is not None:
                    assert got.dtype == torch.int64 and got.tolist() == ent["out"], tag
                else:
                    n_none += 1
            assert a.free_pages.tolist() == ent["free"][0], tag
            assert a.release_pages.tolist() == ent["free"][1], tag
            ps = case["page_size"]
            assert a.available_size() == (len(ent["free"][0]) + len(ent["free"][1])) * (1 if ps == 1 else ps), tag
            n_ops += 1
    assert n_ops > 250, n_ops


def test_mirror_stays_exact_without_reading_the_device():
    """Allocation, segment frees and sort-merges keep the host's length mirror exact: no device read-back happens
    (the reference synchronises in every paged free through torch.unique, paged.py:261-271).  Only a slot-list free
    makes a length unknown, and the next query re-reads it."""
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator

    ps = 16
    a = PagedTokenToKVPoolAllocator(64 * ps, ps, torch.bfloat16, DEV, None, need_sort=False)
    reads = []
    orig = a._list.counts
    a._list.counts = lambda: (reads.append(1), orig())[1]
    seq = torch.tensor([40, 17])
    pre = torch.zeros(2, dtype=torch.int64)
    out = a.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, _t([-1, -1]), 57)
    assert a.available_size() == (64 - 5) * ps
    seq1 = seq + 1
    dec = a.alloc_decode(seq1.to(DEV), seq1, torch.stack((out[39], out[56])))
    assert dec.tolist() == [int(out[39]) + 1, int(out[56]) + 1] and a.available_size() == (64 - 5) * ps
    a.free_segment(out[40:57], start_pos=0)       # request 2's 17 slots = 2 pages, count known on the host
    assert a.available_size() == (64 - 3) * ps and not reads
    a.free(out[:40])                              # data dependent: the mirror gives up ...
    assert a._n_free is None and not reads
    assert a.available_size() == 64 * ps and len(reads) == 1   # ... and re-reads lazily, once
    assert a.free_pages[:5].tolist() == [1, 2, 3, 4, 5]       # sorted set in front of [4, 5] in front of the rest


def test_out_of_pages_is_decided_on_the_host_before_any_launch():
    """the kernels never see a free list shorter than what they index."""
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator

    ps = 16
    a = PagedTokenToKVPoolAllocator(4 * ps, ps, torch.bfloat16, DEV, None)
    seq = torch.tensor([5 * ps])
    pre = torch.zeros(1, dtype=torch.int64)
    assert a.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, _t([-1]), 5 * ps) is None
    assert a.alloc(5 * ps) is None and a.available_size() == 4 * ps
    got = a.alloc(4 * ps)
    assert got.tolist() == list(range(ps, 5 * ps))
    one = torch.tensor([4 * ps + 1])
    assert a.alloc_decode(one.to(DEV), one, got[-1:]) is None    # needs a fifth page
    assert a._list.counts() == (0, 0, 0, 0)                     # nothing was refused ON the device


def test_double_free_is_detected_on_the_device():
    """the rings hold num_ids + 1 entries; a double free (or a page freed by free() AND free_segment())
    would wrap a list over live entries and hand a page out twice.  The grow / sorted-insert kernels count the
    event in a state word and the host's next read-back raises."""
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator

    a = TokenToKVPoolAllocator(64, torch.bfloat16, DEV, None)
    x = a.alloc(10)
    a.free(x)
    assert a.available_size() == 64
    a.free(x[:3])                                              # again: 67 ids in a pool of 64
    a._n_free = None
    with pytest.raises(RuntimeError, match="double free"):
        a.available_size()
    ps = 16
    b = PagedTokenToKVPoolAllocator(8 * ps, ps, torch.bfloat16, DEV, None)
    y = b.alloc(8 * ps)
    b.free(y)                                                  # sorted insert of 8 pages: the pool is whole again
    assert b.available_size() == 8 * ps
    b.free(y[: 2 * ps])                                        # the same pages once more
    b._n_free = None
    with pytest.raises(RuntimeError, match="double free"):
        b.available_size()


def test_large_pool_sorted_insert_and_merge_use_many_tiles():
    """1 Mi slots, page_size 1 with need_sort (ids span 513 compaction tiles) and a shuffled multi-tile paged
    free: the tiled scan keeps ascending order across tile boundaries."""
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator

    n = 1 << 20
    a = TokenToKVPoolAllocator(n, torch.bfloat16, DEV, None, need_sort=True)
    x = a.alloc(n - 7)
    g = torch.Generator(device=DEV).manual_seed(0)
    perm = x[torch.randperm(x.numel(), device=DEV, generator=g)]
    a.free(perm[: n // 2])
    a.free(perm[n // 2:])
    assert a.available_size() == n
    assert a.alloc(8) .tolist() == list(range(1, 9))            # 7 left in free, the merge brings the sorted rest
    fp = a.free_pages
    assert fp.numel() == n - 8 and bool((fp[1:] > fp[:-1]).all())
    ps = 16
    b = PagedTokenToKVPoolAllocator(n, ps, torch.bfloat16, DEV, None)
    y = b.alloc(n - 3 * ps)
    sub = y[torch.randperm(y.numel(), device=DEV, generator=g)][:300000]
    b.free(sub)
    want = torch.unique(sub // ps)                               # what the reference prepends (paged.py:261-271)
    fp = b.free_pages
    assert fp.numel() == want.numel() + 3 and torch.equal(fp[: want.numel()], want)
    assert fp[want.numel():].tolist() == [n // ps - 2, n // ps - 1, n // ps]


@pytest.mark.parametrize("ps", [1, 16])
def test_alloc_decode_rows_equals_gather_alloc_scatter(ps):
    """alloc_for_decode as one launch (rx_pool_alloc_decode_rows) == last-slot gather + alloc_decode (alloc for
    page_size 1) + row scatter, on the same free list: same out_cache_loc, same rows, same list afterwards."""
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator

    def make():
        return (TokenToKVPoolAllocator(4096, torch.bfloat16, DEV) if ps == 1
                else PagedTokenToKVPoolAllocator(4096, ps, torch.bfloat16, DEV))

    lens = torch.tensor([16, 1, 31, 32, 47, 100, 15])         # BEFORE the new token; 16 / 32: a fresh page at ps 16
    bs, rows = len(lens), torch.tensor([3, 1, 7, 2, 5, 4, 6], dtype=torch.int64, device=DEV)
    outs = []
    for fused in (False, True):
        a = make()
        g = torch.Generator(device=DEV).manual_seed(2)
        a.free_pages = a.free_pages[torch.randperm(a.free_pages.numel(), device=DEV, generator=g)]
        r2t = torch.zeros(8, 128, dtype=torch.int32, device=DEV)
        for r, n in zip(rows.tolist(), lens.tolist()):        # fill the rows through the allocator itself
            pre, seq = torch.zeros(1, dtype=torch.int64), torch.tensor([n])
            got = a.alloc(n) if ps == 1 else a.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, _t([-1]), n)
            r2t[r, :n] = got.to(torch.int32)
        if fused:
            loc = a.alloc_decode_rows(r2t, rows, lens.to(DEV), lens)
        else:
            last = r2t[rows, (lens - 1).to(DEV)].to(torch.int64)
            loc = a.alloc(bs) if ps == 1 else a.alloc_decode((lens + 1).to(DEV), lens + 1, last)
            r2t[rows, lens.to(DEV)] = loc.to(torch.int32)
        outs.append((loc.clone(), r2t.clone(), a.free_pages.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert outs[0][0].unique().numel() == bs


def test_alloc_extend_rows_replays_the_reference_logs_and_writes_the_rows():
    """rx_pool_alloc_extend_rows (alloc_for_extend as ONE copy + ONE launch): the reference op logs
    replayed with every `alloc_extend` -- and, at page size 1, every `alloc` -- routed through the one-call path (prefix
    tensors whose last slot is the log's last_loc): the returned slots, both free lists after every call and the calls the
    reference answers with None stay bit-exact, AND the req_to_token rows hold prefix slots + new slots
    (write_cache_indices, srt/mem_cache/allocation.py:55-101)."""
    cases = json.load(open(os.path.join(GOLD, "alloc_sequences.json")))
    n_ext = 0
    for case in cases:
        a = _make(case)
        ps = case["page_size"]
        for step, ent in enumerate(case["log"]):
            tag = (ps, case["need_sort"], step, ent["op"])
            if ent["op"] == "alloc_extend" or (ent["op"] == "alloc" and ps == 1 and ent["need"] > 0):
                if ent["op"] == "alloc":
                    pre_l, seq_l, last = [0, 0], [ent["need"] // 2, ent["need"] - ent["need"] // 2], [-1, -1]
                else:
                    pre_l, seq_l, last = ent["prefix_lens"], ent["seq_lens"], ent["last_loc"]
                bs = len(pre_l)
                r2t = torch.full((bs + 1, max(seq_l) + 3), -7, dtype=torch.int32, device=DEV)
                prefixes = []
                for p, ll in zip(pre_l, last):
                    t = torch.arange(900000, 900000 + p, dtype=torch.int64, device=DEV)
                    if p:
                        t[-1] = ll
                    prefixes.append(t)
                table = torch.tensor([list(range(1, bs + 1)), pre_l, seq_l, [t.data_ptr() if t.numel() else 0 for t in prefixes]],
                                     dtype=torch.int64)
                got = a.alloc_extend_rows(r2t, table, int(sum(seq_l) - sum(pre_l)))
                assert (got is None) == (ent["out"] is None), tag
                if got is not None:
                    assert got.dtype == torch.int64 and got.tolist() == ent["out"], tag
                    off = 0
                    for i, (p, s) in enumerate(zip(pre_l, seq_l)):
                        row = r2t[i + 1].tolist()
                        assert row[:p] == [int(x) for x in prefixes[i].tolist()], tag
                        assert row[p:s] == ent["out"][off: off + s - p], tag
                        assert all(x == -7 for x in row[s:]), tag
                        off += s - p
                    assert r2t[0].eq(-7).all()
                    n_ext += 1
            else:
                got = _replay(a, ent)
                if got != "void":
                    assert (got is None) == (ent["out"] is None), tag
                    if got is not None:
                        assert got.tolist() == ent["out"], tag
            assert a.free_pages.tolist() == ent["free"][0], tag
            assert a.release_pages.tolist() == ent["free"][1], tag
    assert n_ext > 40, n_ext
