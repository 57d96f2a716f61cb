"""Radix tree: the oracle restatement and the native tree (libradix_hip.so, host only -- runs
without a GPU) replay op logs recorded from the reference's RadixCache
(tests/golden/radix_sequences.json, made by make_golden.py f7) and must reproduce every matched
index run, prefix length, lock delta, evicted segment (order included) and size counter.
Plus the behaviours the reference's unit tests name (test_radix_cache_unit.py:324-770)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.radix_oracle import RadixTreeOracle


def _load(golden_dir):
    with open(os.path.join(golden_dir, "radix_sequences.json")) as f:
        return json.load(f)


def test_oracle_radix_tree_matches_reference_logs(golden_dir):
    cases = _load(golden_dir)
    assert len(cases) == 9
    for case in cases:
        t = RadixTreeOracle(case["page_size"], case["policy"])
        held = {}
        for i, e in enumerate(case["log"]):
            tag = (case["policy"], case["page_size"], i, e["op"])
            if e["op"] == "insert":
                pl, _ = t.insert(e["tokens"], e["values"], e["extra"], e["priority"], e["chunked"])
                assert pl == e["prefix_len"], tag
            elif e["op"] == "match":
                idx, node = t.match_prefix(e["tokens"], e["extra"])
                assert idx == e["indices"], tag
                assert (node is t.root) == e["last_is_root"] and len(node.key) == e["last_key_len"], tag
            elif e["op"] == "lock":
                idx, node = t.match_prefix(e["tokens"], e["extra"])
                assert idx == e["indices"], tag
                assert t.inc_lock_ref(node) == e["delta"], tag
                held[e["handle"]] = node
            elif e["op"] == "unlock":
                assert t.dec_lock_ref(held.pop(e["handle"])) == e["delta"], tag
            elif e["op"] == "evict":
                n, segs = t.evict(e["num_tokens"])
                assert n == e["evicted"] and segs == e["segments"], tag
            assert [t.evictable_size_, t.protected_size_, t.total_size()] == e["sizes"], tag


class _RecAlloc:
    device = "cpu"

    def __init__(self):
        self.freed = []

    def free_segment(self, idx, *, start_pos):
        self.freed.append(idx.tolist())

    def free_segments(self, segments):
        for idx, start in segments:
            if idx.numel():
                self.freed.append(idx.tolist())


def test_native_radix_tree_matches_reference_logs(golden_dir):
    from sglang_amd.mem_cache.radix_cache import (EvictParams, InsertParams, MatchPrefixParams,
                                                  RadixCache, RadixKey)

    for case in _load(golden_dir):
        alloc = _RecAlloc()
        c = RadixCache(None, alloc, case["page_size"], eviction_policy=case["policy"])
        held = {}
        for i, e in enumerate(case["log"]):
            tag = (case["policy"], case["page_size"], i, e["op"])
            if e["op"] == "insert":
                r = c.insert(InsertParams(RadixKey(e["tokens"], e["extra"]),
                                          torch.tensor(e["values"], dtype=torch.int64),
                                          e["priority"], e["chunked"]))
                assert r.prefix_len == e["prefix_len"], tag
            elif e["op"] in ("match", "lock"):
                m = c.match_prefix(MatchPrefixParams(RadixKey(e["tokens"], e["extra"])))
                assert m.device_indices.tolist() == e["indices"], tag
                assert m.device_indices.dtype == torch.int64
                if e["op"] == "match":
                    assert (m.last_device_node == c.root_node) == e["last_is_root"], tag
                    assert m.last_device_node.key_len() == e["last_key_len"], tag
                else:
                    assert c.inc_lock_ref(m.last_device_node).delta == e["delta"], tag
                    held[e["handle"]] = m.last_device_node
            elif e["op"] == "unlock":
                assert c.dec_lock_ref(held.pop(e["handle"])).delta == e["delta"], tag
            elif e["op"] == "evict":
                alloc.freed = []
                r = c.evict(EvictParams(e["num_tokens"]))
                assert r.num_tokens_evicted == e["evicted"] and alloc.freed == e["segments"], tag
            assert [c.evictable_size(), c.protected_size(), c.total_size()] == e["sizes"], tag


def test_named_behaviours():
    from sglang_amd.mem_cache.radix_cache import (EvictParams, InsertParams, MatchPrefixParams,
                                                  RadixCache, RadixKey)

    c = RadixCache.create_simulated(page_size=1)
    assert c.insert(InsertParams(RadixKey([1, 2, 3, 4]))).prefix_len == 0
    m = c.match_prefix(MatchPrefixParams(RadixKey([1, 2, 3, 9])))  # ends inside a node -> split
    assert m.device_indices.tolist() == [1, 2, 3] and c.num_nodes() == 3
    assert c.insert(InsertParams(RadixKey([1, 2, 3, 4, 5]))).prefix_len == 4
    # extra_key namespaces never share nodes
    assert c.match_prefix(MatchPrefixParams(RadixKey([1, 2, 3], "lora"))).device_indices.numel() == 0
    # empty key / disabled cache
    assert c.match_prefix(MatchPrefixParams(RadixKey([]))).last_device_node == c.root_node
    d = RadixCache.create_simulated(disable=True)
    assert d.insert(InsertParams(RadixKey([1, 2]))).prefix_len == 0 and d.total_size() == 0
    # lock protects from eviction; unlock makes evictable again
    node = c.match_prefix(MatchPrefixParams(RadixKey([1, 2, 3, 4, 5]))).last_device_node
    assert c.inc_lock_ref(node).delta == -5 and c.protected_size() == 5 and c.evictable_size() == 0
    assert c.evict(EvictParams(100)).num_tokens_evicted == 0
    assert c.dec_lock_ref(node).delta == 5
    assert c.evict(EvictParams(100)).num_tokens_evicted == 5 and c.total_size() == 0
    with pytest.raises(KeyError):
        c.inc_lock_ref(node)  # the node was evicted
    # page alignment: keys and matches truncate to whole pages
    p = RadixCache.create_simulated(page_size=4)
    assert p.insert(InsertParams(RadixKey(list(range(10))))).prefix_len == 0 and p.total_size() == 8
    assert p.match_prefix(MatchPrefixParams(RadixKey(list(range(7))))).device_indices.tolist() == [0, 1, 2, 3]
    c.reset()
    assert c.total_size() == 0 and c.num_nodes() == 1
    with pytest.raises(ValueError):
        RadixCache(eviction_policy="nope")
