"""End-to-end RadixAttention flow on the GPU: radix tree (native) -> shared req_to_token prefixes
-> paged allocation (HIP kernels) -> KV store -> extend with a radix hit -> decode, compared with
a full recompute by the oracle; plus the conservation law of the reference's strict leak check
(available + evictable + protected == pool size)."""
import numpy as np
import pytest
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bits(t):
    return t.detach().cpu().contiguous().view(torch.uint16).numpy()


class _World:
    def __init__(self, page_size, hq=8, hkv=2, d=128, size=4096):
        from sglang_amd.attention.backend import HipRadixAttnBackend
        from sglang_amd.attention.radix_attention import RadixAttention
        from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
        from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool
        from sglang_amd.mem_cache.radix_cache import RadixCache

        self.ps, self.hq, self.hkv, self.d, self.size = page_size, hq, hkv, d, size
        self.pool = MHATokenToKVPool(size, page_size, torch.bfloat16, hkv, d, 1, DEV)
        self.r2t = ReqToTokenPool(8, 1024, DEV)
        self.alloc = (TokenToKVPoolAllocator(size, torch.bfloat16, DEV, self.pool) if page_size == 1 else
                      PagedTokenToKVPoolAllocator(size, page_size, torch.bfloat16, DEV, self.pool))
        self.tree = RadixCache(self.r2t, self.alloc, page_size)

        class MC:
            num_attention_heads, num_key_value_heads, context_len = hq, hkv, 1024

        class MR:
            device = DEV
            req_to_token_pool = self.r2t
            token_to_kv_pool = self.pool
            token_to_kv_pool_allocator = self.alloc
            model_config = MC
            page_size = self.ps

        self.backend = HipRadixAttnBackend(MR)
        self.layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
        self.gen = torch.Generator().manual_seed(3)
        # "model": deterministic per-token K/V/Q so a shared prefix really has identical KV
        self.emb_k = torch.randn(50, hkv * d, generator=self.gen).to(torch.bfloat16).to(DEV)
        self.emb_v = torch.randn(50, hkv * d, generator=self.gen).to(torch.bfloat16).to(DEV)
        self.emb_q = torch.randn(50, hq * d, generator=self.gen).to(torch.bfloat16).to(DEV)

    def conservation(self):
        return self.alloc.available_size() + self.tree.evictable_size() + self.tree.protected_size()

    def prefill(self, req):
        """schedule: match_prefix -> lock -> alloc_for_extend -> forward_extend -> cache_unfinished_req"""
        from sglang_amd.forward_batch import ForwardBatch
        from sglang_amd.mem_cache.allocation import alloc_for_extend
        from sglang_amd.mem_cache.radix_cache import MatchPrefixParams, RadixKey

        toks = req.origin_input_ids
        m = self.tree.match_prefix(MatchPrefixParams(RadixKey(toks[:-1], req.extra_key)))  # keep >=1 token to run
        req.prefix_indices, req.last_node = m.device_indices, m.last_device_node
        req.cache_protected_len = len(m.device_indices)
        self.tree.inc_lock_ref(req.last_node)
        pre, seq = len(req.prefix_indices), len(toks)
        loc, rpi = alloc_for_extend([req], [pre], [seq], self.r2t, self.alloc, self.tree)
        ids = torch.tensor(toks[pre:], device=DEV)
        q, k, v = self.emb_q[ids], self.emb_k[ids], self.emb_v[ids]
        fb = ForwardBatch.for_extend(rpi, torch.tensor([seq], device=DEV), loc, [pre], [seq - pre])
        self.backend.init_forward_metadata(fb)
        o = self.layer(q, k, v, fb, self.backend)
        self.tree.cache_unfinished_req(req)
        return o, pre

    def decode(self, req, new_tok):
        from sglang_amd.forward_batch import ForwardBatch
        from sglang_amd.mem_cache.allocation import alloc_for_decode

        cur = len(req.origin_input_ids) + len(req.output_ids)
        rpi = torch.tensor([req.req_pool_idx], dtype=torch.int64, device=DEV)
        seq = torch.tensor([cur], dtype=torch.int64)
        loc = alloc_for_decode(rpi, seq.to(DEV), seq, self.r2t, self.alloc, self.tree)
        req.output_ids.append(new_tok)
        ids = torch.tensor([new_tok], device=DEV)
        fb = ForwardBatch.for_decode(rpi, (seq + 1).to(DEV), loc, seq + 1)
        self.backend.init_forward_metadata(fb)
        return self.layer(self.emb_q[ids], self.emb_k[ids], self.emb_v[ids], fb, self.backend)

    def reference_out(self, toks, q_positions):
        """Full recompute of causal attention over the token list (no cache) for given positions."""
        ids = np.array(toks)
        k = _bits(self.emb_k.cpu()[ids].view(len(ids), self.hkv, self.d))
        v = _bits(self.emb_v.cpu()[ids].view(len(ids), self.hkv, self.d))
        q = _bits(self.emb_q.cpu()[ids].view(len(ids), self.hq, self.d))
        qo = np.array([0, len(ids)], dtype=np.int64)
        o = orc.extend_attention(q, k, v, k, v, qo, np.array([0, 0], dtype=np.int32), np.zeros(0, dtype=np.int64),
                                 sm_scale=self.d ** -0.5)
        return o[q_positions]


@pytest.mark.parametrize("page_size", [1, 16])
def test_radix_hit_shares_pages_and_matches_full_recompute(page_size):
    from sglang_amd.mem_cache.radix_cache import Req

    w = _World(page_size)
    total = w.conservation()
    rng = np.random.default_rng(1)
    sys_prompt = [int(x) for x in rng.integers(0, 50, size=48)]
    a = Req(origin_input_ids=sys_prompt + [int(x) for x in rng.integers(0, 50, size=21)], output_ids=[])
    b = Req(origin_input_ids=sys_prompt + [int(x) for x in rng.integers(0, 50, size=37)], output_ids=[])

    oa, pre_a = w.prefill(a)
    assert pre_a == 0
    want = w.reference_out(a.origin_input_ids, slice(0, None))
    assert np.abs(oa.view(-1, w.hq, w.d).float().cpu().numpy() - want).max() < 2e-2

    ob, pre_b = w.prefill(b)  # radix hit on the shared system prompt
    assert pre_b == 48, pre_b
    r2t = w.r2t.req_to_token
    assert torch.equal(r2t[a.req_pool_idx, :48], r2t[b.req_pool_idx, :48])  # the SAME physical pages
    assert not torch.equal(r2t[a.req_pool_idx, 48:60], r2t[b.req_pool_idx, 48:60])
    want_b = w.reference_out(b.origin_input_ids, slice(48, None))
    assert np.abs(ob.view(-1, w.hq, w.d).float().cpu().numpy() - want_b).max() < 2e-2

    # a few decode steps on both, each checked against a full recompute
    for step in range(3):
        for req in (a, b):
            tok = int(rng.integers(0, 50))
            o = w.decode(req, tok)
            full = req.origin_input_ids + req.output_ids
            want_d = w.reference_out(full, slice(len(full) - 1, len(full)))
            assert np.abs(o.view(1, w.hq, w.d).float().cpu().numpy() - want_d).max() < 2e-2
    assert w.pool.check_errors() == 0

    # finish both: KV goes to the tree, duplicates and unaligned tails go back to the allocator
    for req in (a, b):
        kv_len = len(req.origin_input_ids) + len(req.output_ids)
        w.tree.cache_finished_req(req, kv_len_to_handle=kv_len)
        w.r2t.free(req.req_pool_idx)
    assert w.tree.protected_size() == 0
    assert w.conservation() == total, "KV slots leaked"
    # a third request re-using A's whole prompt hits everything cached (page aligned)
    c = Req(origin_input_ids=list(a.origin_input_ids) + [7], output_ids=[])
    _, pre_c = w.prefill(c)
    assert pre_c == (len(a.origin_input_ids) // page_size) * page_size


def test_eviction_frees_pages_under_pressure():
    from sglang_amd.mem_cache.radix_cache import Req

    w = _World(16, size=512)  # 32 pages only
    total = w.conservation()
    rng = np.random.default_rng(2)
    for i in range(12):  # 12 x 100 tokens >> 512: older finished prompts must be evicted
        r = Req(origin_input_ids=[int(x) for x in rng.integers(0, 50, size=100)], output_ids=[])
        w.prefill(r)
        w.tree.cache_finished_req(r, kv_len_to_handle=100)
        w.r2t.free(r.req_pool_idx)
        assert w.conservation() == total
    assert w.tree.evictable_size() <= 512 and w.pool.check_errors() == 0


def test_scheduler_flow_replays_reference_log_bit_exact(golden_dir):
    """SURVEY a6 + a16's request hooks against a log RECORDED FROM THE REFERENCE (tests/golden/make_golden.py::gen_flow:
    its RadixCache.cache_unfinished_req / cache_finished_req / match_prefix / evict, its CPU allocators and a
    req_to_token table driven like the scheduler drives them).  Replayed here through alloc_for_extend /
    alloc_for_decode (one fused kernel), the native radix tree's rx_radix_cache_req and the device-resident allocator:
    after EVERY op the request's row, its cache_protected_len / prefix_indices, both allocator lists and the tree's
    evictable / protected / total sizes are identical."""
    import json
    import os

    from sglang_amd.mem_cache.allocation import alloc_for_decode, alloc_for_extend
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
    from sglang_amd.mem_cache.memory_pool import ReqToTokenPool
    from sglang_amd.mem_cache.radix_cache import EvictParams, MatchPrefixParams, RadixCache, RadixKey, Req

    cases = json.load(open(os.path.join(golden_dir, "scheduler_flow.json")))
    n_ops = 0
    for case in cases:
        ps = case["page_size"]
        alloc = (TokenToKVPoolAllocator(case["size"], torch.bfloat16, DEV) if ps == 1 else
                 PagedTokenToKVPoolAllocator(case["size"], ps, torch.bfloat16, DEV))
        pool = ReqToTokenPool(case["rows"], case["ctx"], DEV)
        tree = RadixCache(pool, alloc, ps)
        live = {}

        def check(ent, req=None):
            a = ent["after"]
            tag = (ps, ent["op"], ent.get("rid"))
            assert alloc.free_pages.tolist() == a["free"], tag
            assert alloc.release_pages.tolist() == a["release"], tag
            assert [tree.evictable_size(), tree.protected_size(), tree.total_size()] == a["sizes"], tag
            if req is not None:
                n = len(req.get_fill_ids())
                assert pool.req_to_token[req.req_pool_idx, :n].tolist() == a["row"], tag
                assert req.cache_protected_len == a["protected"], tag
                assert (None if req.prefix_indices is None else req.prefix_indices.tolist()) == a["prefix_indices"], tag

        for ent in case["log"]:
            op = ent["op"]
            if op == "new":
                toks = ent["tokens"]
                req = Req(origin_input_ids=list(toks), output_ids=[], req_pool_idx=ent["row"])
                m = tree.match_prefix(MatchPrefixParams(key=RadixKey(toks, None)))
                req.prefix_indices, req.last_node = m.device_indices, m.last_device_node
                tree.inc_lock_ref(req.last_node)
                pre = len(req.prefix_indices)
                assert pre == ent["matched"]
                req.cache_protected_len = pre
                out, _ = alloc_for_extend([req], [pre], [len(toks)], pool, alloc, tree)
                assert out.tolist() == ent["out"], (ps, "new", ent["rid"])
                tree.cache_unfinished_req(req)
                live[ent["rid"]] = req
                check(ent, req)
            elif op == "decode":
                req = live[ent["rid"]]
                n = len(req.get_fill_ids())
                rpi = torch.tensor([req.req_pool_idx], dtype=torch.int64, device=DEV)
                seq = torch.tensor([n], dtype=torch.int64)
                loc = alloc_for_decode(rpi, seq.to(DEV), seq, pool, alloc, tree)
                assert loc.tolist() == ent["loc"], (ps, "decode", ent["rid"])
                req.output_ids.append(ent["token"])
                check(ent, req)
            elif op == "recache":
                req = live[ent["rid"]]
                tree.cache_unfinished_req(req)
                check(ent, req)
            elif op == "finish":
                req = live.pop(ent["rid"])
                tree.cache_finished_req(req, is_insert=ent["insert"], kv_len_to_handle=len(req.get_fill_ids()))
                check(ent)
            elif op == "evict":
                r = tree.evict(EvictParams(num_tokens=ent["num_tokens"]))
                assert r.num_tokens_evicted == ent["evicted"]
                check(ent)
            n_ops += 1
    assert n_ops > 200
