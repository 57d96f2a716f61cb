"""Backend-level GPU tests: HipRadixAttnBackend driven the way SGLang's runner drives a backend
(mock ModelRunner with real pools, `init_forward_metadata` + `RadixAttention.forward`), over the
dense-attention case matrix of the reference's kit
(python/sglang/test/kits/attention_unittest/attention_methods/dense_attention.py:102-215: page 1/16/32,
zero-prefix / exact-page / cross-page / ragged, decode at page boundaries, GQA 4/2, MQA 4/1) and its
slot layouts (:593-712 contiguous / shuffled_pages / interleaved_pages).  Slots come from OUR
allocators (HIP alloc kernels), req_to_token rows from rx_write_req_to_token, KV from rx_store_kv;
expected outputs from the oracle's torch-native semantics (a14)."""
import numpy as np
import pytest

import parity_util as parity
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [
    # name, mode, page, prefix_lens, extend_lens, Hq, Hkv
    ("mha_extend_page_size_1", "extend", 1, (2, 4), (3, 1), 4, 4),
    ("mha_extend_zero_prefix_exact_page", "extend", 16, (0,), (16,), 4, 4),
    ("mha_extend_zero_prefix_input_page_edges", "extend", 16, (0, 0, 0), (15, 16, 17), 4, 4),
    ("mha_extend_prefix_exact_page", "extend", 16, (16,), (2,), 4, 4),
    ("mha_extend_total_exact_page", "extend", 16, (8,), (8,), 4, 4),
    ("mha_extend_cross_page_boundary", "extend", 16, (15,), (2,), 4, 4),
    ("mha_extend_ragged_page_boundary", "extend", 16, (0, 8, 16), (15, 8, 1), 4, 4),
    ("mha_extend_page32_cross_boundary", "extend", 32, (31,), (2,), 4, 4),
    ("mha_decode_page_boundary", "decode", 16, (14, 15, 16), None, 4, 4),
    ("mha_decode_bsz1_nonzero_prefix", "decode", 16, (7,), None, 4, 4),
    ("gqa_decode_page_boundary", "decode", 16, (14, 15, 16), None, 4, 2),
    ("mqa_extend_total_exact_page", "extend", 16, (8,), (8,), 4, 1),
    # larger than the kit: long ragged batch on the MFMA path
    ("gqa_extend_long_ragged", "extend", 16, (100, 0, 513), (260, 129, 31), 8, 2),
    ("gqa_decode_long_ragged", "decode", 16, (1000, 1, 300, 4095), None, 32, 8),
]


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


class _Harness:
    def __init__(self, page_size, hq, hkv, d, dtype, layout, index_mode, max_ctx=4200, max_reqs=8,
                 split_policy="native", size=8192, server_args_extra=None):
        from sglang_amd.attention.backend import HipRadixAttnBackend
        from sglang_amd.attention.radix_attention import RadixAttention
        from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator, TokenToKVPoolAllocator
        from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

        self.ps, self.hq, self.hkv, self.d, self.dtype = page_size, hq, hkv, d, dtype
        self.pool = MHATokenToKVPool(size, page_size, dtype, hkv, d, 1, DEV)
        self.r2t = ReqToTokenPool(max_reqs, max_ctx, DEV)
        if page_size == 1:
            self.alloc = TokenToKVPoolAllocator(size, dtype, DEV, self.pool)
        else:
            self.alloc = PagedTokenToKVPoolAllocator(size, page_size, dtype, DEV, self.pool, debug_mode=True)
        g = torch.Generator().manual_seed(5)
        n = len(self.alloc.free_pages)
        if layout == "shuffled_pages":
            self.alloc.free_pages = self.alloc.free_pages[torch.randperm(n, generator=g).to(DEV)]
        elif layout == "interleaved_pages":
            fp = self.alloc.free_pages
            self.alloc.free_pages = torch.cat((fp[0::2], fp[1::2]))

        class MC:
            num_attention_heads, num_key_value_heads, context_len = hq, hkv, max_ctx

        class MR:
            device = DEV
            req_to_token_pool = self.r2t
            token_to_kv_pool = self.pool
            token_to_kv_pool_allocator = self.alloc
            model_config = MC
            page_size = self.ps

            class server_args:
                triton_attention_num_kv_splits = 8

        for key, val in (server_args_extra or {}).items():
            setattr(MR.server_args, key, val)
        self.backend = HipRadixAttnBackend(MR, decode_index_mode=index_mode, split_policy=split_policy)
        self.layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
        self.gen = torch.Generator().manual_seed(11)

    def rand(self, *shape):
        return torch.randn(*shape, generator=self.gen).to(self.dtype).to(DEV)

    def alloc_extend(self, rows, prefix_lens, seq_lens):
        """alloc_for_extend (srt/mem_cache/allocation.py:303-403): slots + req_to_token rows."""
        from sglang_amd import ops

        pre = torch.tensor(prefix_lens, dtype=torch.int64)
        seq = torch.tensor(seq_lens, dtype=torch.int64)
        ext = seq - pre
        if self.ps == 1:
            out = self.alloc.alloc(int(ext.sum()))
        else:
            last = torch.tensor([int(self.r2t.req_to_token[r, p - 1]) if p > 0 else -1
                                 for r, p in zip(rows, prefix_lens)], dtype=torch.int64, device=DEV)
            out = self.alloc.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, last, int(ext.sum()))
        assert out is not None
        ops.write_req_to_token(self.r2t.req_to_token, torch.tensor(rows, dtype=torch.int64, device=DEV),
                               None, pre.to(DEV), seq.to(DEV), ext.to(DEV), out)
        return out

    def fill_prefix(self, rows, prefix_lens):
        if sum(prefix_lens) == 0:
            return
        loc = self.alloc_extend(rows, [0] * len(rows), list(prefix_lens))
        k = self.rand(sum(prefix_lens), self.hkv, self.d)
        v = self.rand(sum(prefix_lens), self.hkv, self.d)
        self.pool.set_kv_buffer(self.layer, loc, k, v)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("d", [16, 96, 128, 256])
@pytest.mark.parametrize("layout", ["contiguous", "shuffled_pages", "interleaved_pages"])
def test_dense_case_matrix(case, d, layout):
    from sglang_amd.forward_batch import ForwardBatch

    name, mode, ps, prefix_lens, extend_lens, hq, hkv = case
    if ps == 1 and layout != "contiguous":
        pytest.skip("page layouts need page_size > 1")
    dtype = torch.float16
    # decode: both index modes, and both split schedules (the MI355X-native one and the reference's K3)
    variants = ([("paged", "native"), ("indices", "native"), ("paged", "reference")] if mode == "decode"
                else [("paged", "native")])
    for index_mode, policy in variants:
        hs = _Harness(ps, hq, hkv, d, dtype, layout, index_mode, split_policy=policy)
        bs = len(prefix_lens)
        rows = hs.r2t.alloc(bs)
        hs.fill_prefix(rows, prefix_lens)
        rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
        if mode == "extend":
            seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
            loc = hs.alloc_extend(rows, list(prefix_lens), seq_lens)
            T = sum(extend_lens)
            q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)
            fb = ForwardBatch.for_extend(rpi, torch.tensor(seq_lens, device=DEV), loc, list(prefix_lens),
                                         list(extend_lens))
            hs.backend.init_forward_metadata(fb)
            o = hs.layer(q, k, v, fb, hs.backend)
            kb, vb = hs.pool.get_kv_buffer(0)
            want = orc.sdpa_extend_req_to_token(
                _bits(q.view(T, hq, d)), _bits(kb), _bits(vb), _bits(hs.r2t.req_to_token), np.array(rows),
                np.array(seq_lens), np.array(prefix_lens), np.array(extend_lens), d ** -0.5)
            got = _bits(o.view(T, hq, d)).astype(np.float64)
        else:
            seq_lens = [p + 1 for p in prefix_lens]
            seq_t = torch.tensor(seq_lens, dtype=torch.int64)
            if ps == 1:
                loc = hs.alloc.alloc(bs)
            else:
                last = torch.tensor([int(hs.r2t.req_to_token[r, p - 1]) for r, p in zip(rows, prefix_lens)],
                                    dtype=torch.int64, device=DEV)
                loc = hs.alloc.alloc_decode(seq_t.to(DEV), seq_t, last)
            # alloc_for_decode writes req_to_token[req, seq_len-1] (allocation.py:578-580)
            hs.r2t.req_to_token[rpi, torch.tensor(prefix_lens, device=DEV)] = loc.to(torch.int32)
            q, k, v = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
            fb = ForwardBatch.for_decode(rpi, seq_t.to(DEV), loc, seq_t)
            hs.backend.init_forward_metadata(fb)
            o = hs.layer(q, k, v, fb, hs.backend)
            kb, vb = hs.pool.get_kv_buffer(0)
            want = orc.sdpa_decode_req_to_token(_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb),
                                                _bits(hs.r2t.req_to_token), np.array(rows),
                                                np.array(seq_lens), d ** -0.5)
            got = _bits(o.view(bs, hq, d)).astype(np.float64)
        assert hs.pool.check_errors() == 0
        # north-star bound (1e-3 for fp16 outputs below 2, one ulp above); the kit's own tolerance is 3e-2
        # (dense_attention.py:35-36)
        parity.check_out(got, want, dtype, (name, index_mode, policy))


def test_idle_mode_and_graph_state():
    from sglang_amd.forward_batch import ForwardBatch, ForwardMode

    hs = _Harness(16, 8, 2, 128, torch.bfloat16, "contiguous", "paged")
    fb = ForwardBatch(forward_mode=ForwardMode.IDLE, batch_size=0,
                      req_pool_indices=torch.zeros(0, dtype=torch.int64, device=DEV),
                      seq_lens=torch.zeros(0, dtype=torch.int64, device=DEV), out_cache_loc=None)
    hs.backend.init_forward_metadata(fb)
    q = torch.zeros(0, 8 * 128, dtype=torch.bfloat16, device=DEV)
    assert hs.backend.forward(q, None, None, hs.layer, fb).shape == (0, 8 * 128)
    hs.backend.init_cuda_graph_state(4, 4)
    assert hs.backend.get_cuda_graph_seq_len_fill_value() == 1


def test_decode_step_is_hip_graph_capturable():
    """No host sync / allocation inside the C ABI: a decode layer replays under a HIP graph."""
    from sglang_amd.forward_batch import ForwardBatch

    hs = _Harness(16, 32, 8, 128, torch.bfloat16, "shuffled_pages", "paged")
    prefix = (600, 33, 1024)
    rows = hs.r2t.alloc(3)
    hs.fill_prefix(rows, prefix)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor([p + 1 for p in prefix], dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, p - 1]) for r, p in zip(rows, prefix)],
                        dtype=torch.int64, device=DEV)
    loc = hs.alloc.alloc_decode(seq.to(DEV), seq, last)
    hs.r2t.req_to_token[rpi, torch.tensor(prefix, device=DEV)] = loc.to(torch.int32)
    q, k, v = hs.rand(3, 32 * 128), hs.rand(3, 8 * 128), hs.rand(3, 8 * 128)
    fb = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
    hs.backend.init_forward_metadata(fb)
    eager = hs.layer(q, k, v, fb, hs.backend).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        hs.layer(q, k, v, fb, hs.backend)  # warm allocator
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = hs.layer(q, k, v, fb, hs.backend)
    q.copy_(hs.rand(3, 32 * 128))  # new input, same addresses
    graph.replay()
    torch.cuda.synchronize()
    fresh = hs.layer(q, k, v, fb, hs.backend)
    assert torch.equal(out, fresh) and not torch.equal(out, eager)


def test_mla_latent_pool_and_decode():
    """Config-5-shaped MLA decode (absorbed form): Hq=16 per GPU, Hkv=1, Dk=576 (512 latent + 64
    rope), Dv=512 = first 512 columns of the SAME rows (triton_backend.py:1739-1757, memory_pool.py:
    3906-4179), through MLATokenToKVPool + HipRadixAttnBackend (generic HIP decode kernel)."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool

    hq, rank, rope = 16, 512, 64
    lens = [70, 129, 5]
    bs = len(lens)
    pool = MLATokenToKVPool(1024, 1, torch.bfloat16, rank, rope, 1, DEV)
    r2t = ReqToTokenPool(4, 256, DEV)
    g = torch.Generator().manual_seed(0)
    perm = torch.randperm(1023, generator=g) + 1
    rows = r2t.alloc(bs)
    layer = RadixAttention(hq, rank + rope, (128 + 64) ** -0.5, 1, 0, v_head_dim=rank)
    off = 0
    for r, n in zip(rows, lens):
        slots = perm[off: off + n].to(DEV); off += n
        r2t.req_to_token[r, :n] = slots.int()
        nope = torch.randn(n, 1, rank, generator=g).to(torch.bfloat16).to(DEV)
        rp = torch.randn(n, 1, rope, generator=g).to(torch.bfloat16).to(DEV)
        pool.set_mla_kv_buffer(layer, slots, nope, rp)
        back_n, back_r = pool.get_mla_kv_buffer(layer, slots)
        assert torch.equal(back_n, nope) and torch.equal(back_r, rp)

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, 1, 256

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        model_config = MC
        page_size = 1

    be = HipRadixAttnBackend(MR)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor(lens, dtype=torch.int64)
    # the new token's latent row goes in through set_kv_buffer (whole 576-wide row)
    loc = torch.tensor([int(r2t.req_to_token[r, n - 1]) for r, n in zip(rows, lens)], device=DEV)
    k_new = torch.randn(bs, 1, rank + rope, generator=g).to(torch.bfloat16).to(DEV)
    q = torch.randn(bs, hq * (rank + rope), generator=g).to(torch.bfloat16).to(DEV)
    fb = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
    be.init_forward_metadata(fb)
    o = layer(q, k_new, k_new[..., :rank], fb, be)
    assert o.shape == (bs, hq * rank)
    kb = pool.get_key_buffer(0)
    want, absw = parity.want_and_absw(orc.sdpa_decode_req_to_token, (_bits(q.view(bs, hq, rank + rope)), _bits(kb),
                                                                     _bits(kb[..., :rank].contiguous()), _bits(r2t.req_to_token),
                                                                     np.array(rows), np.array(lens), layer.scaling), (2,))
    got = o.view(bs, hq, rank).float().cpu().numpy()
    parity.check_out(got, want, o.dtype, "mla latent pool decode", ulps=1, absw=absw)


@pytest.mark.parametrize("own_v", [False, True, "flag"], ids=["v_view_of_k", "v_own_tensor", "v_own_tensor_flagged_as_latent_prefix"])
def test_mla_latent_radix_hit_extend(own_v):
    """A radix-cache hit of an MLA model: cached latent prefix + new tokens through HipRadixAttnBackend.forward_extend
    (absorbed form, q 576 / v 512 over ONE kv head: forward_absorb_core -> attn_mqa -> forward_extend,
    triton_backend.py:1290-1437) -- rx::extend_mla_kernel.  The new tokens' v is either a view of their k rows or, as in
    the reference's model code, a tensor of its own (k is a fresh concat of k_nope and k_pe there)."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool

    hq, rank, rope = 16, 512, 64
    pre, ext = [70, 0, 300], [9, 40, 130]
    bs = len(pre)
    pool = MLATokenToKVPool(1024, 1, torch.bfloat16, rank, rope, 1, DEV)
    r2t = ReqToTokenPool(4, 512, DEV)
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(1023, generator=g) + 1
    rows = r2t.alloc(bs)
    layer = RadixAttention(hq, rank + rope, (128 + 64) ** -0.5, 1, 0, v_head_dim=rank)
    off = 0
    new_loc = []
    for r, p_, e_ in zip(rows, pre, ext):
        slots = perm[off: off + p_ + e_].to(DEV); off += p_ + e_
        r2t.req_to_token[r, : p_ + e_] = slots.int()
        if p_:
            pool.set_mla_kv_buffer(layer, slots[:p_], (torch.randn(p_, 1, rank, generator=g) * 0.5).to(torch.bfloat16).to(DEV),
                                   (torch.randn(p_, 1, rope, generator=g) * 0.5).to(torch.bfloat16).to(DEV))
        new_loc.append(slots[p_:])
    loc = torch.cat(new_loc)

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, 1, 512

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        model_config = MC
        page_size = 1

    be = HipRadixAttnBackend(MR, mla_v_is_latent_prefix=(own_v == "flag"))
    T = sum(ext)
    k_new = (torch.randn(T, 1, rank + rope, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    v_new = k_new[..., :rank].contiguous() if own_v else k_new[..., :rank]
    q = torch.randn(T, hq * (rank + rope), generator=g).to(torch.bfloat16).to(DEV)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    fb = ForwardBatch.for_extend(rpi, torch.tensor([p_ + e_ for p_, e_ in zip(pre, ext)], device=DEV), loc, pre, ext)
    be.init_forward_metadata(fb)
    o = layer(q, k_new, v_new, fb, be)
    assert o.shape == (T, hq * rank)
    # the new tokens were stored as well
    assert torch.equal(pool.get_key_buffer(0)[loc], k_new)
    kb = pool.get_key_buffer(0)
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.concatenate([_bits(r2t.req_to_token[r, :p_]).astype(np.int64) for r, p_ in zip(rows, pre)])
    kbn = kb.float().cpu().numpy()
    want = orc.extend_attention(q.view(T, hq, rank + rope).float().cpu().numpy(), k_new.float().cpu().numpy(),
                                k_new[..., :rank].float().cpu().numpy(), kbn, kbn[..., :rank], qo, kv_indptr, kv_indices,
                                sm_scale=layer.scaling)
    got = o.view(T, hq, rank).float().cpu().numpy()
    parity.check_out(got, want, torch.bfloat16, ("mla_radix_hit_extend", own_v))


def test_hnd_pool_store_and_decode():
    """HND pool [pages, Hkv, page, D] (memory_pool.py:2032-2036): rx_store_kv_layout scatter +
    paged decode through the backend equal the NHD result bit for bit."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

    hq, hkv, d, ps = 8, 2, 128, 16
    lens = [40, 17, 129]
    g = torch.Generator().manual_seed(0)
    outs = []
    for use_hnd in (False, True):
        pool = MHATokenToKVPool(512, ps, torch.bfloat16, hkv, d, 1, DEV, use_hnd=use_hnd)
        r2t = ReqToTokenPool(4, 256, DEV)
        rows = r2t.alloc(3)
        layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
        gg = torch.Generator().manual_seed(1)
        perm = torch.randperm(511, generator=gg) + 1
        off = 0
        for r, n in zip(rows, lens):
            slots = perm[off: off + n].to(DEV); off += n
            r2t.req_to_token[r, :n] = slots.int()
            k = torch.randn(n, hkv, d, generator=gg).to(torch.bfloat16).to(DEV)
            v = torch.randn(n, hkv, d, generator=gg).to(torch.bfloat16).to(DEV)
            pool.set_kv_buffer(layer, slots, k, v)
            if use_hnd:  # the scatter landed at [page, head, off, :]
                kb = pool.get_key_buffer(0)
                assert torch.equal(kb[slots // ps, :, slots % ps, :], k)

        class MC:
            num_attention_heads, num_key_value_heads, context_len = hq, hkv, 256

        class MR:
            device = DEV
            req_to_token_pool = r2t
            token_to_kv_pool = pool
            model_config = MC
            page_size = ps

        be = HipRadixAttnBackend(MR)
        rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
        seq = torch.tensor(lens, dtype=torch.int64)
        q = torch.randn(3, hq * d, generator=gg).to(torch.bfloat16).to(DEV)
        fb = ForwardBatch.for_decode(rpi, seq.to(DEV), None, seq)
        be.init_forward_metadata(fb)
        outs.append(be.forward_decode(q, None, None, layer, fb, save_kv_cache=False))
        assert pool.check_errors() == 0
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype,vd", [(torch.bfloat16, 128), (torch.float16, 64), (torch.float8_e4m3fn, 128)])
def test_move_kv_cache_on_hnd_and_nhd_pools_agree(dtype, vd):
    """move_kv_cache (memory_pool.py:2775-2842) on the HND layout [pages, Hkv, page, D] -- the pool layout the
    backend and the bench default to: every layer's K and V rows src -> tgt, identical to the NHD pool's move."""
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool

    hkv, d, ps, layers, size = 4, 128, 16, 3, 1024
    g = torch.Generator().manual_seed(3)
    src = (torch.randperm(size - 1, generator=g)[:200] + 1).to(DEV)
    tgt = (torch.randperm(size - 1, generator=g)[:200] + 1).to(DEV)
    tgt = tgt[~torch.isin(tgt, src)]              # a move never reads what it writes
    src = src[: tgt.numel()]
    pools = {}
    for use_hnd in (False, True):
        pool = pools[use_hnd] = MHATokenToKVPool(size, ps, dtype, hkv, d, layers, DEV, v_head_dim=vd, use_hnd=use_hnd)
        gg = torch.Generator(device=DEV).manual_seed(7)
        for l in range(layers):
            for bufs, hd in ((pool.k_buffer, d), (pool.v_buffer, vd)):
                rows = torch.randint(0, 120, (size + ps, hkv, hd), generator=gg, device=DEV, dtype=torch.uint8)
                rows = rows if pool.is_fp8 else rows.to(dtype)
                if use_hnd:
                    bufs[l].copy_(rows.view(-1, ps, hkv, hd).permute(0, 2, 1, 3))
                else:
                    bufs[l].copy_(rows)
        pool.move_kv_cache(tgt, src)
    torch.cuda.synchronize()
    for l in range(layers):
        for a, b in ((pools[False].k_buffer[l], pools[True].k_buffer[l]), (pools[False].v_buffer[l], pools[True].v_buffer[l])):
            hd = a.shape[-1]
            assert torch.equal(a.view(-1, ps, hkv, hd).permute(0, 2, 1, 3), b)
        kb = pools[True].k_buffer[l]
        assert torch.equal(kb[tgt // ps, :, tgt % ps, :], kb[src // ps, :, src % ps, :])


def test_hnd_pool_extend_with_prefix():
    """Extend (prefix gathered from an HND pool through the shift/mask page addressing)."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

    hq, hkv, d, ps = 8, 2, 128, 16
    pre, ext = [48, 0, 130], [20, 70, 3]
    outs = []
    for use_hnd in (False, True):
        pool = MHATokenToKVPool(1024, ps, torch.float16, hkv, d, 1, DEV, use_hnd=use_hnd)
        r2t = ReqToTokenPool(4, 512, DEV)
        rows = r2t.alloc(3)
        layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
        gg = torch.Generator().manual_seed(1)
        perm = torch.randperm(1023, generator=gg) + 1
        off, locs = 0, []
        for r, p, e in zip(rows, pre, ext):
            slots = perm[off: off + p + e].to(DEV); off += p + e
            r2t.req_to_token[r, : p + e] = slots.int()
            if p:
                pool.set_kv_buffer(layer, slots[:p], torch.randn(p, hkv, d, generator=gg).half().to(DEV),
                                   torch.randn(p, hkv, d, generator=gg).half().to(DEV))
            locs.append(slots[p:])

        class MC:
            num_attention_heads, num_key_value_heads, context_len = hq, hkv, 512

        class MR:
            device = DEV
            req_to_token_pool = r2t
            token_to_kv_pool = pool
            model_config = MC
            page_size = ps

        be = HipRadixAttnBackend(MR)
        T = sum(ext)
        q = torch.randn(T, hq * d, generator=gg).half().to(DEV)
        k = torch.randn(T, hkv * d, generator=gg).half().to(DEV)
        v = torch.randn(T, hkv * d, generator=gg).half().to(DEV)
        fb = ForwardBatch.for_extend(torch.tensor(rows, device=DEV), torch.tensor([p + e for p, e in zip(pre, ext)], device=DEV),
                                     torch.cat(locs), pre, ext)
        be.init_forward_metadata(fb)
        outs.append(layer(q, k, v, fb, be))
        assert pool.check_errors() == 0
    assert torch.equal(outs[0], outs[1])


def _swa_expected(q3, kb, vb, r2t, rows, seq_lens, prefix_lens, extend_lens, d, window, mask=None, abs_v=False):
    """Semantics of a sliding-window layer: the reference's window metadata (last min(prefix, W) prefix
    tokens, triton_backend.py:2043-2110) + the in-kernel window mask (extend_attention.py:391-397,550-556)."""
    bs = len(rows)
    pre = np.array(prefix_lens)
    wlen = np.minimum(pre, window)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, np.array(rows), wlen, kv_start=pre - wlen)
    T = int(sum(extend_lens))
    qo = np.concatenate([[0], np.cumsum(extend_lens)]).astype(np.int64)
    ke = np.zeros((T,) + kb.shape[1:], dtype=kb.dtype)
    ve = np.zeros((T,) + vb.shape[1:], dtype=vb.dtype)
    for i in range(bs):
        sl = r2t[rows[i], prefix_lens[i]: seq_lens[i]]
        ke[qo[i]: qo[i + 1]], ve[qo[i]: qo[i + 1]] = kb[sl], vb[sl]
    kw = dict(custom_mask=mask[0], mask_indptr=mask[1], window_kv_offsets=pre - wlen) if mask else {}
    if abs_v:  # the |V| twin of parity_util.check_out's absw term
        ve, vb = parity.abs_values(ve), parity.abs_values(vb)
    return orc.extend_attention(q3, ke, ve, kb, vb, qo, kv_indptr, kv_indices, is_causal=True, sm_scale=d ** -0.5,
                                sliding_window_size=window, **kw)


@pytest.mark.parametrize("index_mode", ["paged", "indices"])
def test_sliding_window_layers_decode_and_extend(index_mode):
    """Hybrid SWA model: window metadata (window_kv_indptr / indices / num_kv_splits / offsets) is built
    next to the full one and sliding-window layers read it (triton_backend.py:748-767,1353-1365,1770-1781)."""
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch

    ps, hq, hkv, d, W = 16, 8, 2, 128, 48
    hs = _Harness(ps, hq, hkv, d, torch.float16, "shuffled_pages", index_mode)
    hs.backend.sliding_window_size = W
    hs.backend.window_kv_indptr = torch.zeros_like(hs.backend.kv_indptr)
    swa_layer = RadixAttention(hq, d, d ** -0.5, hkv, 0, sliding_window_size=W)
    prefix_lens, extend_lens = (10, 100, 47, 300), (5, 70, 1, 33)
    bs = len(prefix_lens)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, prefix_lens)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
    loc = hs.alloc_extend(rows, list(prefix_lens), seq_lens)
    T = sum(extend_lens)
    q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)
    fb = ForwardBatch.for_extend(rpi, torch.tensor(seq_lens, device=DEV), loc, list(prefix_lens), list(extend_lens))
    hs.backend.init_forward_metadata(fb)
    assert hs.backend.forward_metadata.window_kv_indptr is not None
    o_swa = swa_layer(q, k, v, fb, hs.backend)
    o_full = hs.layer(q, k, v, fb, hs.backend, save_kv_cache=False)
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = _bits(hs.r2t.req_to_token)
    swa_args = (_bits(q.view(T, hq, d)), _bits(kb), _bits(vb), r2t, rows, seq_lens, prefix_lens, extend_lens, d, W)
    want, absw = _swa_expected(*swa_args), _swa_expected(*swa_args, abs_v=True)
    parity.check_out(o_swa.view(T, hq, d).float().cpu().numpy(), want, o_swa.dtype, "swa extend", ulps=1, absw=absw)
    want_full, absw_full = parity.want_and_absw(orc.sdpa_extend_req_to_token, (
        _bits(q.view(T, hq, d)), _bits(kb), _bits(vb), r2t, np.array(rows), np.array(seq_lens), np.array(prefix_lens),
        np.array(extend_lens), d ** -0.5), (2,))
    parity.check_out(o_full.view(T, hq, d).float().cpu().numpy(), want_full, o_full.dtype, "full-attention layer of the swa model", ulps=1,
                     absw=absw_full)
    # ---- decode: the window layer sees the last min(seq, W) tokens
    seq_t = torch.tensor([s + 1 for s in seq_lens], dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, s - 1]) for r, s in zip(rows, seq_lens)], dtype=torch.int64,
                        device=DEV)
    dloc = hs.alloc.alloc_decode(seq_t.to(DEV), seq_t, last)
    hs.r2t.req_to_token[rpi, torch.tensor(seq_lens, device=DEV)] = dloc.to(torch.int32)
    q1, k1, v1 = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    fbd = ForwardBatch.for_decode(rpi, seq_t.to(DEV), dloc, seq_t)
    hs.backend.init_forward_metadata(fbd)
    o1 = swa_layer(q1, k1, v1, fbd, hs.backend)
    kb, vb = hs.pool.get_kv_buffer(0)
    sl = seq_t.numpy()
    wl = np.minimum(sl, W)
    kv_indptr, kv_indices = orc.build_kv_indices(_bits(hs.r2t.req_to_token), np.array(rows), wl, kv_start=sl - wl)
    want1, absw1 = parity.want_and_absw(orc.decode_attention, (_bits(q1.view(bs, hq, d)), _bits(kb), _bits(vb), kv_indptr,
                                                               kv_indices, d ** -0.5), (2,))
    parity.check_out(o1.view(bs, hq, d).float().cpu().numpy(), want1, o1.dtype, "swa decode", ulps=1, absw=absw1)
    assert hs.pool.check_errors() == 0


def test_target_verify_mode_with_tree_mask():
    """TARGET_VERIFY (speculative decoding): every request extends by its draft tokens over its whole
    cached sequence under the draft tree's mask; metadata as triton_backend.py:801-866."""
    from sglang_amd.forward_batch import ForwardBatch, ForwardMode

    ps, hq, hkv, d, nd = 16, 8, 2, 128, 5
    hs = _Harness(ps, hq, hkv, d, torch.float16, "shuffled_pages", "paged")
    seq_lens = (70, 16, 333)
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
    md = hs.backend.forward_metadata
    assert md.max_extend_len == nd and md.custom_mask is Spec.custom_mask
    assert md.mask_indptr.tolist() == np.concatenate([[0], np.cumsum([nd * (s + nd) for s in seq_lens])]).tolist()
    o = hs.layer(q, k, v, fb, hs.backend)
    assert hs.backend._verify_split_on and hs.backend._verify_split.num_chunks(bs) >= 2  # small batch: split-KV verify
    kb, vb = hs.pool.get_kv_buffer(0)
    r2t = _bits(hs.r2t.req_to_token)
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, np.array(rows), np.array(seq_lens))
    qo = (np.arange(bs + 1) * nd).astype(np.int64)
    kbn, vbn = _bits(kb), _bits(vb)
    ke = np.concatenate([kbn[r2t[rows[i], seq_lens[i]: total[i]]] for i in range(bs)])
    ve = np.concatenate([vbn[r2t[rows[i], seq_lens[i]: total[i]]] for i in range(bs)])
    mi = np.concatenate([[0], np.cumsum([m.size for m in masks])]).astype(np.int64)
    want, absw = parity.want_and_absw(orc.extend_attention, (_bits(q.view(T, hq, d)), ke, ve, kbn, vbn, qo, kv_indptr, kv_indices),
                                      (2, 4), is_causal=True, sm_scale=d ** -0.5, custom_mask=cm, mask_indptr=mi)
    # (a small batch takes the split-KV verify path: 16-bit chunk partials merged by LSE -- two roundings, 2 ulp)
    parity.check_out(o.view(T, hq, d).float().cpu().numpy(), want, o.dtype, "target verify", ulps=2, absw=absw)


def test_native_split_schedule_values():
    """rx_num_kv_splits_native: S = ceil(CUs / (bs * Hkv * ceil(G/16))) capped, short requests fewer."""
    from sglang_amd import ops

    lens = torch.tensor([1, 100, 128, 129, 5000, 40000], dtype=torch.int64, device=DEV)
    out = torch.zeros(6, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits_native(out, lens, 32, 8, 32, 256)          # 6 * 8 = 48 workgroups -> ceil(256/48) = 6
    assert ops.native_max_kv_splits(6, 32, 8, 256, 32) == 6
    assert out.tolist() == [1, 1, 1, 2, 6, 6]
    assert ops.native_max_kv_splits(1, 32, 8, 256, 32) == 32 and ops.native_max_kv_splits(256, 32, 8, 256, 32) == 1
    assert ops.native_max_kv_splits(1, 128, 1, 256, 32) == 32      # MLA: 8 q-blocks of 16 heads


def test_balanced_split_schedule_kernel_matches_its_host_mirror_and_heterogeneous_batch_decodes():
    """rx_num_kv_splits_balanced (the length-aware native schedule): device counts == ops.balanced_kv_splits_host on
    uniform, outlier and ragged batches; a uniform batch is not split at all, an outlier is; then a heterogeneous batch
    (one long request among short ones) through the backend against the oracle."""
    from sglang_amd import ops
    from sglang_amd.forward_batch import ForwardBatch

    rng = np.random.default_rng(5)
    for lens, hq, hkv in ([[4096] * 256, 32, 8], [[32768] + [1024] * 63, 32, 8], [[8192] * 4 + [512] * 124, 32, 8],
                          [rng.integers(0, 6000, size=96).tolist(), 32, 8], [[4096] * 256, 4, 1], [[32768], 32, 8],
                          [[0, 5, 70000], 8, 8],
                          # the fill rule's range (0.7 - 3 whole-request workgroups per CU, near-uniform lengths)
                          [[4096] * 192, 4, 1], [[4096] * 288, 4, 1], [[4096] * 320, 4, 1], [[4096] * 384, 4, 1],
                          [[4096] * 640, 4, 1], [[4096] * 40, 32, 8], [[2048] * 80, 32, 8], [[300] * 300 + [0] * 20, 4, 1],
                          [rng.integers(2048, 4097, size=320).tolist(), 4, 1], [rng.integers(1000, 4097, size=320).tolist(), 4, 1],
                          # ... and below 0.7 per CU, where the even share's count is replaced when it fills the chip badly
                          [[4096] * 176, 4, 1], [[8192] * 104, 4, 1], [[4096] * 144, 4, 1], [[4096] * 20, 32, 8],
                          [[8192] * 72, 4, 1], [rng.integers(3000, 4097, size=100).tolist(), 4, 1], [[300] * 100, 4, 1]):
        lt = torch.tensor(lens, dtype=torch.int64, device=DEV)
        for mt in (128, 1024):
            for mixed in (0, 512, 768, 2048, -1):  # the mixed-batch budget (wg_target_mixed), its overshoot step; -1: the rounds rule; 512 = wg_target: the whole-requests fill form (MLA)
                out = torch.zeros(len(lens), dtype=torch.int32, device=DEV)
                ops.get_num_kv_splits_balanced(out, lt, hq, hkv, 32, 512, mt, mixed)
                want = ops.balanced_kv_splits_host(lens, hq, hkv, 32, 512, mt, mixed)
                assert out.tolist() == want.tolist(), (lens[:4], hq, hkv, mt, mixed)
    mixed = ops.balanced_kv_splits_host([32768] + [1024] * 63, 32, 8, 64, 512, 1024, 768)
    assert mixed[0] == 32 and mixed[1:].max() == 1 and int(mixed.sum()) * 8 <= 768               # equal 1 k pieces, all resident
    assert (ops.balanced_kv_splits_host([4096] * 16, 32, 8, 64, 512, 1024, 768) == 4).all()      # uniform: first pass kept
    rr = ops.balanced_kv_splits_host([32768] + [1024] * 63, 32, 8, 64, 512, 1024, -1)
    assert rr[0] == 16 and rr[1:].max() == 1                                                     # rounds rule: 2 rounds -> pieces of 2 x 1 k
    assert ops.balanced_kv_splits_host([16384] * 2 + [2048] * 30, 32, 8, 64, 512, 1024, -1).max() == 12  # one round: first pass kept
    tiny = ops.balanced_kv_splits_host([30000] * 20 + [16], 32, 8, 64, 512, 128, -1)
    assert tiny[:20].max() <= ops.balanced_kv_splits_host([30000] * 20 + [16], 32, 8, 64, 512, 128).max()  # never finer than the even share
    assert ops.balanced_kv_splits_host([4096] * 256, 32, 8, 32, 512, 1024).max() == 1          # the headline batch: one pass
    out = ops.balanced_kv_splits_host([32768] + [1024] * 63, 32, 8, 32, 512, 1024)
    assert out[0] >= 16 and out[1:].max() == 1                                                  # only the outlier is cut
    assert ops.balanced_kv_splits_host([4096] * 256, 4, 1, 32, 512, 1024).max() == 2            # TP=8 shard, fill rule off: two per CU
    assert ops.balanced_kv_splits_host([4096] * 256, 4, 1, 32, 512, 1024, -1).max() == 1        # ... on (what the backend runs): whole

    ps, hq, hkv, d = 16, 8, 2, 128
    hs = _Harness(ps, hq, hkv, d, torch.bfloat16, "shuffled_pages", "paged", max_ctx=4200, max_reqs=40)
    lens = [4000] + [64, 33, 17, 200, 129] * 6 + [1]
    bs = len(lens)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, [n - 1 for n in lens[:-1]] + [0])
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor(lens, dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, n - 2]) if n > 1 else -1 for r, n in zip(rows, lens)], dtype=torch.int64,
                        device=DEV)
    loc = hs.alloc.alloc_decode(seq.to(DEV), seq, last)
    hs.r2t.req_to_token[rpi, torch.tensor([n - 1 for n in lens], device=DEV)] = loc.to(torch.int32)
    q, k, v = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    fb = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
    hs.backend.init_forward_metadata(fb)
    md = hs.backend.forward_metadata
    # a tiny batch (64 head blocks on 256 CUs): the long request takes every slot, a 200-token one two, the rest one
    assert md.num_kv_splits is not None and int(md.num_kv_splits[0]) >= 16 and int(md.num_kv_splits[1:].max()) <= 2
    o = hs.layer(q, k, v, fb, hs.backend)
    kb, vb = hs.pool.get_kv_buffer(0)
    args = (_bits(hs.r2t.req_to_token), np.array(rows), np.array(lens), d ** -0.5)
    want = orc.sdpa_decode_req_to_token(_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb), *args)
    absw = orc.sdpa_decode_req_to_token(_bits(q.view(bs, hq, d)), _bits(kb), parity.abs_values(_bits(vb)), *args)
    parity.check_out(o.view(bs, hq, d).float().cpu().numpy(), want, torch.bfloat16, "heterogeneous batch", absw=absw)


def test_eager_split_items_table_holds_every_pair_the_device_schedule_emits():
    """the eager metadata sized the (request, split) table and the grid from the host mirror run with
    cap = native_split_cap while the device pass ran with cap = max(host counts); under the rounds rule the smaller cap
    ends the round search earlier and hands out MORE pairs (this batch: host 49, device 59), so the shortest requests got
    no workgroup and their rows stayed uninitialised.  Both passes now take the same cap: the table's count equals the
    host's, every request's output meets the oracle."""
    from sglang_amd import ops
    from sglang_amd.forward_batch import ForwardBatch

    lens = [17702, 2856, 201, 518, 2486, 2851, 822, 1004, 2620, 1327, 892, 2500, 845, 1286, 1967, 1693, 348, 179, 2610,
            2285, 2529, 1660, 2470, 1056, 1412, 2386]
    hq, hkv, d, ps = 32, 8, 128, 16
    host32 = ops.balanced_kv_splits_host(lens, hq, hkv, 32, 512, 1024, -1)
    hostS = ops.balanced_kv_splits_host(lens, hq, hkv, int(host32.max()), 512, 1024, -1)
    assert int(hostS.sum()) > int(host32.sum())  # the batch does separate the two caps (else it tests nothing)
    bs = len(lens)
    hs = _Harness(ps, hq, hkv, d, torch.bfloat16, "shuffled_pages", "paged", max_ctx=18000, max_reqs=32, size=1 << 16)
    assert hs.backend.device_core_count == 256, "the example is tuned to 256 CUs"
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, [n - 1 for n in lens])
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor(lens, dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, n - 2]) for r, n in zip(rows, lens)], dtype=torch.int64, device=DEV)
    loc = hs.alloc.alloc_decode(seq.to(DEV), seq, last)
    hs.r2t.req_to_token[rpi, torch.tensor([n - 1 for n in lens], device=DEV)] = loc.to(torch.int32)
    q, k, v = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    fb = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
    hs.backend.init_forward_metadata(fb)
    md = hs.backend.forward_metadata
    assert md.split_items is not None
    live = int(md.split_items.count.item())
    assert live <= md.split_items.cap, (live, md.split_items.cap)
    assert live == int(host32.sum()) and md.num_kv_splits.tolist() == host32.tolist()
    o = hs.layer(q, k, v, fb, hs.backend)
    kb, vb = hs.pool.get_kv_buffer(0)
    args = (_bits(hs.r2t.req_to_token), np.array(rows), np.array(lens), d ** -0.5)
    want = orc.sdpa_decode_req_to_token(_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb), *args)
    parity.check_out(o.view(bs, hq, d).float().cpu().numpy(), want, torch.bfloat16, "skewed batch, rounds rule")


def test_short_extend_over_long_prefix_takes_split_kv_path():
    """A small batch of short extends over long cached prefixes (a follow-up turn on a long conversation): the
    backend cuts the prefix into chunks (ops.VerifySplitKV with the causal rule) -- same result as the oracle."""
    from sglang_amd.forward_batch import ForwardBatch

    ps, hq, hkv, d = 16, 8, 2, 128
    hs = _Harness(ps, hq, hkv, d, torch.bfloat16, "shuffled_pages", "paged")
    prefix_lens, extend_lens = (2048, 1500), (40, 40)
    bs = len(prefix_lens)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows, prefix_lens)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
    loc = hs.alloc_extend(rows, list(prefix_lens), seq_lens)
    T = sum(extend_lens)
    q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)
    fb = ForwardBatch.for_extend(rpi, torch.tensor(seq_lens, device=DEV), loc, list(prefix_lens), list(extend_lens))
    hs.backend.init_forward_metadata(fb)
    assert hs.backend._extend_split_on and hs.backend._verify_split.num_chunks(bs, 40) >= 2
    o = hs.layer(q, k, v, fb, hs.backend)
    kb, vb = hs.pool.get_kv_buffer(0)
    want, absw = parity.want_and_absw(orc.sdpa_extend_req_to_token, (
        _bits(q.view(T, hq, d)), _bits(kb), _bits(vb), _bits(hs.r2t.req_to_token), np.array(rows), np.array(seq_lens),
        np.array(prefix_lens), np.array(extend_lens), d ** -0.5), (2,))
    got = o.view(T, hq, d).float().cpu().numpy().astype(np.float64)
    assert hs.pool.check_errors() == 0
    parity.check_out(got, want, o.dtype, "short extend, split-KV path", ulps=2, absw=absw)  # (16-bit chunk partials: 2 ulp)


def test_mla_fp8_latent_pool_radix_hit_extend():
    """An fp8 e4m3 latent pool under an extend over its cached rows: the backend upcasts the rows the batch reads
    (exactly) and runs the 16-bit MFMA kernel on the copy -- vs the oracle on the pool's dequantised values; the new
    tokens take part in 16 bits (as in the reference: k / v of the extend part are the layer's tensors)."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool

    hq, rank, rope = 16, 512, 64
    pre, ext = [70, 0, 300], [9, 40, 130]
    bs = len(pre)
    pool = MLATokenToKVPool(1024, 1, torch.float8_e4m3fn, rank, rope, 1, DEV)
    r2t = ReqToTokenPool(4, 512, DEV)
    g = torch.Generator().manual_seed(8)
    perm = torch.randperm(1023, generator=g) + 1
    rows = r2t.alloc(bs)
    layer = RadixAttention(hq, rank + rope, (128 + 64) ** -0.5, 1, 0, v_head_dim=rank)
    off, new_loc = 0, []
    for r, p_, e_ in zip(rows, pre, ext):
        slots = perm[off: off + p_ + e_].to(DEV); off += p_ + e_
        r2t.req_to_token[r, : p_ + e_] = slots.int()
        if p_:
            pool.set_mla_kv_buffer(layer, slots[:p_], (torch.randn(p_, 1, rank, generator=g) * 0.5).to(torch.bfloat16).to(DEV),
                                   (torch.randn(p_, 1, rope, generator=g) * 0.5).to(torch.bfloat16).to(DEV))
        new_loc.append(slots[p_:])
    loc = torch.cat(new_loc)

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, 1, 512

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        model_config = MC
        page_size = 1

    be = HipRadixAttnBackend(MR)
    T = sum(ext)
    k_new = (torch.randn(T, 1, rank + rope, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    q = torch.randn(T, hq * (rank + rope), generator=g).to(torch.bfloat16).to(DEV)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    fb = ForwardBatch.for_extend(rpi, torch.tensor([p_ + e_ for p_, e_ in zip(pre, ext)], device=DEV), loc, pre, ext)
    be.init_forward_metadata(fb)
    o = layer(q, k_new, k_new[..., :rank], fb, be)
    kbn = pool.get_key_buffer(0).float().cpu().numpy()  # dequantised rows
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.concatenate([_bits(r2t.req_to_token[r, :p_]).astype(np.int64) for r, p_ in zip(rows, pre)])
    want = orc.extend_attention(q.view(T, hq, rank + rope).float().cpu().numpy(), k_new.float().cpu().numpy(),
                                k_new[..., :rank].float().cpu().numpy(), kbn, kbn[..., :rank], qo, kv_indptr, kv_indices,
                                sm_scale=layer.scaling)
    got = o.view(T, hq, rank).float().cpu().numpy()
    parity.check_out(got, want, torch.bfloat16, ("mla_fp8_radix_hit_extend",))


def test_short_mla_extend_over_long_prefix_takes_split_kv_path():
    """The same follow-up-turn shape on an MLA pool (latent rows 576 / 512, one kv head): the backend's split-KV extend
    runs rx::extend_mla_kernel over chunks of the prefix and merges -- same result as the oracle."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MLATokenToKVPool, ReqToTokenPool

    hq, rank, rope = 16, 512, 64
    pre, ext = [2048, 1500], [24, 24]
    bs = len(pre)
    pool = MLATokenToKVPool(4096, 1, torch.bfloat16, rank, rope, 1, DEV)
    r2t = ReqToTokenPool(4, 4096, DEV)
    g = torch.Generator().manual_seed(4)
    perm = torch.randperm(4095, generator=g) + 1
    rows = r2t.alloc(bs)
    layer = RadixAttention(hq, rank + rope, (128 + 64) ** -0.5, 1, 0, v_head_dim=rank)
    off, new_loc = 0, []
    for r, p_, e_ in zip(rows, pre, ext):
        slots = perm[off: off + p_ + e_].to(DEV); off += p_ + e_
        r2t.req_to_token[r, : p_ + e_] = slots.int()
        pool.set_mla_kv_buffer(layer, slots[:p_], (torch.randn(p_, 1, rank, generator=g) * 0.5).to(torch.bfloat16).to(DEV),
                               (torch.randn(p_, 1, rope, generator=g) * 0.5).to(torch.bfloat16).to(DEV))
        new_loc.append(slots[p_:])
    loc = torch.cat(new_loc)

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, 1, 4096

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        model_config = MC
        page_size = 1

    be = HipRadixAttnBackend(MR)
    T = sum(ext)
    k_new = (torch.randn(T, 1, rank + rope, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    q = torch.randn(T, hq * (rank + rope), generator=g).to(torch.bfloat16).to(DEV)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    fb = ForwardBatch.for_extend(rpi, torch.tensor([p_ + e_ for p_, e_ in zip(pre, ext)], device=DEV), loc, pre, ext)
    be.init_forward_metadata(fb)
    assert be._extend_split_on and be._verify_split.num_chunks(bs, 24) >= 2 and be._verify_split.dv == 512
    o = layer(q, k_new, k_new[..., :rank], fb, be)
    kbn = pool.get_key_buffer(0).float().cpu().numpy()
    kv_indptr = np.concatenate([[0], np.cumsum(pre)]).astype(np.int32)
    qo = np.concatenate([[0], np.cumsum(ext)]).astype(np.int64)
    kv_indices = np.concatenate([_bits(r2t.req_to_token[r, :p_]).astype(np.int64) for r, p_ in zip(rows, pre)])
    want, absw = parity.want_and_absw(orc.extend_attention, (
        q.view(T, hq, rank + rope).float().cpu().numpy(), k_new.float().cpu().numpy(), k_new[..., :rank].float().cpu().numpy(),
        kbn, kbn[..., :rank], qo, kv_indptr, kv_indices), (2, 4), sm_scale=layer.scaling)
    got = o.view(T, hq, rank).float().cpu().numpy().astype(np.float64)
    parity.check_out(got, want, o.dtype, "short mla extend, split-KV path", ulps=2, absw=absw)  # (16-bit chunk partials: 2 ulp)


def _capture(fn):
    """Warm on a side stream (allocator, lazy launchers), then capture fn() into a HIP graph."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
    return graph, out


@pytest.mark.parametrize("index_mode,policy", [("paged", "native"), ("indices", "native"), ("paged", "reference"),
                                               ("indices", "reference")])
def test_graph_replay_refreshes_static_metadata(index_mode, policy):
    """The runner's contract (decode_cuda_graph_runner.py:946-953,1168): capture once with seq_lens = the fill
    value in address-stable input buffers, then before EVERY replay copy the live batch into those buffers and call
    init_forward_metadata_out_graph(fb) -- which must refill the static num_kv_splits / partials / kv_indices the
    captured kernels read (triton_backend.py:572-632).  Replay == eager on the same live batch."""
    from sglang_amd.forward_batch import ForwardBatch

    hs = _Harness(16, 32, 8, 128, torch.bfloat16, "shuffled_pages", index_mode, split_policy=policy)
    be = hs.backend
    bs = 3
    be.init_cuda_graph_state(4, 4)
    fill = be.get_cuda_graph_seq_len_fill_value()
    s_rpi = torch.zeros(bs, dtype=torch.int64, device=DEV)           # padding row 0
    s_seq = torch.full((bs,), fill, dtype=torch.int64, device=DEV)
    s_loc = torch.zeros(bs, dtype=torch.int64, device=DEV)           # padding slot 0
    q, k, v = hs.rand(bs, 32 * 128), hs.rand(bs, 8 * 128), hs.rand(bs, 8 * 128)
    fb_g = ForwardBatch.for_decode(s_rpi, s_seq, s_loc, torch.full((bs,), fill, dtype=torch.int64))
    be.init_forward_metadata_out_graph(fb_g, in_capture=True)
    graph, out = _capture(lambda: hs.layer(q, k, v, fb_g, be))

    # two different live batches replayed through the SAME graph
    rows = hs.r2t.alloc(bs)
    for prefix in ((600, 33, 1024), (1, 2047, 130)):
        hs.fill_prefix(rows, prefix)
        rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
        seq = torch.tensor([p + 1 for p in prefix], dtype=torch.int64)
        last = torch.tensor([int(hs.r2t.req_to_token[r, p - 1]) for r, p in zip(rows, prefix)],
                            dtype=torch.int64, device=DEV)
        loc = hs.alloc.alloc_decode(seq.to(DEV), seq, last)
        hs.r2t.req_to_token[rpi, torch.tensor(prefix, device=DEV)] = loc.to(torch.int32)
        q.copy_(hs.rand(bs, 32 * 128)); k.copy_(hs.rand(bs, 8 * 128)); v.copy_(hs.rand(bs, 8 * 128))
        s_rpi.copy_(rpi); s_seq.copy_(seq.to(DEV)); s_loc.copy_(loc)
        fb_r = ForwardBatch.for_decode(s_rpi, s_seq, s_loc, seq)
        be.init_forward_metadata_out_graph(fb_r)
        graph.replay()
        torch.cuda.synchronize()
        replayed = out.clone()
        fb_e = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
        be.init_forward_metadata(fb_e)
        eager = hs.layer(q, k, v, fb_e, be)
        kb, vb = hs.pool.get_kv_buffer(0)
        want, absw = parity.want_and_absw(orc.sdpa_decode_req_to_token, (
            _bits(q.view(bs, 32, 128)), _bits(kb), _bits(vb), _bits(hs.r2t.req_to_token), np.array(rows), seq.numpy(),
            128 ** -0.5), (2,))
        parity.check_out(replayed.view(bs, 32, 128).float().cpu().numpy(), want, replayed.dtype,
                         ("graph replay", index_mode, policy, prefix), ulps=1, absw=absw)
        assert torch.equal(replayed, eager), (index_mode, policy, prefix)
        # free the batch's pages so that the next one starts from empty rows
        for r, p in zip(rows, prefix):
            hs.alloc.free(hs.r2t.req_to_token[r, : p + 1].to(torch.int64))
    assert hs.pool.check_errors() == 0


@pytest.mark.parametrize("bs,ntok,cap_to_bs", [(257, 1536, False), (129, 2048, False), (257, 1536, True)])
def test_graph_replay_of_a_fill_rule_batch_runs_every_pair(bs, ntok, cap_to_bs):
    """a TP=8 shard (Hq 8 / Hkv 1) replaying a near-uniform batch from a HIP graph.  The fill rule
    gives every request the same count (up to 6), far more (request, split) pairs than the even-share bound the graph
    path used to size its table and grid with -- the shortest requests then merged partial rows nobody wrote.
    Replay must equal the eager step bit for bit and meet the oracle.  cap_to_bs: the bound is forced down to bs, so
    the guarded build must replace the schedule (whole requests) rather than drop pairs -- still correct."""
    from sglang_amd.attention import backend as bk
    from sglang_amd.forward_batch import ForwardBatch

    hq, hkv, d = 8, 1, 128
    hs = _Harness(16, hq, hkv, d, torch.bfloat16, "shuffled_pages", "paged", max_ctx=ntok + 64, max_reqs=bs + 1,
                  size=(bs + 2) * (ntok + 32))
    be = hs.backend
    if cap_to_bs:
        be._split_pairs_bound = lambda b, slots: b
    be.init_cuda_graph_state(bs, bs)
    fill = be.get_cuda_graph_seq_len_fill_value()
    s_rpi = torch.zeros(bs, dtype=torch.int64, device=DEV)
    s_seq = torch.full((bs,), fill, dtype=torch.int64, device=DEV)
    s_loc = torch.zeros(bs, dtype=torch.int64, device=DEV)
    q, k, v = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    fb_g = ForwardBatch.for_decode(s_rpi, s_seq, s_loc, torch.full((bs,), fill, dtype=torch.int64))
    be.init_forward_metadata_out_graph(fb_g, in_capture=True)
    graph, out = _capture(lambda: hs.layer(q, k, v, fb_g, be))
    rows = hs.r2t.alloc(bs)
    prefix = [ntok - 1] * bs
    hs.fill_prefix(rows, prefix)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor([p + 1 for p in prefix], dtype=torch.int64)
    last = hs.r2t.req_to_token[rpi, ntok - 2].to(torch.int64)
    loc = hs.alloc.alloc_decode(seq.to(DEV), seq, last)
    hs.r2t.req_to_token[rpi, ntok - 1] = loc.to(torch.int32)
    s_rpi.copy_(rpi); s_seq.copy_(seq.to(DEV)); s_loc.copy_(loc)
    fb_r = ForwardBatch.for_decode(s_rpi, s_seq, s_loc, seq)
    be.init_forward_metadata_out_graph(fb_r)
    graph.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    md = be.forward_metadata
    counts = md.num_kv_splits.cpu().numpy()
    live = int(md.split_items.count.item())
    assert live == int(counts.sum()) <= md.split_items.cap        # the table holds every live pair
    if cap_to_bs:
        assert int(md.split_items.overflow.item()) == 1 and (counts == 1).all()
    else:
        assert int(md.split_items.overflow.item()) == 0 and counts.max() > 1
        if bs == 257 and be.device_core_count == 256:
            assert live == 6 * bs > bs + 3 * be.device_core_count + 1   # beyond the OLD bound: the case of the finding
        assert live <= bk.split_pairs_bound(bs, int(md.max_kv_splits), 1, be.device_core_count)
    kb, vb = hs.pool.get_kv_buffer(0)
    want, absw = parity.want_and_absw(orc.sdpa_decode_req_to_token, (
        _bits(q.view(bs, hq, d)), _bits(kb), _bits(vb), _bits(hs.r2t.req_to_token), np.array(rows), seq.numpy(),
        d ** -0.5), (2,))
    parity.check_out(replayed.view(bs, hq, d).float().cpu().numpy(), want, replayed.dtype,
                     ("graph replay, fill rule", bs, ntok, cap_to_bs), ulps=1, absw=absw)
    if not cap_to_bs:
        fb_e = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
        be.init_forward_metadata(fb_e)
        eager = hs.layer(q, k, v, fb_e, be)
        assert torch.equal(replayed, eager)
    assert hs.pool.check_errors() == 0


def test_target_verify_graph_replay_refreshes_indices_and_mask():
    """TARGET_VERIFY under a HIP graph: qo_indptr / kv_indices / mask_indptr / the mask bytes are address-stable
    and refilled by init_forward_metadata_out_graph before each replay (triton_backend.py:1016-1063)."""
    from sglang_amd.forward_batch import ForwardBatch, ForwardMode

    ps, hq, hkv, d, nd = 16, 8, 2, 128, 4
    hs = _Harness(ps, hq, hkv, d, torch.float16, "shuffled_pages", "paged")
    be = hs.backend
    bs = 2
    be.init_cuda_graph_state(2, 2 * nd)
    T = bs * nd
    s_rpi = torch.zeros(bs, dtype=torch.int64, device=DEV)
    s_seq = torch.full((bs,), 1, dtype=torch.int64, device=DEV)
    s_loc = torch.zeros(T, dtype=torch.int64, device=DEV)
    s_mask = torch.ones(bs * nd * (hs.backend.max_context_len + nd), dtype=torch.bool, device=DEV)
    q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)

    class Spec:
        draft_token_num = nd
        custom_mask = s_mask

    def make_fb(seq_cpu):
        return ForwardBatch(forward_mode=ForwardMode.TARGET_VERIFY, batch_size=bs, req_pool_indices=s_rpi,
                            seq_lens=s_seq, out_cache_loc=s_loc, seq_lens_sum=int(seq_cpu.sum()),
                            seq_lens_cpu=seq_cpu, spec_info=Spec)

    fb_g = make_fb(torch.ones(bs, dtype=torch.int64))
    be.init_forward_metadata_out_graph(fb_g, in_capture=True)
    graph, out = _capture(lambda: hs.layer(q, k, v, fb_g, be))
    rows = hs.r2t.alloc(bs)
    rng = np.random.default_rng(9)
    for seq_lens in ((300, 77), (64, 1500)):
        hs.fill_prefix(rows, seq_lens)
        total = [s + nd for s in seq_lens]
        loc = hs.alloc_extend(rows, list(seq_lens), total)
        masks = []
        for s in seq_lens:
            m = np.ones((nd, s + nd), dtype=bool)
            tri = np.tril(rng.random((nd, nd)) < 0.5)
            np.fill_diagonal(tri, True)
            m[:, s:] = tri
            masks.append(m.reshape(-1))
        cm = np.concatenate(masks)
        s_mask[: cm.size].copy_(torch.from_numpy(cm).to(DEV))
        s_rpi.copy_(torch.tensor(rows, device=DEV)); s_seq.copy_(torch.tensor(seq_lens, device=DEV)); s_loc.copy_(loc)
        q.copy_(hs.rand(T, hq * d)); k.copy_(hs.rand(T, hkv * d)); v.copy_(hs.rand(T, hkv * d))
        be.init_forward_metadata_out_graph(make_fb(torch.tensor(seq_lens, dtype=torch.int64)))
        graph.replay()
        torch.cuda.synchronize()
        kb, vb = hs.pool.get_kv_buffer(0)
        r2t = _bits(hs.r2t.req_to_token)
        kv_indptr, kv_indices = orc.build_kv_indices(r2t, np.array(rows), np.array(seq_lens))
        qo = (np.arange(bs + 1) * nd).astype(np.int64)
        kbn, vbn = _bits(kb), _bits(vb)
        ke = np.concatenate([kbn[r2t[rows[i], seq_lens[i]: total[i]]] for i in range(bs)])
        ve = np.concatenate([vbn[r2t[rows[i], seq_lens[i]: total[i]]] for i in range(bs)])
        mi = np.concatenate([[0], np.cumsum([m.size for m in masks])]).astype(np.int64)
        want, absw = parity.want_and_absw(orc.extend_attention, (_bits(q.view(T, hq, d)), ke, ve, kbn, vbn, qo, kv_indptr,
                                                                 kv_indices), (2, 4), is_causal=True, sm_scale=d ** -0.5,
                                          custom_mask=cm, mask_indptr=mi)
        parity.check_out(out.view(T, hq, d).float().cpu().numpy(), want, out.dtype, ("verify graph replay", seq_lens), ulps=2, absw=absw)  # (split-KV partials: 2 ulp)
        for r, t in zip(rows, total):
            hs.alloc.free(hs.r2t.req_to_token[r, :t].to(torch.int64))
    assert hs.pool.check_errors() == 0


def test_roctx_ranges_do_not_change_a_decode_step():
    """Option `roctx` (named ranges around the library's launches and per layer from the backend): the same decode
    step with the option on and off -- libroctx64.so present or not -- gives the same bits."""
    from sglang_amd import lib as rxlib
    from sglang_amd.forward_batch import ForwardBatch

    hs = _Harness(16, 32, 8, 128, torch.bfloat16, "shuffled_pages", "paged")
    rows = hs.r2t.alloc(3)
    prefix = (200, 33, 1024)
    hs.fill_prefix(rows, prefix)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor([p + 1 for p in prefix], dtype=torch.int64)
    last = torch.tensor([int(hs.r2t.req_to_token[r, p - 1]) for r, p in zip(rows, prefix)], dtype=torch.int64, device=DEV)
    loc = hs.alloc.alloc_decode(seq.to(DEV), seq, last)
    hs.r2t.req_to_token[rpi, torch.tensor(prefix, device=DEV)] = loc.to(torch.int32)
    q, k, v = hs.rand(3, 32 * 128), hs.rand(3, 8 * 128), hs.rand(3, 8 * 128)
    outs = []
    for on in (0, 1):
        with rxlib.option("roctx", on):
            fb = ForwardBatch.for_decode(rpi, seq.to(DEV), loc, seq)
            hs.backend.init_forward_metadata(fb)
            outs.append(hs.layer(q, k, v, fb, hs.backend).clone())
            torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.gpu
def test_clock_probe_reads_a_plausible_shader_clock():
    """rx_clock_probe (bench.py's sustained-clock figure): one sleeping wave, s_memtime cycles against 100-MHz s_memrealtime
    ticks over the requested time."""
    from sglang_amd import lib as rxlib

    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    rxlib.clock_probe(out, 2000, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cyc, ticks = (int(x) for x in out.tolist())
    assert 2000 * 100 <= ticks < 2000 * 100 * 2, ticks           # 2 ms of the constant-rate clock (it overshoots by one sleep at most)
    assert 50.0 < cyc / ticks * 100.0 < 3000.0, (cyc, ticks)     # MHz: anything from the idle state to the 2.4 GHz peak
    with pytest.raises(rxlib.RadixHipError):
        rxlib.clock_probe(out, 0, torch.cuda.current_stream().cuda_stream)


@pytest.mark.gpu
def test_bidirectional_decoder_layer_extends_without_the_causal_triangle():
    """triton_backend.py:235-245,1318-1327: a DECODER_BIDIRECTIONAL layer (image tokens) attends non-causally in an extend
    when prefills are whole (chunked_prefill_size == -1, no graph mode) and causally otherwise."""
    from sglang_amd.attention.radix_attention import AttentionType, RadixAttention
    from sglang_amd.forward_batch import ForwardBatch

    hq, hkv, d = 4, 2, 128
    outs = {}
    for whole in (True, False):
        hs = _Harness(16, hq, hkv, d, torch.float16, "contiguous", "paged",
                      server_args_extra={"chunked_prefill_size": -1 if whole else 8192, "disable_cuda_graph": True})
        assert hs.backend.allow_bidirectional_attention_in_extend == whole
        layer = RadixAttention(hq, d, d ** -0.5, hkv, 0, attn_type=AttentionType.DECODER_BIDIRECTIONAL)
        prefix_lens, extend_lens = [40, 0], [70, 33]
        rows = hs.r2t.alloc(2)
        hs.fill_prefix(rows, prefix_lens)
        seq_lens = [p + e for p, e in zip(prefix_lens, extend_lens)]
        loc = hs.alloc_extend(rows, prefix_lens, seq_lens)
        T = sum(extend_lens)
        q, k, v = hs.rand(T, hq * d), hs.rand(T, hkv * d), hs.rand(T, hkv * d)
        fb = ForwardBatch.for_extend(torch.tensor(rows, dtype=torch.int64, device=DEV), torch.tensor(seq_lens, device=DEV), loc,
                                     prefix_lens, extend_lens)
        hs.backend.init_forward_metadata(fb)
        o = layer(q, k, v, fb, hs.backend)
        kb, vb = hs.pool.get_kv_buffer(0)
        r2t = hs.r2t.req_to_token.cpu().numpy()
        kvi = np.concatenate([r2t[r, :p] for r, p in zip(rows, prefix_lens)]).astype(np.int64)
        kvp = np.concatenate([[0], np.cumsum(prefix_lens)]).astype(np.int32)
        qo = np.concatenate([[0], np.cumsum(extend_lens)]).astype(np.int64)
        want, absw = parity.want_and_absw(orc.extend_attention, (_bits(q.view(T, hq, d)), _bits(k.view(T, hkv, d)), _bits(v.view(T, hkv, d)),
                                                                 _bits(kb), _bits(vb), qo, kvp, kvi), (2, 4), is_causal=not whole, sm_scale=d ** -0.5)
        parity.check_out(_bits(o.view(T, hq, d)).astype(np.float64), want, torch.float16, ("bidirectional", whole), absw=absw)
        outs[whole] = o
    assert not torch.equal(outs[True], outs[False])
    with pytest.raises(ValueError):
        layer.logit_capping_method = "sigmoid"
        layer(q, k, v, fb, hs.backend)
