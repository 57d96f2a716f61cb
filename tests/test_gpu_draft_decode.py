"""EAGLE multi-step draft decode (SURVEY 8f-4 widening): rx_draft_decode_kv_indices against the golden
of the reference's Triton kernel (bit-exact, int64 and int32 index words), and HipRadixMultiStepDraftBackend driving
speculative_num_steps - 1 decode steps over top-k branches -- eagerly and from HIP graphs -- against the fp64 oracle."""
import os

import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("word", [torch.int64, torch.int32], ids=["int64", "int32"])
def test_draft_decode_kv_indices_golden_bit_exact(golden_dir, word):
    from sglang_amd import ops

    z = np.load(os.path.join(golden_dir, "draft_kv_indices.npz"))
    for n in sorted({k.split(".")[0] for k in z.files}):
        g = {k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(n + ".")}
        steps, topk, ps = int(g["num_steps"]), int(g["topk"]), int(g["page_size"])
        kvi = torch.full(g["kv_indices"].shape, -1, dtype=word, device=DEV)
        kvp = torch.zeros(g["kv_indptr"].shape, dtype=torch.int32, device=DEV)
        for rpi_dt, len_dt in ((torch.int64, torch.int64), (torch.int32, torch.int32)):
            kvi.fill_(-1)
            kvp.zero_()
            ops.generate_draft_decode_kv_indices(_t(g["req_pool_indices"]).to(rpi_dt), _t(g["req_to_token"]), _t(g["seq_lens"]).to(len_dt),
                                                 kvi, kvp, _t(g["positions"]).to(len_dt), topk, steps, ps)
            torch.cuda.synchronize()
            assert np.array_equal(kvi.cpu().numpy().astype(np.int64), g["kv_indices"]), (n, word)
            assert np.array_equal(kvp.cpu().numpy(), g["kv_indptr"]), (n, word)


class _Spec:
    kv_indptr = None
    kv_indices = None


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("topk,page_size", [(1, 16), (4, 1), (4, 16)])
def test_multi_step_draft_backend_against_the_oracle(topk, page_size, graph):
    """num_steps = 3: two draft decode steps.  Every branch's draft tokens sit where the draft worker's allocator puts them
    (assign_draft_cache_locs, cache_locs.py:160-230: contiguous behind the request for page_size 1 or topk 1, on pages of
    their own per branch otherwise); step i's query of branch (b, k) must see the request's cached tokens and the branch's
    own first i + 1 draft tokens, nothing of the other branches."""
    from sglang_amd.attention.backend import HipRadixMultiStepDraftBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

    hq, hkv, d, steps = 8, 2, 128, 3
    lens = [70, 16, 129]
    num_seqs = len(lens)
    max_ctx = 512
    rng = np.random.default_rng(topk * 10 + page_size)
    size = 4096
    pool = MHATokenToKVPool(size, page_size, torch.bfloat16, hkv, d, 1, DEV)
    r2t = ReqToTokenPool(8, max_ctx, DEV)
    rows = r2t.alloc(num_seqs)
    # page table: cached tokens on shuffled pages, then the draft region as the reference lays it out
    npages = size // page_size
    perm = rng.permutation(np.arange(1, npages))
    pi = 0
    g = torch.Generator().manual_seed(7)
    kb, vb = pool.get_kv_buffer(0)
    kb.copy_(torch.randn(kb.shape, generator=g).to(torch.bfloat16))
    vb.copy_(torch.randn(vb.shape, generator=g).to(torch.bfloat16))
    branch_slots = {}
    for b, (r, n) in enumerate(zip(rows, lens)):
        need = n + (steps + page_size) * topk + page_size
        k = -(-need // page_size)
        sl = np.concatenate([np.arange(p * page_size, (p + 1) * page_size) for p in perm[pi: pi + k]])
        pi += k
        r2t.req_to_token[r, : len(sl)] = torch.from_numpy(sl.astype(np.int32)).to(DEV)
        for kk in range(topk):
            if page_size == 1 or topk == 1:
                start = n + kk * steps
            else:
                last = n % page_size
                start = n // page_size * page_size + kk * (-(-(last + steps) // page_size)) * page_size + last
            branch_slots[(b, kk)] = sl[start: start + steps]

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, hkv, max_ctx

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        model_config = MC

        class server_args:
            triton_attention_num_kv_splits = 8

    MR.page_size = page_size
    be = HipRadixMultiStepDraftBackend(MR, topk, steps)
    assert len(be.attn_backends) == steps - 1
    layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
    bs = num_seqs * topk
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    seq = torch.tensor(lens, dtype=torch.int64, device=DEV)
    spec = _Spec()
    fb = ForwardBatch.for_decode(rpi, seq, torch.zeros(bs, dtype=torch.int64, device=DEV), torch.tensor(lens, dtype=torch.int64))
    fb.spec_info = spec
    fb.positions = seq.repeat_interleave(topk)
    if graph:
        be.init_cuda_graph_state(num_seqs, bs)
        be.init_forward_metadata_out_graph(fb, in_capture=True)
    else:
        be.init_forward_metadata(fb)
    kbn, vbn = _bits(kb), _bits(vb)
    r2t_np = r2t.req_to_token.cpu().numpy()
    for i in range(steps - 1):
        # the draft worker has written draft tokens 0 .. i of every branch before step i's attention (its own token i included)
        q = torch.randn(bs, hq * d, generator=g).to(torch.bfloat16).to(DEV)
        want_idx, want_ptr = [], [0]
        for b, n in enumerate(lens):
            for kk in range(topk):
                want_idx.append(np.concatenate([r2t_np[rows[b], :n], branch_slots[(b, kk)][: i + 1]]))
                want_ptr.append(want_ptr[-1] + n + i + 1)
        md = be.attn_backends[i].forward_metadata
        assert md.draft and md.kv_indptr.cpu().numpy().tolist() == want_ptr
        got_idx = md.kv_indices.cpu().numpy()[: want_ptr[-1]]
        assert np.array_equal(got_idx, np.concatenate(want_idx).astype(np.int64))
        o = be.attn_backends[i].forward_decode(q, None, None, layer, fb, save_kv_cache=False)
        torch.cuda.synchronize()
        want, absw = parity.want_and_absw(orc.decode_attention, (_bits(q.view(bs, hq, d)), kbn, vbn, np.asarray(want_ptr, dtype=np.int32),
                                                                   np.concatenate(want_idx).astype(np.int64), d ** -0.5), (2,))
        parity.check_out(o.view(bs, hq, d).float().cpu().numpy(), want, torch.bfloat16, ("multi-step draft", topk, page_size, i, graph),
                         ulps=1, absw=absw)


def test_multi_step_draft_backend_replays_from_a_hip_graph():
    """The draft worker's contract under CUDA graphs (triton_backend.py:2002-2036): capture the two draft decode steps once
    on the address-stable buffers of init_cuda_graph_state, then for every new batch copy its lengths / rows into the static
    inputs, call init_forward_metadata_out_graph (rebuilds all steps' page tables and the split counts outside the graph)
    and replay.  Two different batches through the same graph match the oracle."""
    from sglang_amd.attention.backend import HipRadixMultiStepDraftBackend
    from sglang_amd.attention.radix_attention import RadixAttention
    from sglang_amd.forward_batch import ForwardBatch
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

    hq, hkv, d, steps, topk, page_size = 8, 2, 128, 3, 4, 1
    num_seqs, max_ctx, size = 3, 400, 2048
    bs = num_seqs * topk
    pool = MHATokenToKVPool(size, page_size, torch.bfloat16, hkv, d, 1, DEV)
    r2t = ReqToTokenPool(8, max_ctx, DEV)
    rows = r2t.alloc(num_seqs)
    g = torch.Generator().manual_seed(3)
    kb, vb = pool.get_kv_buffer(0)
    kb.copy_(torch.randn(kb.shape, generator=g).to(torch.bfloat16))
    vb.copy_(torch.randn(vb.shape, generator=g).to(torch.bfloat16))

    class MC:
        num_attention_heads, num_key_value_heads, context_len = hq, hkv, max_ctx

    class MR:
        device = DEV
        req_to_token_pool = r2t
        token_to_kv_pool = pool
        model_config = MC
        page_size = 1

        class server_args:
            triton_attention_num_kv_splits = 8

    be = HipRadixMultiStepDraftBackend(MR, topk, steps)
    layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
    s_rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    s_seq = torch.ones(num_seqs, dtype=torch.int64, device=DEV)
    s_pos = torch.ones(bs, dtype=torch.int64, device=DEV)
    qs = [torch.zeros(bs, hq * d, dtype=torch.bfloat16, device=DEV) for _ in range(steps - 1)]
    spec = _Spec()
    fb = ForwardBatch.for_decode(s_rpi, s_seq, torch.zeros(bs, dtype=torch.int64, device=DEV), torch.ones(num_seqs, dtype=torch.int64))
    fb.spec_info, fb.positions, fb.seq_lens_sum = spec, s_pos, None
    be.init_cuda_graph_state(num_seqs, bs)
    be.init_forward_metadata_out_graph(fb, in_capture=True)

    def run_steps():
        return [be.attn_backends[i].forward_decode(qs[i], None, None, layer, fb, save_kv_cache=False) for i in range(steps - 1)]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run_steps()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = run_steps()
    kbn, vbn = _bits(kb), _bits(vb)
    rng = np.random.default_rng(5)
    for lens in ([70, 16, 129], [5, 300, 41]):
        perm = rng.permutation(np.arange(1, size))
        pi = 0
        for r, n in zip(rows, lens):
            need = n + steps * topk
            r2t.req_to_token[r, :need] = torch.from_numpy(perm[pi: pi + need].astype(np.int32)).to(DEV)
            pi += need
        s_seq.copy_(torch.tensor(lens, dtype=torch.int64))
        s_pos.copy_(s_seq.repeat_interleave(topk))
        for qbuf in qs:
            qbuf.copy_(torch.randn(bs, hq * d, generator=g).to(torch.bfloat16))
        be.init_forward_metadata_out_graph(fb)
        graph.replay()
        torch.cuda.synchronize()
        r2t_np = r2t.req_to_token.cpu().numpy()
        for i in range(steps - 1):
            idx, ptr = [], [0]
            for b, n in enumerate(lens):
                for kk in range(topk):
                    idx.append(np.concatenate([r2t_np[rows[b], :n], r2t_np[rows[b], n + kk * steps: n + kk * steps + i + 1]]))
                    ptr.append(ptr[-1] + n + i + 1)
            want, absw = parity.want_and_absw(orc.decode_attention, (_bits(qs[i].view(bs, hq, d)), kbn, vbn, np.asarray(ptr, dtype=np.int32),
                                                                       np.concatenate(idx).astype(np.int64), d ** -0.5), (2,))
            parity.check_out(outs[i].view(bs, hq, d).float().cpu().numpy(), want, torch.bfloat16, ("multi-step draft, graph replay", lens, i),
                             ulps=1, absw=absw)
