"""Full-size property tests: BASELINE.json's metric shape (Llama-3-8B, bs 256, ctx 4096, Hq 32 / Hkv 8,
D 128, bf16, page 16) is far too large for the CPU oracle, so parity at that size goes through properties
that do not depend on the size: softmax normalisation, one-hot retrieval through the page table, layout
invariance, split invariance, extend/decode cross-consistency, allocator invariants, store round trips.
One layer's pools (2 GiB K + 2 GiB V) are used."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BS, CTX, HQ, HKV, D, PS = 256, 4096, 32, 8, 128, 16


@pytest.fixture(scope="module")
def ops():
    from sglang_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def paged():
    """Shuffled page table for BS requests of CTX tokens (page 0 reserved), as bench.py builds it."""
    pages_per_req = CTX // PS
    rng = np.random.default_rng(0)
    perm = rng.permutation(np.arange(1, BS * pages_per_req + 1))
    slots = (perm.reshape(BS, pages_per_req)[:, :, None] * PS + np.arange(PS)[None, None, :]).reshape(BS, -1)
    r2t = torch.zeros(BS + 1, CTX, dtype=torch.int32, device=DEV)
    r2t[1:] = torch.from_numpy(slots.astype(np.int32)).to(DEV)
    rpi = torch.arange(1, BS + 1, dtype=torch.int64, device=DEV)
    lens = torch.full((BS,), CTX, dtype=torch.int64, device=DEV)
    return r2t, rpi, lens, (BS * pages_per_req + 1) * PS


def _decode(ops, q, kb, vb, paged, splits=1):
    r2t, rpi, lens, _ = paged
    o = torch.empty_like(q)
    if splits == 1:
        ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, None, None, None, 1, D ** -0.5, page_size=PS)
    else:
        ns = torch.full((BS,), splits, dtype=torch.int32, device=DEV)
        al = torch.empty(BS, HQ, splits, D, dtype=torch.float32, device=DEV)
        lse = torch.empty(BS, HQ, splits, dtype=torch.float32, device=DEV)
        ops.decode_attention_fwd_paged(q, kb, vb, o, r2t, rpi, lens, al, lse, ns, splits, D ** -0.5, page_size=PS)
    return o


def test_decode_full_size_properties(ops, paged):
    r2t, rpi, lens, pool = paged
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn(BS, HQ, D, device=DEV, generator=g).to(torch.bfloat16)
    kb = torch.randn(pool, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    # (1) normalisation: V rows all equal to one vector per kv head -> the output IS that vector
    c = torch.randn(HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    vb = c.expand(pool, HKV, D).contiguous()
    o = _decode(ops, q, kb, vb, paged)
    want = c.repeat_interleave(HQ // HKV, dim=0).float()
    assert (o.float() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-6
    # (2) one-hot retrieval through the page table: a single key per (request, kv head) dominates
    vb = torch.randn(pool, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    qg = torch.randn(BS, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    q1 = qg.repeat_interleave(HQ // HKV, dim=1).contiguous()      # the heads of a group share q
    pos = torch.randint(0, CTX, (BS, HKV), device=DEV, generator=g)
    kb1 = torch.zeros(pool, HKV, D, dtype=torch.bfloat16, device=DEV)
    slot = torch.gather(r2t[1:].long(), 1, pos)                      # [BS, HKV]
    hk = torch.arange(HKV, device=DEV).expand(BS, HKV)
    kb1[slot, hk] = (qg.float() * (80.0 / (qg.float().pow(2).sum(-1, keepdim=True) * D ** -0.5))).to(torch.bfloat16)
    o1 = _decode(ops, q1, kb1, vb, paged)                            # winning logit ~ 80, the rest 0
    want1 = vb[slot, hk].repeat_interleave(HQ // HKV, dim=1)
    assert torch.equal(o1, want1)
    # (3) split invariance: 8 splits + stage-2 merge == single pass (to bf16 rounding of the output)
    o_s = _decode(ops, q, kb, vb, paged, splits=8)
    o_1 = _decode(ops, q, kb, vb, paged)
    assert (o_s.float() - o_1.float()).abs().max().item() <= 2e-3
    # (4) layout invariance: the same logical KV under a different page permutation, HND pool -> same bits
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool

    hnd = MHATokenToKVPool(pool - PS, PS, torch.bfloat16, HKV, D, 1, DEV, use_hnd=True)
    kh, vh = hnd.get_kv_buffer(0)
    kh.copy_(kb.view(-1, PS, HKV, D).permute(0, 2, 1, 3))
    vh.copy_(vb.view(-1, PS, HKV, D).permute(0, 2, 1, 3))
    o_h = torch.empty_like(q)
    ops.decode_attention_fwd_paged(q, kh, vh, o_h, r2t, rpi, lens, None, None, None, 1, D ** -0.5, page_size=PS,
                                   kv_layout=ops.kv_layout_hnd(kh, vh))
    assert torch.equal(o_h, o_1)


def test_extend_full_size_properties(ops):
    """Config-3 chunk: 32 requests x (3584-token shared prefix + 512 new tokens)."""
    P, E, chunk = 3584, 512, 32
    g = torch.Generator(device=DEV).manual_seed(2)
    pool = P + 16
    kb = torch.randn(pool, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    vb = torch.randn(pool, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    T = chunk * E
    q = torch.randn(T, HQ, D, device=DEV, generator=g).to(torch.bfloat16)
    ke = torch.randn(T, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    ve = torch.randn(T, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    kv_indices = torch.arange(16, 16 + P, device=DEV, dtype=torch.int64).repeat(chunk)
    kv_indptr = (torch.arange(chunk + 1, device=DEV) * P).to(torch.int32)
    qo = (torch.arange(chunk + 1, device=DEV) * E).to(torch.int64)

    def run(k_ext, v_ext, kbuf, vbuf, **kw):
        o = torch.empty_like(q)
        lse = torch.empty(T, HQ, dtype=torch.float32, device=DEV)
        ops.extend_attention_fwd(q, k_ext, v_ext, o, kbuf, vbuf, qo, kv_indptr, kv_indices, None, True, None, E,
                                 1.0, 1.0, sm_scale=D ** -0.5, lse_extend=lse, **kw)
        return o, lse

    # (1) normalisation
    c = torch.randn(HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    o, _ = run(ke, c.expand(T, HKV, D).contiguous(), kb, c.expand(pool, HKV, D).contiguous())
    want = c.repeat_interleave(HQ // HKV, dim=0).float()
    assert (o.float() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item() + 1e-6
    # (2) every request shares the prefix and q rows are independent: request 7 alone == request 7 in the batch
    o_all, lse_all = run(ke, ve, kb, vb)
    sl = slice(7 * E, 8 * E)
    o7 = torch.empty(E, HQ, D, dtype=torch.bfloat16, device=DEV)
    ops.extend_attention_fwd(q[sl], ke[sl], ve[sl], o7, kb, vb, qo[:2], kv_indptr[:2], kv_indices[:P], None, True,
                             None, E, 1.0, 1.0, sm_scale=D ** -0.5)
    assert torch.equal(o7, o_all[sl])
    # (3) cascade: (prefix only) merged with (new tokens only) == one pass
    o_p, l_p = run(ke, ve, kb, vb, skip_extend=True)
    o_e, l_e = run(ke, ve, kb, vb, skip_prefix=True)
    o_m, l_m = ops.merge_state(o_p, l_p, o_e, l_e)
    assert (o_m.float() - o_all.float()).abs().max().item() <= 3e-2
    assert (l_m - lse_all).abs().max().item() <= 2e-3
    # (4) the FIRST new token of a request sees the prefix + itself: that is a decode over P + 1 tokens
    first = torch.arange(chunk, device=DEV) * E
    kb2 = torch.cat([kb, ke[first]]).contiguous()                    # new rows appended at slots pool + i
    vb2 = torch.cat([vb, ve[first]]).contiguous()
    r2t = torch.zeros(chunk + 1, P + 1, dtype=torch.int32, device=DEV)
    r2t[1:, :P] = torch.arange(16, 16 + P, device=DEV, dtype=torch.int32)
    r2t[1:, P] = pool + torch.arange(chunk, device=DEV, dtype=torch.int32)
    od = torch.empty(chunk, HQ, D, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_fwd_paged(q[first].contiguous(), kb2, vb2, od, r2t, torch.arange(1, chunk + 1, device=DEV),
                                   torch.full((chunk,), P + 1, device=DEV, dtype=torch.int64), None, None, None, 1,
                                   D ** -0.5)
    assert (od.float() - o_all[first].float()).abs().max().item() <= 2e-2


def test_allocator_and_store_full_size_properties(ops):
    """bs 256 x 4096 tokens through the paged allocator kernels and the KV store."""
    from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator
    from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool

    size = BS * CTX
    pool = MHATokenToKVPool(size, PS, torch.bfloat16, HKV, D, 1, DEV)
    alloc = PagedTokenToKVPoolAllocator(size, PS, torch.bfloat16, DEV, pool)
    pre = torch.zeros(BS, dtype=torch.int64)
    seq = torch.full((BS,), CTX - 1, dtype=torch.int64)
    out = alloc.alloc_extend(pre.to(DEV), pre, seq.to(DEV), seq, torch.full((BS,), -1, dtype=torch.int64, device=DEV),
                             int(seq.sum()))
    assert out is not None and out.numel() == BS * (CTX - 1)
    o = out.view(BS, CTX - 1)
    assert int(out.min()) >= PS and int(out.max()) < size + PS              # page 0 never handed out
    assert torch.unique(out).numel() == out.numel()                          # no slot twice
    assert bool(((o[:, 1:] - o[:, :-1] == 1) | ((o[:, 1:] % PS == 0))).all())  # contiguous inside a page
    assert bool((o[:, 0] % PS == 0).all())                                   # requests start on a page
    # decode step: the last page of every request has exactly one free slot, so no new page is taken
    free_before = alloc.available_size()
    seq1 = seq + 1
    d = alloc.alloc_decode(seq1.to(DEV), seq1, o[:, -1].contiguous())
    assert torch.equal(d, o[:, -1] + 1) and alloc.available_size() == free_before
    # store round trip on a 64 Ki-token slice (checksum of checksums over the written rows)
    n = 65536
    g = torch.Generator(device=DEV).manual_seed(3)
    k = torch.randn(n, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(n, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    loc = out[torch.randperm(out.numel(), device=DEV, generator=g)[:n]]

    class L:
        layer_id = 0

    pool.set_kv_buffer(L, loc, k, v)
    kb, vb = pool.get_kv_buffer(0)
    assert torch.equal(kb[loc], k) and torch.equal(vb[loc], v)
    assert int(kb.view(torch.int16).to(torch.int64).sum()) == int(k.view(torch.int16).to(torch.int64).sum())
    assert pool.check_errors() == 0


def test_cascade_decode_full_size_properties(ops):
    """Config 3's radix-hit batch at full size (256 requests, 3584 shared + 512 private tokens, page 16):
    shared-prefix decode == per-request decode; the plan finds exactly the shared length; a one-hot V row inside
    the shared prefix comes back with the weight the per-request kernel gives it; ragged private lengths."""
    shared, uniq = 3584, 512
    rng = np.random.default_rng(5)
    n_pages = shared // PS + BS * (uniq // PS) + 1
    ids = rng.permutation(np.arange(1, n_pages))
    sh = (ids[: shared // PS, None] * PS + np.arange(PS)[None]).reshape(-1)
    priv = (ids[shared // PS:].reshape(BS, uniq // PS)[:, :, None] * PS + np.arange(PS)[None, None]).reshape(BS, -1)
    r2t_np = np.zeros((BS + 1, CTX), dtype=np.int32)
    r2t_np[1:, :shared] = sh[None]
    r2t_np[1:, shared:] = priv
    r2t = torch.from_numpy(r2t_np).to(DEV)
    rpi = torch.arange(1, BS + 1, dtype=torch.int64, device=DEV)
    lens_np = shared + rng.integers(1, uniq + 1, size=BS)
    lens_np[:4] = [shared + 1, shared + uniq, shared + 17, shared + 256]
    lens = torch.from_numpy(lens_np.astype(np.int64)).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    pool = n_pages * PS
    kb = torch.randn(pool, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    vb = torch.randn(pool, HKV, D, device=DEV, generator=g).to(torch.bfloat16)
    q = torch.randn(BS, HQ, D, device=DEV, generator=g).to(torch.bfloat16)
    sm = D ** -0.5
    ref = torch.empty_like(q)
    ops.decode_attention_fwd_paged(q, kb, vb, ref, r2t, rpi, lens, None, None, None, 1, sm, page_size=PS)
    cd = ops.CascadeDecode(BS, HQ, HKV, D, torch.bfloat16, DEV, max_shared=CTX)
    cd.plan(r2t, rpi, lens)
    assert cd.shared_len() == shared
    assert torch.equal(cd.suffix_lens[:BS].cpu(), torch.from_numpy((lens_np - shared).astype(np.int32)))
    o = torch.empty_like(q)
    cd(q, kb, vb, o, sm, page_size=PS)
    assert (o.float() - ref.float()).abs().max().item() <= 1.5e-2
    # one-hot V at a shared token: every request reads the same row, with its own softmax weight
    vb2 = torch.zeros_like(vb)
    tok = 1234
    vb2[int(sh[tok]), :, 7] = 1.0
    ops.decode_attention_fwd_paged(q, kb, vb2, ref, r2t, rpi, lens, None, None, None, 1, sm, page_size=PS)
    cd(q, kb, vb2, o, sm, page_size=PS)
    assert ref[:, :, 7].float().abs().max().item() > 0
    assert (o.float() - ref.float()).abs().max().item() <= 2e-3
    assert o[:, :, :7].abs().max().item() == 0 and o[:, :, 8:].abs().max().item() == 0
