"""Shared-prefix (cascade) decode: rx_shared_prefix_plan + extend-over-shared-rows + suffix decode
+ stage-2 merge must equal plain decode attention (fp64 oracle) for any batch, whatever the common prefix is."""
import numpy as np
import pytest

import parity_util as parity
import torch

from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from sglang_amd import ops as _ops

    return _ops


def _bits(t):
    if t.dtype == torch.bfloat16:
        return t.detach().cpu().contiguous().view(torch.uint16).numpy()
    return t.detach().cpu().numpy()


def _table(rng, shared, lens, page_size, ctx):
    """req_to_token rows (row 0 = padding) whose first `shared` slots are identical; the rest private pages."""
    bs = len(lens)
    npages_shared = -(-shared // page_size)
    pages_priv = [max(0, -(-int(n) // page_size) - shared // page_size) for n in lens]
    n_pages = npages_shared + sum(pages_priv) + 2
    ids = rng.permutation(np.arange(1, n_pages))
    sh = np.concatenate([np.arange(p * page_size, (p + 1) * page_size) for p in ids[:npages_shared]] or
                        [np.zeros(0, np.int64)])[:shared]
    r2t = np.zeros((bs + 1, ctx), dtype=np.int32)
    pi = npages_shared
    for i, n in enumerate(lens):
        priv = np.concatenate([np.arange(p * page_size, (p + 1) * page_size)
                               for p in ids[pi: pi + pages_priv[i]]] or [np.zeros(0, np.int64)])
        pi += pages_priv[i]
        # a shared prefix that ends inside a page: the private part starts on a fresh page, like a
        # radix split at a page boundary would; the tail of the shared page is simply unused here
        row = np.concatenate([sh, priv])[: int(n)]
        r2t[i + 1, : len(row)] = row
    return r2t, n_pages * page_size


def _oracle_plan(r2t, rpi, lens, max_shared, min_shared):
    rows = r2t[rpi]
    m = int(min(int(lens.min()), max_shared))
    L = 0
    while L < m and np.all(rows[:, L] == rows[0, L]):
        L += 1
    return 0 if L < min_shared else L


CASES = [
    # bs, hq, hkv, d, page, shared(target), lens, min_shared
    (8, 8, 2, 128, 16, 512, [600, 513, 700, 640, 1000, 512 + 17, 530, 800], 64),
    (5, 4, 4, 128, 1, 100, [101, 150, 333, 100, 129], 1),          # one request is the prefix itself
    (3, 16, 2, 64, 32, 256, [300, 257, 512], 64),
    (4, 8, 1, 128, 16, 0, [100, 200, 300, 64], 64),                # nothing shared -> plain decode
    (6, 8, 2, 128, 16, 48, [100, 200, 300, 64, 90, 77], 64),       # shared but below min_shared -> L = 0
    (1, 8, 2, 128, 16, 0, [777], 16),                              # bs 1: everything is "shared"
    (130, 4, 1, 128, 64, 1024, None, 256),                         # > 128 queries per head (two M blocks)
    (40, 8, 2, 256, 16, 600, None, 64),                            # head dim 256: extend_d256_kernel + the MFMA decode kernel
    (6, 16, 2, 256, 16, 300, [301, 420, 333, 300, 512, 400], 64),  # (few rows per kv head: extend_nd_kernel in phase 1)
    (24, 8, 8, 96, 16, 400, None, 64),                             # head dim 96 (Phi-3-class)
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_cascade_decode_matches_oracle(ops, case, dtype):
    bs, hq, hkv, d, page, shared, lens, min_shared = CASES[case]
    rng = np.random.default_rng(500 + case)
    if lens is None:
        lens = shared + rng.integers(1, 200, size=bs)
    lens = np.asarray(lens, dtype=np.int64)
    ctx = int(lens.max()) + page
    r2t, pool = _table(rng, shared, lens, page, ctx)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    g = torch.Generator().manual_seed(case)
    kb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    vb = torch.randn(pool, hkv, d, generator=g).to(dtype)
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    sinks = torch.randn(hq, generator=g) if case % 2 else None
    cap = 30.0 if case == 2 else 0.0
    ks, vs = (0.9, 1.1) if case % 3 == 0 else (1.0, 1.0)
    sm = d ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want, absw = parity.want_and_absw(orc.decode_attention, (_bits(q), _bits(kb), _bits(vb), kv_indptr, kv_indices, sm), (2,),
                                      k_scale=ks, v_scale=vs, logit_cap=cap, sinks=None if sinks is None else sinks.numpy())
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    cd = ops.CascadeDecode(bs, hq, hkv, d, dtype, DEV, max_shared=ctx, min_shared=min_shared,
                           num_chunks=[None, 1, 3][case % 3], overlap=bool(case % 2 == 0))
    r2t_d, rpi_d, lens_d = T(r2t), T(rpi if case % 2 else rpi.astype(np.int32)), T(lens if case % 2 else lens.astype(np.int32))
    cd.plan(r2t_d, rpi_d, lens_d)
    L = _oracle_plan(r2t, rpi, lens, ctx, min_shared)
    assert cd.shared_len() == L
    if case not in (3, 4):
        assert L >= shared
    ci = cd.chunk_indptr.cpu().numpy()
    assert ci[0] == 0 and ci[-1] == L and np.all(np.diff(ci) >= 0)
    assert np.array_equal(cd.shared_indices.cpu().numpy()[:L], r2t[1, :L])
    assert np.array_equal(cd.kv_start[:bs].cpu().numpy(), np.full(bs, L))
    assert np.array_equal(cd.suffix_lens[:bs].cpu().numpy(), lens - L)
    o = torch.zeros(bs, hq, d, dtype=dtype, device=DEV)
    cd(q.to(DEV), kb.to(DEV), vb.to(DEV), o, sm, ks, vs, cap, None if sinks is None else sinks.to(DEV),
       page_size=page)
    # (two 16-bit roundings: the chunk partials cross 16-bit buffers before the merge -- 2 ulp)
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("cascade", case), ulps=2, absw=absw)


MLA_CASES = [
    # bs, hq, page, shared(target), lens, min_shared
    (8, 16, 16, 512, [600, 513, 700, 640, 1000, 512 + 17, 530, 800], 64),
    (5, 16, 1, 100, [101, 150, 333, 100, 129], 1),       # one request is the prefix itself (empty suffix)
    (4, 16, 16, 0, [100, 200, 300, 64], 64),             # nothing shared -> plain MLA decode
    (20, 8, 64, 1500, None, 256),                        # 160 (query, head) rows per chunk: two workgroups each
    (300, 16, 16, 256, None, 64),                        # a batch that fills the chip: one suffix split per request
]


@pytest.mark.parametrize("case", range(len(MLA_CASES)))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_cascade_decode_latent_mla(ops, case, dtype):
    """The cascade on latent MLA rows (q 576 / v 512 over one kv head): phase 1 = rx::extend_mla_kernel over the shared
    rows with all (query, head) pairs as its rows, phase 2 = the MLA decode kernel over the suffixes (kv_start on the
    req_to_token lookup), stage 2 merges -- equals plain decode attention (fp64 oracle)."""
    bs, hq, page, shared, lens, min_shared = MLA_CASES[case]
    dk, dv = 576, 512
    rng = np.random.default_rng(900 + case)
    if lens is None:
        lens = shared + rng.integers(1, 200, size=bs)
    lens = np.asarray(lens, dtype=np.int64)
    ctx = int(lens.max()) + page
    r2t, pool = _table(rng, shared, lens, page, ctx)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    g = torch.Generator().manual_seed(case)
    kb = (torch.randn(pool, 1, dk, generator=g) * 0.5).to(dtype)
    q = torch.randn(bs, hq, dk, generator=g).to(dtype)
    sinks = torch.randn(hq, generator=g) if case % 2 else None
    ks, vs = (0.9, 1.1) if case == 0 else (1.0, 1.0)
    sm = 192 ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want, absw = parity.want_and_absw(orc.decode_attention, (_bits(q), _bits(kb), _bits(kb[..., :dv].contiguous()), kv_indptr,
                                                             kv_indices, sm), (2,), k_scale=ks, v_scale=vs,
                                      sinks=None if sinks is None else sinks.numpy())
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    cd = ops.CascadeDecode(bs, hq, 1, dk, dtype, DEV, max_shared=ctx, min_shared=min_shared, v_head_dim=dv,
                           num_chunks=[None, 1, 3, None, None][case])
    cd.plan(T(r2t), T(rpi), T(lens))
    L = _oracle_plan(r2t, rpi, lens, ctx, min_shared)
    assert cd.shared_len() == L
    o = torch.zeros(bs, hq, dv, dtype=dtype, device=DEV)
    kbd = kb.to(DEV)
    cd(q.to(DEV), kbd, kbd[..., :dv], o, sm, ks, vs, 0.0, None if sinks is None else sinks.to(DEV), page_size=page)
    parity.check_out(o.float().cpu().numpy(), want, dtype, ("cascade mla", case), ulps=2, absw=absw)  # (two 16-bit roundings: the chunk partials cross 16-bit buffers before the merge -- 2 ulp)


def test_cascade_decode_fp8_pool(ops):
    """fp8 e4m3fn pool: both phases read the same bytes as plain decode."""
    bs, hq, hkv, d, page, shared = 16, 8, 2, 128, 16, 768
    rng = np.random.default_rng(9)
    lens = (shared + rng.integers(1, 300, size=bs)).astype(np.int64)
    ctx = int(lens.max()) + page
    r2t, pool = _table(rng, shared, lens, page, ctx)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    g = torch.Generator().manual_seed(3)
    kb = torch.randn(pool, hkv, d, generator=g).to(torch.float8_e4m3fn)
    vb = torch.randn(pool, hkv, d, generator=g).to(torch.float8_e4m3fn)
    q = torch.randn(bs, hq, d, generator=g).to(torch.bfloat16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    sm, ks, vs = d ** -0.5, 0.5, 2.0
    qd, kbd, vbd = q.to(DEV), kb.view(torch.uint8).to(DEV), vb.view(torch.uint8).to(DEV)
    ref = torch.zeros(bs, hq, d, dtype=torch.bfloat16, device=DEV)
    ops.decode_attention_fwd_paged(qd, kbd, vbd, ref, T(r2t), T(rpi), T(lens), None, None, None, 1, sm, ks, vs,
                                   page_size=page)
    cd = ops.CascadeDecode(bs, hq, hkv, d, torch.bfloat16, DEV, max_shared=ctx, min_shared=64)
    cd.plan(T(r2t), T(rpi), T(lens))
    assert cd.shared_len() >= shared
    o = torch.zeros_like(ref)
    cd(qd, kbd, vbd, o, sm, ks, vs, page_size=page)
    assert (o.float() - ref.float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("page_size", [1, 16])
def test_backend_cascade_decode_on_radix_hit_batch(page_size):
    """HipRadixAttnBackend(cascade_decode=True) driven like the runner drives a backend: requests whose
    req_to_token rows start with the same slots (a radix hit), private suffixes, one decode step with the KV
    store -- against the torch-native semantics of the oracle."""
    from sglang_amd.attention.backend import HipRadixAttnBackend
    from sglang_amd.forward_batch import ForwardBatch
    from tests.test_gpu_backend import _Harness

    hq, hkv, d, bs, shared = 8, 2, 128, 9, 512
    hs = _Harness(page_size, hq, hkv, d, torch.bfloat16, "shuffled_pages" if page_size > 1 else "contiguous",
                  "paged", max_ctx=1200, max_reqs=16)
    hs.backend = HipRadixAttnBackend(_runner_of(hs), cascade_decode=True, cascade_min_bs=2, cascade_min_shared=64)
    rows = hs.r2t.alloc(bs)
    hs.fill_prefix(rows[:1], [shared])
    for r in rows[1:]:  # radix hit: the cached prefix's slots are written into the new request's row
        hs.r2t.req_to_token[r, :shared] = hs.r2t.req_to_token[rows[0], :shared]
    priv = [1, 40, 129, 16, 300, 77, 5, 250, 64]
    prefix_lens = [shared + p for p in priv]
    loc = hs.alloc_extend(rows, [shared] * bs, prefix_lens)
    hs.pool.set_kv_buffer(hs.layer, loc, hs.rand(sum(priv), hkv, d), hs.rand(sum(priv), hkv, d))
    seq_lens = [p + 1 for p in prefix_lens]
    seq_t = torch.tensor(seq_lens, dtype=torch.int64)
    rpi = torch.tensor(rows, dtype=torch.int64, device=DEV)
    if page_size == 1:
        loc = hs.alloc.alloc(bs)
    else:
        last = torch.tensor([int(hs.r2t.req_to_token[r, p - 1]) for r, p in zip(rows, prefix_lens)],
                            dtype=torch.int64, device=DEV)
        loc = hs.alloc.alloc_decode(seq_t.to(DEV), seq_t, last)
    hs.r2t.req_to_token[rpi, torch.tensor(prefix_lens, device=DEV)] = loc.to(torch.int32)
    q, k, v = hs.rand(bs, hq * d), hs.rand(bs, hkv * d), hs.rand(bs, hkv * d)
    fb = ForwardBatch.for_decode(rpi, seq_t.to(DEV), loc, seq_t)
    hs.backend.init_forward_metadata(fb)
    o = hs.layer(q, k, v, fb, hs.backend)
    assert hs.backend._cascade is not None and hs.backend._cascade.shared_len() == shared
    kb, vb = hs.pool.get_kv_buffer(0)
    want, absw = parity.want_and_absw(orc.sdpa_decode_req_to_token, (_bits(q.view(bs, hq, d)), _bits(kb), _bits(vb),
                                                                     _bits(hs.r2t.req_to_token), np.array(rows),
                                                                     np.array(seq_lens), d ** -0.5), (2,))
    got = o.view(bs, hq, d).float().cpu().numpy().astype(np.float64)
    assert hs.pool.check_errors() == 0
    parity.check_out(got, want, o.dtype, "backend cascade", ulps=2, absw=absw)  # (two 16-bit roundings: the chunk partials cross 16-bit buffers before the merge -- 2 ulp)


def _runner_of(hs):
    class MC:
        num_attention_heads, num_key_value_heads, context_len = hs.hq, hs.hkv, hs.r2t.req_to_token.shape[1]

    class MR:
        device = DEV
        req_to_token_pool = hs.r2t
        token_to_kv_pool = hs.pool
        token_to_kv_pool_allocator = hs.alloc
        model_config = MC
        page_size = hs.ps
        dtype = hs.dtype

        class server_args:
            triton_attention_num_kv_splits = 8

    return MR


def test_cascade_plan_and_layer_replay_under_hip_graph(ops):
    """The plan reads the page table on the device: a captured {plan, layer} graph must follow a CHANGED
    common prefix (and changed lengths) on replay, with no host involvement."""
    bs, hq, hkv, d, page = 12, 8, 2, 128, 16
    rng = np.random.default_rng(77)
    ctx = 1024
    lens_a = (512 + rng.integers(1, 200, size=bs)).astype(np.int64)
    lens_b = (256 + rng.integers(1, 600, size=bs)).astype(np.int64)
    r2t_a, pool_a = _table(rng, 512, lens_a, page, ctx)
    r2t_b, pool_b = _table(rng, 256, lens_b, page, ctx)
    pool = max(pool_a, pool_b)
    g = torch.Generator().manual_seed(1)
    kb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    vb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
    q = torch.randn(bs, hq, d, generator=g).to(torch.bfloat16).to(DEV)
    o = torch.zeros_like(q)
    r2t = torch.from_numpy(r2t_a).to(DEV)
    lens = torch.from_numpy(lens_a).to(DEV)
    rpi = torch.arange(1, bs + 1, dtype=torch.int64, device=DEV)
    sm = d ** -0.5
    cd = ops.CascadeDecode(bs, hq, hkv, d, torch.bfloat16, DEV, max_shared=ctx, min_shared=64)

    def step():
        cd.plan(r2t, rpi, lens)
        cd(q, kb, vb, o, sm, page_size=page)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()  # warm the allocator and the scratch buffers
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for r2t_np, lens_np, want_shared in ((r2t_b, lens_b, 256), (r2t_a, lens_a, 512)):
        r2t.copy_(torch.from_numpy(r2t_np).to(DEV))
        lens.copy_(torch.from_numpy(lens_np).to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        assert cd.shared_len() >= want_shared
        ref = torch.zeros_like(q)
        ops.decode_attention_fwd_paged(q, kb, vb, ref, r2t, rpi, lens, None, None, None, 1, sm, page_size=page)
        assert (o.float() - ref.float()).abs().max().item() <= 1.5e-2


def test_cascade_decode_hnd_pool(ops):
    """HND pool ([pages, Hkv, page, D], the bench's default layout): both phases address it through rx_kv_layout."""
    bs, hq, hkv, d, page, shared = 24, 8, 2, 128, 16, 1024
    rng = np.random.default_rng(21)
    lens = (shared + rng.integers(1, 400, size=bs)).astype(np.int64)
    ctx = int(lens.max()) + page
    r2t, pool = _table(rng, shared, lens, page, ctx)
    rpi = np.arange(1, bs + 1, dtype=np.int64)
    g = torch.Generator().manual_seed(8)
    kb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16)      # NHD data ...
    vb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16)
    q = torch.randn(bs, hq, d, generator=g).to(torch.bfloat16)
    kh = kb.view(pool // page, page, hkv, d).permute(0, 2, 1, 3).contiguous().to(DEV)   # ... re-laid as HND
    vh = vb.view(pool // page, page, hkv, d).permute(0, 2, 1, 3).contiguous().to(DEV)
    lay = ops.kv_layout_hnd(kh, vh)
    sm = d ** -0.5
    kv_indptr, kv_indices = orc.build_kv_indices(r2t, rpi, lens)
    want, absw = parity.want_and_absw(orc.decode_attention, (_bits(q), _bits(kb), _bits(vb), kv_indptr, kv_indices, sm), (2,))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    cd = ops.CascadeDecode(bs, hq, hkv, d, torch.bfloat16, DEV, max_shared=ctx, min_shared=64)
    cd.plan(T(r2t), T(rpi), T(lens))
    assert cd.shared_len() >= shared
    o = torch.zeros(bs, hq, d, dtype=torch.bfloat16, device=DEV)
    cd(q.to(DEV), kh, vh, o, sm, page_size=page, kv_layout=lay)
    parity.check_out(o.float().cpu().numpy(), want, torch.bfloat16, "cascade hnd", ulps=2, absw=absw)  # (two 16-bit roundings: the chunk partials cross 16-bit buffers before the merge -- 2 ulp)


def test_cascade_chunk_count_follows_the_batch(ops):
    """A CascadeDecode sized for a large pool (max_bs 512) planning a small batch picks the chunk count of THAT
    batch (phase 1 must still fill the chip) and stays correct; then a large batch on the same object."""
    hq, hkv, d, page, shared = 8, 2, 128, 16, 2048
    cd = ops.CascadeDecode(512, hq, hkv, d, torch.bfloat16, DEV, max_shared=4096, min_shared=64)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    seen = []
    for bs in (6, 300):
        rng = np.random.default_rng(bs)
        lens = (shared + rng.integers(1, 200, size=bs)).astype(np.int64)
        ctx = int(lens.max()) + page
        r2t, pool = _table(rng, shared, lens, page, ctx)
        rpi = np.arange(1, bs + 1, dtype=np.int64)
        g = torch.Generator().manual_seed(bs)
        kb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
        vb = torch.randn(pool, hkv, d, generator=g).to(torch.bfloat16).to(DEV)
        q = torch.randn(bs, hq, d, generator=g).to(torch.bfloat16).to(DEV)
        sm = d ** -0.5
        ref = torch.zeros_like(q)
        ops.decode_attention_fwd_paged(q, kb, vb, ref, T(r2t), T(rpi), T(lens), None, None, None, 1, sm, page_size=page)
        cd.plan(T(r2t), T(rpi), T(lens))
        seen.append(cd.num_chunks)
        assert cd.shared_len() >= shared and cd.chunk_indptr.numel() == cd.num_chunks + 1
        o = torch.zeros_like(q)
        cd(q, kb, vb, o, sm, page_size=page)
        assert (o.float() - ref.float()).abs().max().item() <= 1.5e-2
    assert seen[0] > seen[1] >= 1   # 16 chunks for 6 requests, fewer for 300
