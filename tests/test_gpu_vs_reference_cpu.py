"""Full-size parity against the REFERENCE'S OWN compiled kernels: BASELINE.json's shapes are far too
large for the fp64 oracle, but the reference's native CPU backend (`decode_attention_cpu`, `extend_attention_cpu`:
aot/csrc/cpu/decode.cpp:1586, extend.cpp:425, compiled in place by oracle/build_ref.py into oracle/_ref) runs them in a
fraction of a second on the GPU box's host cores.  Each case: one seeded ForwardBatch-shaped input, the reference kernel
in a CPU-only child process (tests/ref_cpu_child.py), the HIP path through the C ABI on identical tensors, and the
difference held to THE REFERENCE'S OWN test tolerance (test/registered/cpu/test_decode.py:266 atol 3e-2 / rtol 1e-6;
test_extend.py:344 atol 1e-2 / rtol 1e-2); the observed max-abs error is recorded through parity_util.check
(profiles/rNN_parity_errors.json).  Both sides are 16-bit kernels with their own summation orders, so the distance between
them is two bf16 roundings of the output, not the oracle's one.  Skipped when oracle/_ref was never built."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import parity_util as parity

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_ref():
    import glob

    return bool(glob.glob(os.path.join(ROOT, "oracle", "_ref", "rx_ref_cpu*.so")))


needs_ref = pytest.mark.skipif(not _have_ref(), reason="oracle/_ref not built (the reference's CPU kernels)")


def _run_child(kind, **kw):
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    out = tempfile.mkdtemp(prefix="rx_refcpu_", dir=base)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "ref_cpu_child.py"), "--kind", kind, "--out", out]
    for k, v in kw.items():
        cmd += ["--" + k.replace("_", "-"), str(v)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(os.cpu_count()), OMP_WAIT_POLICY="passive")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    except Exception:
        shutil.rmtree(out, ignore_errors=True)
        raise
    if r.returncode != 0:
        shutil.rmtree(out, ignore_errors=True)
        if r.returncode < 0 or r.returncode == 3:  # a signal (SIGILL: ISA this host lacks) or "not built": no verdict possible
            pytest.skip(f"reference CPU kernel child rc={r.returncode}: {r.stderr.strip()[-300:]}")
        raise AssertionError(f"reference CPU child failed rc={r.returncode}: {r.stderr[-2000:]}")
    info = json.loads(r.stdout.strip().splitlines()[-1])
    return out, info


def _load(out, name, bf16=False):
    a = np.load(os.path.join(out, name + ".npy"), mmap_mode="r")
    t = torch.from_numpy(np.array(a)).to(DEV)
    return t.view(torch.bfloat16) if bf16 else t


def _hnd_pool(out, ps):
    """The child's NHD pools [slots, Hkv, D] as HND pages [pages, Hkv, ps, D] (the MI355X default layout), same values."""
    kb, vb = _load(out, "k_buffer", True), _load(out, "v_buffer", True)
    hkv, d = kb.shape[1], kb.shape[2]
    kh = kb.view(-1, ps, hkv, d).permute(0, 2, 1, 3).contiguous()
    vh = vb.view(-1, ps, hkv, d).permute(0, 2, 1, 3).contiguous()
    del kb, vb
    return kh, vh


def _close(got, want, atol, rtol, tag):
    """torch.testing.assert_close's rule |got - want| <= atol + rtol * |want| (the reference tests' own), with the observed
    error recorded: the element with the worst error / bound ratio goes to parity_util.check (err, its bound)."""
    g, w = got.float(), want.float()
    diff = (g - w).abs()
    bound = atol + rtol * w.abs()
    ratio = diff / bound
    i = int(ratio.argmax().item())
    err_max = float(diff.max().item())
    cos = float(torch.nn.functional.cosine_similarity(g.flatten(), w.flatten(), dim=0).item())
    parity.check(float(diff.flatten()[i].item()), float(bound.flatten()[i].item()),
                 tag=f"{tag} vs reference CPU kernel (max-abs err {err_max:.3e}, cos {cos:.6f}, bound atol {atol} + rtol {rtol} |want|)")
    assert cos > 0.99, (tag, cos)
    return err_max


DECODE_CASES = [
    # name, bs, ctx, min_ctx, hq, hkv, the instance rx_last_dispatch must name
    ("configs2_llama8b_bs256_ctx4k", 256, 4096, 0, 32, 8, "decode_mfma_kernel<rx::BF16, 128, int, false, false, true, false>"),
    ("configs3_70b_tp8_shard_bs128_ctx4k", 128, 4096, 0, 8, 1, "decode_mfma_kernel<rx::BF16, 128, int"),
    ("configs1_llama8b_bs64_2k_plus_gen_ragged", 64, 2176, 2049, 32, 8, "decode_mfma_kernel<rx::BF16, 128, int"),
]


@needs_ref
@pytest.mark.parametrize("case", DECODE_CASES, ids=[c[0] for c in DECODE_CASES])
def test_decode_full_size_vs_reference_cpu_kernel(case):
    """One layer of a decode step, KV store included (the reference kernel stores k_new / v_new at `loc` itself; the HIP
    launch is the bench's fused-store instance on a page-16 shuffled HND pool, walking req_to_token)."""
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    name, bs, ctx, min_ctx, hq, hkv, expect = case
    D, ps = 128, 16
    out, info = _run_child("decode", bs=bs, ctx=ctx, min_ctx=min_ctx, hq=hq, hkv=hkv, d=D, ps=ps)
    try:
        kh, vh = _hnd_pool(out, ps)
        q, k_new, v_new = _load(out, "q", True), _load(out, "k_new", True), _load(out, "v_new", True)
        r2t, lens, loc = _load(out, "req_to_token"), _load(out, "seq_lens"), _load(out, "loc")
        want = _load(out, "out", True)
        k_st, v_st = _load(out, "k_stored", True), _load(out, "v_stored", True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    rpi = torch.arange(1, bs + 1, dtype=torch.int64, device=DEV)
    o = torch.empty_like(q)
    ops.decode_attention_fwd_paged(q, kh, vh, o, r2t, rpi, lens, None, None, None, 1, D ** -0.5, page_size=ps,
                                   kv_layout=ops.kv_layout_hnd(kh, vh), k_new=k_new, v_new=v_new)
    torch.cuda.synchronize()
    assert expect in rxlib.last_dispatch(), rxlib.last_dispatch()
    # the fused store wrote the same bytes to the same slots as the reference's store
    page, off = loc // ps, loc % ps
    assert torch.equal(kh[page, :, off], k_st) and torch.equal(vh[page, :, off], v_st)
    _close(o, want, 3e-2, 1e-6, name)          # test/registered/cpu/test_decode.py:266
    # the same launch through split-KV (8 splits + stage 2) against the same reference output
    ns = torch.full((bs,), 8, dtype=torch.int32, device=DEV)
    al = torch.empty(bs, hq, 8, D, dtype=torch.float32, device=DEV)
    lse = torch.empty(bs, hq, 8, dtype=torch.float32, device=DEV)
    o8 = torch.empty_like(q)
    ops.decode_attention_fwd_paged(q, kh, vh, o8, r2t, rpi, lens, al, lse, ns, 8, D ** -0.5, page_size=ps,
                                   kv_layout=ops.kv_layout_hnd(kh, vh))
    _close(o8, want, 3e-2, 1e-6, name + " [8 splits]")


EXTEND_CASES = [
    # name, requests, shared prefix, new tokens, hq, hkv
    ("configs2_chunk_32x_3584_shared_plus_512", 32, 3584, 512, 32, 8),
    ("configs1_prefill_8x2048_no_prefix", 8, 0, 2048, 32, 8),
    ("configs3_70b_tp8_shard_chunk_16x_3584_plus_512", 16, 3584, 512, 8, 1),
]


@needs_ref
@pytest.mark.parametrize("case", EXTEND_CASES, ids=[c[0] for c in EXTEND_CASES])
def test_extend_full_size_vs_reference_cpu_kernel(case):
    """One layer of the radix-hit extend: requests sharing one cached prefix (identical req_to_token prefixes) + their
    new tokens, causal; HIP: extend_attention_fwd on the page-16 shuffled HND pool (the bench's D = 128 MFMA kernel)."""
    from sglang_amd import lib as rxlib
    from sglang_amd import ops

    name, chunk, P, E, hq, hkv = case
    D, ps = 128, 16
    out, info = _run_child("extend", bs=chunk, prefix=P, extend=E, hq=hq, hkv=hkv, d=D, ps=ps)
    try:
        kh, vh = _hnd_pool(out, ps)
        q, ke, ve = _load(out, "q", True), _load(out, "k_extend", True), _load(out, "v_extend", True)
        r2t = _load(out, "req_to_token")
        want = _load(out, "out", True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
    T = chunk * E
    kv_indices = r2t[1:, :P].reshape(-1).to(torch.int64)
    kv_indptr = (torch.arange(chunk + 1, device=DEV) * P).to(torch.int32)
    qo_indptr = (torch.arange(chunk + 1, device=DEV) * E).to(torch.int64)
    o = torch.empty(T, hq, D, device=DEV, dtype=torch.bfloat16)
    ops.extend_attention_fwd(q, ke, ve, o, kh, vh, qo_indptr, kv_indptr, kv_indices, None, True, None, E, 1.0, 1.0,
                             sm_scale=D ** -0.5, page_size=ps, kv_layout=ops.kv_layout_hnd(kh, vh))
    torch.cuda.synchronize()
    assert "extend_mfma32_kernel<rx::BF16" in rxlib.last_dispatch(), rxlib.last_dispatch()
    _close(o, want, 1e-2, 1e-2, name)           # test/registered/cpu/test_extend.py:344
    if P:
        # the deterministic one-stage form over the unified kv list (prefix slots + the new tokens' slots) must agree too;
        # it reads the new tokens from the pool, so store them first
        loc = r2t[1:, P: P + E].reshape(-1).to(torch.int64)
        ops.store_cache_layout(ke, ve, ops.kv_layout_hnd(kh, vh), loc, hkv, D, D, size_limit=kh.shape[0] * ps)
        start = (torch.arange(chunk, device=DEV) * E).to(torch.int32)
        ext = torch.full((chunk,), E, dtype=torch.int32, device=DEV)
        u_indptr, u_indices, prefix_lens = ops.build_unified_kv_indices(kv_indptr, kv_indices, start, ext, loc, chunk,
                                                                        max_tokens_per_request=P + E)
        o_u = torch.empty_like(o)
        ops.extend_attention_fwd_unified(q, o_u, kh, vh, 1.0, 1.0, qo_indptr, u_indptr, u_indices, prefix_lens, E,
                                         sm_scale=D ** -0.5, is_causal=True, page_size=ps, kv_layout=ops.kv_layout_hnd(kh, vh))
        _close(o_u, want, 1e-2, 1e-2, name + " [unified one-stage]")
