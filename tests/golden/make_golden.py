"""Generate golden fixtures by RUNNING THE REFERENCE in the build container.

    cd /root/repo && python tests/golden/make_golden.py

Imports the reference's own kernels from /root/reference (Triton kernels under
TRITON_INTERPRET=1 on CPU tensors; the paged/token allocators on device="cpu")
and stores (inputs, outputs) as small .npz files next to this script.  Only the
.npz data travels to the GPU box; the tests never import the reference.

Fixture families:
  F1 allocator op sequences      -> alloc_*.npz
  F2 kv-indices                  -> kv_indices.npz
  F3 num_kv_splits               -> kv_splits.npz
  F4 store (torch index_put fallback of memory_pool.py:189-192) -> store_kv.npz
  F5 decode attention (Triton decode_attention_fwd) -> decode_*.npz
  F6 extend attention (Triton extend_attention_fwd) -> extend_*.npz
  F7 radix tree op sequences (reference RadixCache, recording allocator) -> radix_sequences.json
  F15 decode context parallel: per-rank lengths, local kv indices, the cross-rank LSE merge -> dcp.npz
  F14 the reference's torch-native SDPA helpers (a14) on fp32 tensors -> torch_native.npz
  F13 scheduler flow (reference RadixCache request hooks + allocators + req_to_token rows) -> scheduler_flow.json
  F8 bf16 decode/extend from the compiled reference C++ CPU kernels -> cpu_native.npz
  F10 decode with the xai temperature -> decode_xai.npz
  F12 unified one-stage extend (deterministic inference) -> extend_unified.npz
  F11 rotary embedding (torch-native apply_rotary_emb) -> rope.npz
  F9 extend with custom (tree) masks, sliding window (+ window_kv_offsets), xai temperature -> extend_mask.npz
  F16 ROCm MLA decode with fused RoPE (stage-1 kernel of rocm_mla_decode_rope.py, LSE-merged) -> mla_rope.npz
  F17 QK-norm + RoPE (RMSNorm.forward_native + apply_rotary_emb, the pair the reference tests its fused kernel against) -> qknorm_rope.npz
  F19 relative-position score bias (score_mod = relative_bias_score_mod, aux_tensors = [rel_logits]) through the extend,
      unified-extend and decode kernels -> score_bias.npz
  F20 the unified kv list of the one-stage extend (build_unified_kv_indices) -> unified_kv_indices.npz
  F21 fused_fp8_qkv_kv_cache: the reference test's expectation formula (torch CPU casts) -> fused_fp8_qkv.npz
  F18 EAGLE multi-step draft decode: per-step kv_indices / kv_indptr of the top-k branches (generate_draft_decode_kv_indices) -> draft_kv_indices.npz
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sglang.kernels.ops.attention.decode_attention import decode_attention_fwd  # noqa: E402
from sglang.kernels.ops.attention.extend_attention import extend_attention_fwd  # noqa: E402
from sglang.kernels.ops.attention.metadata import get_num_kv_splits_triton  # noqa: E402
from sglang.kernels.ops.kvcache.kv_indices import (  # noqa: E402
    create_flashinfer_kv_indices_triton,
)
from sglang.srt.mem_cache.allocator.paged import PagedTokenToKVPoolAllocator  # noqa: E402
from sglang.srt.mem_cache.allocator.token import TokenToKVPoolAllocator  # noqa: E402


def bits(t: torch.Tensor) -> np.ndarray:
    """bf16 tensor -> uint16 bit patterns; others -> numpy."""
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.uint16).numpy().copy()
    return t.contiguous().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


# ------------------------------------------------------------------ F1
def gen_allocator(page_size: int, need_sort: bool, seed: int):
    """Drive the reference allocator with a small scheduler-like workload and
    log every call's inputs, result and the free list afterwards."""
    rng = np.random.default_rng(seed)
    size = 64 * max(page_size, 4)
    if page_size == 1:
        alloc = TokenToKVPoolAllocator(size, torch.bfloat16, "cpu", None, need_sort)
    else:
        alloc = PagedTokenToKVPoolAllocator(
            size, page_size, torch.bfloat16, "cpu", None, need_sort
        )
    log = []
    reqs = {}  # rid -> list of slots
    next_rid = 0

    def snap():
        return alloc.free_pages.numpy().tolist(), alloc.release_pages.numpy().tolist()

    for step in range(40):
        op = rng.choice(["extend", "decode", "free", "free_segment", "sort", "group_free"],
                        p=[0.3, 0.3, 0.15, 0.1, 0.05, 0.1])
        if op == "extend" or not reqs:
            bs = int(rng.integers(1, 4))
            rids, prefix, seqs, last = [], [], [], []
            for _ in range(bs):
                if reqs and rng.random() < 0.5:
                    rid = int(rng.choice(list(reqs.keys())))
                    if rid in rids:
                        continue
                else:
                    rid = next_rid
                    next_rid += 1
                    reqs[rid] = []
                pre = len(reqs[rid])
                ext = int(rng.integers(1, 3 * page_size + 3))
                rids.append(rid); prefix.append(pre); seqs.append(pre + ext)
                last.append(reqs[rid][-1] if pre else -1)
            if page_size == 1:
                need = int(sum(s - p for s, p in zip(seqs, prefix)))
                out = alloc.alloc(need)
                res = None if out is None else out.numpy().tolist()
                log.append(dict(op="alloc", need=need, out=res, free=snap()))
            else:
                pl = torch.tensor(prefix, dtype=torch.int64)
                sl = torch.tensor(seqs, dtype=torch.int64)
                ll = torch.tensor(last, dtype=torch.int64)
                out = alloc.alloc_extend(pl, pl, sl, sl, ll, int((sl - pl).sum()))
                res = None if out is None else out.numpy().tolist()
                log.append(dict(op="alloc_extend", prefix_lens=prefix, seq_lens=seqs,
                                last_loc=last, out=res, free=snap()))
            if res is not None:
                o = 0
                for rid, p, s in zip(rids, prefix, seqs):
                    reqs[rid].extend(res[o : o + s - p]); o += s - p
            else:
                for rid in rids:
                    if not reqs[rid]:
                        del reqs[rid]
        elif op == "decode":
            rids = [r for r in reqs if reqs[r]]
            if not rids:
                continue
            rids = list(rng.choice(rids, size=min(len(rids), int(rng.integers(1, 5))), replace=False))
            seqs = [len(reqs[r]) + 1 for r in rids]
            last = [reqs[r][-1] for r in rids]
            if page_size == 1:
                out = alloc.alloc(len(rids))
                res = None if out is None else out.numpy().tolist()
                log.append(dict(op="alloc", need=len(rids), out=res, free=snap()))
            else:
                sl = torch.tensor(seqs, dtype=torch.int64)
                out = alloc.alloc_decode(sl, sl, torch.tensor(last, dtype=torch.int64))
                res = None if out is None else out.numpy().tolist()
                log.append(dict(op="alloc_decode", seq_lens=seqs, last_loc=last,
                                out=res, free=snap()))
            if res is not None:
                for r, x in zip(rids, res):
                    reqs[int(r)].append(x)
        elif op == "free":
            rid = int(rng.choice(list(reqs.keys())))
            idx = reqs.pop(rid)
            alloc.free(torch.tensor(idx, dtype=torch.int64))
            log.append(dict(op="free", idx=idx, free=snap()))
        elif op == "free_segment" and page_size > 1:
            rid = int(rng.choice(list(reqs.keys())))
            row = reqs[rid]
            if len(row) < page_size + 1:
                continue
            # free the tail from a page boundary onward (a page is freed by one call)
            start = (int(rng.integers(0, len(row))) // page_size) * page_size
            idx = row[start:]
            reqs[rid] = row[:start]
            if not reqs[rid]:
                del reqs[rid]
            alloc.free_segment(torch.tensor(idx, dtype=torch.int64), start_pos=start)
            log.append(dict(op="free_segment", idx=idx, start_pos=start, free=snap()))
        elif op == "sort":
            alloc.merge_and_sort_free()
            log.append(dict(op="merge_and_sort_free", free=snap()))
        elif op == "group_free" and len(reqs) >= 2:
            alloc.free_group_begin()
            freed = []
            for rid in list(reqs.keys())[:2]:
                idx = reqs.pop(rid)
                alloc.free(torch.tensor(idx, dtype=torch.int64))
                freed.append(idx)
            alloc.free_group_end()
            log.append(dict(op="free_group", idx=freed, free=snap()))
    return dict(page_size=page_size, need_sort=need_sort, size=size, log=log)


def f1():
    cases = []
    for ps in (1, 4, 16, 32):
        for need_sort in (False, True):
            cases.append(gen_allocator(ps, need_sort, seed=100 + ps))
    with open(os.path.join(HERE, "alloc_sequences.json"), "w") as f:
        json.dump(cases, f)
    print("wrote alloc_sequences.json", [len(c["log"]) for c in cases])


# ------------------------------------------------------------------ F2
def f2():
    rng = np.random.default_rng(7)
    out = {}
    for ci, (batch, max_batch, ctx) in enumerate([(1, 8, 64), (5, 16, 300), (37, 64, 700)]):
        r2t = torch.from_numpy(rng.integers(0, 1 << 20, size=(max_batch, ctx)).astype(np.int32))
        rpi = torch.from_numpy(rng.choice(max_batch, size=batch, replace=False).astype(np.int32))
        lens = torch.from_numpy(rng.integers(0, ctx // 2, size=batch).astype(np.int32))
        start = torch.from_numpy(rng.integers(0, ctx // 2, size=batch).astype(np.int32))
        for use_start in (False, True):
            kv_indptr = torch.zeros((batch + 1,), dtype=torch.int32)
            kv_indptr[1:] = torch.cumsum(lens, 0)
            kv_indices = torch.full((int(kv_indptr[-1]),), -1, dtype=torch.int64)
            create_flashinfer_kv_indices_triton[(batch,)](
                r2t, rpi, lens, kv_indptr, start if use_start else None, kv_indices, r2t.stride(0)
            )
            tag = f"c{ci}_{int(use_start)}"
            out.update({f"{tag}_req_to_token": r2t.numpy(), f"{tag}_req_pool_indices": rpi.numpy(),
                        f"{tag}_lens": lens.numpy(), f"{tag}_start": start.numpy(),
                        f"{tag}_kv_indptr": kv_indptr.numpy(), f"{tag}_kv_indices": kv_indices.numpy()})
    save("kv_indices.npz", **out)


# ------------------------------------------------------------------ F3
def f3():
    rng = np.random.default_rng(11)
    rows = []
    import triton
    for (hq, hkv) in [(32, 8), (16, 4), (8, 2), (4, 1), (12, 12), (128, 1), (64, 8)]:
        for bs in (1, 3, 64, 256, 300):
            for kind in ("uniform", "ragged", "short"):
                if kind == "uniform":
                    sl = np.full((bs,), 4096)
                elif kind == "ragged":
                    sl = rng.integers(1, 8192, size=bs)
                else:
                    sl = rng.integers(1, 64, size=bs)
                for max_splits in (8, 16):
                    for num_group in (1, 2):
                        seq = torch.from_numpy(sl.astype(np.int32))
                        outp = torch.zeros((bs * num_group,), dtype=torch.int32)
                        sched = 256 if bs < 256 else triton.next_power_of_2(bs)
                        get_num_kv_splits_triton[(1,)](outp, seq, bs, num_group, hq, hkv,
                                                       max_splits, 256, MAX_NUM_SEQ=sched)
                        rows.append(dict(hq=hq, hkv=hkv, max_splits=max_splits, num_group=num_group,
                                         cores=256, seq_lens=sl.tolist(), out=outp.numpy().tolist()))
    with open(os.path.join(HERE, "kv_splits.json"), "w") as f:
        json.dump(rows, f)
    print("wrote kv_splits.json", len(rows))


# ------------------------------------------------------------------ F4
def f4():
    g = torch.Generator().manual_seed(3)
    out = {}
    for ci, (n, rows, hkv, d, idt) in enumerate([(7, 40, 2, 64, torch.int64), (33, 200, 8, 128, torch.int32)]):
        k = torch.randn(n, hkv, d, generator=g).to(torch.bfloat16)
        v = torch.randn(n, hkv, d, generator=g).to(torch.bfloat16)
        kc = torch.randn(rows, hkv, d, generator=g).to(torch.bfloat16)
        vc = torch.randn(rows, hkv, d, generator=g).to(torch.bfloat16)
        loc = (torch.randperm(rows - 1, generator=g)[:n] + 1).to(idt)  # unique, never slot 0
        kc0, vc0 = kc.clone(), vc.clone()
        # the reference's torch fallback (memory_pool.py:189-192)
        kc[loc.long()] = k
        vc[loc.long()] = v
        out.update({f"c{ci}_k": bits(k), f"c{ci}_v": bits(v), f"c{ci}_kc_in": bits(kc0),
                    f"c{ci}_vc_in": bits(vc0), f"c{ci}_loc": loc.numpy(),
                    f"c{ci}_kc_out": bits(kc), f"c{ci}_vc_out": bits(vc)})
    save("store_kv.npz", **out)


# ------------------------------------------------------------------ F5
def run_decode(q, kb, vb, kv_indptr, kv_indices, num_kv_splits, max_kv_splits, sm_scale,
               logit_cap=0.0):
    b, hq, _ = q.shape
    dv = vb.shape[-1]
    o = torch.zeros(b, hq, dv, dtype=q.dtype)
    attn_logits = torch.zeros(b, hq, max_kv_splits, dv, dtype=torch.float32)
    attn_lse = torch.zeros(b, hq, max_kv_splits, dtype=torch.float32)
    decode_attention_fwd(q, kb, vb, o, kv_indptr, kv_indices, attn_logits, attn_lse,
                         num_kv_splits, max_kv_splits, sm_scale, 1.0, 1.0, logit_cap=logit_cap)
    return o, attn_logits, attn_lse


def f5():
    torch.manual_seed(42)
    # Triton's CPU interpreter has no bf16 scalar support (InterpreterBuilder.get_bf16
    # is missing), so the Triton-derived goldens use fp16; bf16 goldens come from the
    # compiled reference C++ CPU kernels (oracle/_ref, see f8).
    dtype = torch.float16
    cases = {}
    # (a) the reference's own decode test configs (test_triton_attention_kernels.py:563-573)
    for ci, (B, HQ, HKV, D) in enumerate([(2, 4, 4, 64), (2, 4, 2, 64), (2, 4, 4, 80), (2, 4, 4, 13)]):
        S = 10
        q = torch.randn(B, HQ, D).to(dtype)
        kb = torch.randn(B * S, HKV, D).to(dtype)
        vb = torch.randn(B * S, HKV, D).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.cumsum(torch.full((B,), S), 0)
        kv_indices = torch.arange(B * S)
        nsplit = torch.full((B,), 4, dtype=torch.int32)
        o, al, lse = run_decode(q, kb, vb, kv_indptr, kv_indices, nsplit, 8, 1.0 / D**0.5)
        cases[f"ref{ci}"] = dict(q=q, kb=kb, vb=vb, kv_indptr=kv_indptr, kv_indices=kv_indices,
                                 nsplit=nsplit, max_splits=8, sm_scale=1.0 / D**0.5, o=o,
                                 attn_logits=al, attn_lse=lse)
    # (b) grouped configs (:663-676), trimmed to interpreter-friendly sizes
    for ci, (B, S, HQ, HKV, D, DV) in enumerate([(2, 5, 16, 16, 64, 64), (2, 100, 16, 1, 64, 64),
                                                  (2, 37, 128, 2, 128, 128)]):
        q = torch.randn(B, HQ, D).to(dtype)
        kb = torch.randn(B * S, HKV, D).to(dtype)
        vb = torch.randn(B * S, HKV, DV).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.cumsum(torch.full((B,), S), 0)
        kv_indices = torch.arange(B * S)
        nsplit = torch.full((B,), 4, dtype=torch.int32)
        o, al, lse = run_decode(q, kb, vb, kv_indptr, kv_indices, nsplit, 8, 1.0 / D**0.5)
        cases[f"grp{ci}"] = dict(q=q, kb=kb, vb=vb, kv_indptr=kv_indptr, kv_indices=kv_indices,
                                 nsplit=nsplit, max_splits=8, sm_scale=1.0 / D**0.5, o=o,
                                 attn_logits=al, attn_lse=lse)
    # (c) Llama-shaped GQA, ragged lengths, shuffled slots, K3-chosen splits, logit cap variant
    rng = np.random.default_rng(5)
    for ci, (HQ, HKV, cap) in enumerate([(32, 8, 0.0), (16, 4, 0.0), (8, 2, 30.0), (4, 1, 0.0)]):
        D = 128
        lens = np.array([33, 1, 150, 64], dtype=np.int32)
        B = len(lens)
        total = int(lens.sum())
        pool = total + 17
        q = torch.randn(B, HQ, D).to(dtype)
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(lens))
        kv_indices = torch.from_numpy((rng.permutation(pool - 1)[:total] + 1).astype(np.int64))
        nsplit = torch.zeros(B, dtype=torch.int32)
        get_num_kv_splits_triton[(1,)](nsplit, torch.from_numpy(lens), B, 1, HQ, HKV, 8, 256,
                                       MAX_NUM_SEQ=256)
        o, al, lse = run_decode(q, kb, vb, kv_indptr, kv_indices, nsplit, 8, 1.0 / D**0.5,
                                logit_cap=cap)
        cases[f"llama{ci}"] = dict(q=q, kb=kb, vb=vb, kv_indptr=kv_indptr, kv_indices=kv_indices,
                                   nsplit=nsplit, max_splits=8, sm_scale=1.0 / D**0.5,
                                   logit_cap=cap, o=o, attn_logits=al, attn_lse=lse)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("decode.npz", **flat)


# ------------------------------------------------------------------ F6
def f6():
    torch.manual_seed(42)
    dtype = torch.float16
    rng = np.random.default_rng(9)
    cases = {}
    cfgs = [
        # name, Hq, Hkv, D, prefix lens, extend lens, causal, logit_cap
        ("gqa128", 12, 4, 128, [16, 33, 0], [5, 20, 9], True, 0.0),
        ("gqa64", 4, 2, 64, [0, 7], [40, 1], True, 0.0),
        ("mha96", 4, 4, 96, [12, 3], [9, 17], True, 0.0),
        ("noncausal", 8, 2, 128, [10, 0], [6, 11], False, 0.0),
        ("llama_cap", 32, 8, 128, [64, 5], [3, 33], True, 50.0),
    ]
    for name, HQ, HKV, D, pre, ext, causal, cap in cfgs:
        pre = np.array(pre, dtype=np.int32); ext = np.array(ext, dtype=np.int32)
        B = len(pre)
        T = int(ext.sum())
        total = int((pre + ext).sum())
        pool = total + 9
        slots = rng.permutation(pool - 1)[:total] + 1
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(T, HQ, D).to(dtype)
        k_ext = torch.empty(T, HKV, D, dtype=dtype)
        v_ext = torch.empty(T, HKV, D, dtype=dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(pre))
        qo_indptr = torch.zeros(B + 1, dtype=torch.int64)
        qo_indptr[1:] = torch.from_numpy(np.cumsum(ext))
        kv_indices = torch.empty(int(pre.sum()), dtype=torch.int64)
        so = 0
        for i in range(B):
            s = slots[so : so + pre[i] + ext[i]]; so += pre[i] + ext[i]
            kv_indices[kv_indptr[i] : kv_indptr[i + 1]] = torch.from_numpy(s[: pre[i]])
            es = torch.from_numpy(s[pre[i] :])
            k_ext[qo_indptr[i] : qo_indptr[i + 1]] = kb[es]
            v_ext[qo_indptr[i] : qo_indptr[i + 1]] = vb[es]
        o = torch.zeros(T, HQ, D, dtype=dtype)
        lse = torch.zeros(T, HQ, dtype=torch.float32)
        extend_attention_fwd(q, k_ext, v_ext, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None,
                             causal, None, int(ext.max()), 1.0, 1.0, sm_scale=1.0 / D**0.5,
                             logit_cap=cap, lse_extend=lse)
        cases[name] = dict(q=q, k_ext=k_ext, v_ext=v_ext, kb=kb, vb=vb, qo_indptr=qo_indptr,
                           kv_indptr=kv_indptr, kv_indices=kv_indices, causal=int(causal),
                           sm_scale=1.0 / D**0.5, logit_cap=cap, o=o, lse=lse)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("extend.npz", **flat)


# ------------------------------------------------------------------ F7
def _import_ref_radix():
    """The reference RadixCache needs only its tree logic here: stub the modules it imports for
    type names / metrics / events so `radix_cache.py` itself runs unmodified."""
    import sglang  # noqa: F401  (the real package first)

    for name in ["sglang.srt.mem_cache.memory_pool", "sglang.srt.observability.metrics_collector",
                 "sglang.srt.mem_cache.events", "sglang.srt.mem_cache.cpp_utils",
                 "sglang.srt.mem_cache.cpp_utils.native_hash", "sglang.kernels.ops.kvcache.mla_buffer"]:
        if name not in sys.modules:
            m = _ref_import._Stub(name)
            m.__path__ = []
            sys.modules[name] = m

    class _Mixin:
        def _record_all_cleared_event(self): pass
        def _record_store_event(self, n): pass
        def _record_remove_event(self, n): pass

    sys.modules["sglang.srt.mem_cache.events"].KVCacheEventMixin = _Mixin
    from sglang.srt.mem_cache import base_prefix_cache as bpc
    from sglang.srt.mem_cache import radix_cache as rc
    from sglang.srt.mem_cache.cache_init_params import CacheInitParams

    return rc, bpc, CacheInitParams


class _RecordingAllocator:
    device = "cpu"

    def __init__(self):
        self.freed = []

    def free_segment(self, idx, *, start_pos):
        self.freed.append(idx.tolist())


def gen_radix(page_size, policy, seed):
    from array import array

    rc, bpc, CacheInitParams = _import_ref_radix()
    rng = np.random.default_rng(seed)
    alloc = _RecordingAllocator()
    cache = rc.RadixCache(CacheInitParams(disable=False, req_to_token_pool=None,
                                          token_to_kv_pool_allocator=alloc, page_size=page_size,
                                          eviction_policy=policy))
    log, held = [], []  # held: (handle id, node) locked nodes
    next_slot = [1]
    hid = [0]

    def rand_key():
        # small vocabulary + page-wise construction => many shared prefixes and mid-node splits
        npages = int(rng.integers(1, 6))
        toks = []
        for _ in range(npages):
            toks.extend([int(rng.integers(0, 3))] * page_size if rng.random() < 0.7 else
                        [int(x) for x in rng.integers(0, 3, size=page_size)])
        toks.extend(int(x) for x in rng.integers(0, 3, size=int(rng.integers(0, page_size))))
        return toks

    def sizes():
        return [cache.evictable_size(), cache.protected_size(), cache.total_size()]

    for _ in range(120):
        op = rng.choice(["insert", "match", "lock", "unlock", "evict"], p=[0.35, 0.3, 0.12, 0.1, 0.13])
        extra = None if rng.random() < 0.85 else "lora1"
        if op == "insert":
            toks = rand_key()
            vals = list(range(next_slot[0], next_slot[0] + len(toks)))
            next_slot[0] += len(toks)
            prio = int(rng.integers(0, 3))
            chunked = bool(rng.random() < 0.2)
            r = cache.insert(bpc.InsertParams(key=rc.RadixKey(array("q", toks), extra),
                                              value=torch.tensor(vals, dtype=torch.int64),
                                              priority=prio, chunked=chunked))
            log.append(dict(op="insert", tokens=toks, values=vals, extra=extra, priority=prio,
                            chunked=chunked, prefix_len=r.prefix_len, sizes=sizes()))
        elif op == "match":
            toks = rand_key()
            m = cache.match_prefix(bpc.MatchPrefixParams(key=rc.RadixKey(array("q", toks), extra)))
            log.append(dict(op="match", tokens=toks, extra=extra, indices=m.device_indices.tolist(),
                            last_is_root=m.last_device_node is cache.root_node,
                            last_key_len=len(m.last_device_node.key), sizes=sizes()))
        elif op == "lock":
            toks = rand_key()
            m = cache.match_prefix(bpc.MatchPrefixParams(key=rc.RadixKey(array("q", toks), extra)))
            d = cache.inc_lock_ref(m.last_device_node).delta
            held.append((hid[0], m.last_device_node))
            log.append(dict(op="lock", tokens=toks, extra=extra, indices=m.device_indices.tolist(),
                            handle=hid[0], delta=d, sizes=sizes()))
            hid[0] += 1
        elif op == "unlock" and held:
            i = int(rng.integers(0, len(held)))
            h, node = held.pop(i)
            d = cache.dec_lock_ref(node).delta
            log.append(dict(op="unlock", handle=h, delta=d, sizes=sizes()))
        elif op == "evict":
            n = int(rng.integers(1, 4 * page_size + 2))
            alloc.freed = []
            r = cache.evict(bpc.EvictParams(num_tokens=n))
            log.append(dict(op="evict", num_tokens=n, evicted=r.num_tokens_evicted,
                            segments=alloc.freed, sizes=sizes()))
    return dict(page_size=page_size, policy=policy, log=log)


def f7():
    cases = []
    for ps, pol, seed in [(1, "lru", 1), (4, "lru", 2), (16, "lru", 3), (1, "lfu", 4), (4, "fifo", 5),
                          (1, "priority", 6), (4, "mru", 7), (1, "filo", 8), (4, "slru", 9)]:
        cases.append(gen_radix(ps, pol, seed))
    with open(os.path.join(HERE, "radix_sequences.json"), "w") as f:
        json.dump(cases, f)
    print("wrote radix_sequences.json", [len(c["log"]) for c in cases])


# ------------------------------------------------------------------ F8
def f8():
    """bf16 goldens from the reference's compiled native CPU kernels (oracle/_ref, built by
    oracle/build_ref.py from aot/csrc/cpu/{decode,extend}.cpp): decode_attention_cpu walks
    req_to_token and writes the new K/V at loc in-kernel (decode.cpp:940)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import build_ref

    m = build_ref.load() or __import__("importlib").import_module("oracle.build_ref").load()
    if m is None:
        build_ref.build()
        m = build_ref.load()
    torch.manual_seed(42)
    dt = torch.bfloat16
    out = {}
    for ci, (HQ, HKV, D, lens) in enumerate([(32, 8, 128, [100, 33, 257, 1]), (12, 12, 64, [7, 64, 65]),
                                             (16, 2, 128, [300, 31])]):
        lens = torch.tensor(lens, dtype=torch.int64)
        B = len(lens)
        pool = int(lens.sum()) + 11
        kb = torch.randn(pool, HKV, D).to(dt); vb = torch.randn(pool, HKV, D).to(dt)
        q = torch.randn(B, HQ, D).to(dt)
        r2t = torch.zeros(B + 1, int(lens.max()) + 3, dtype=torch.int32)
        perm = torch.randperm(pool - 1) + 1
        o = 0
        for i, n in enumerate(lens.tolist()):
            r2t[i + 1, :n] = perm[o:o + n].int(); o += n
        rpi = torch.arange(1, B + 1, dtype=torch.int64)
        newk = torch.randn(B, HKV, D).to(dt); newv = torch.randn(B, HKV, D).to(dt)
        loc = torch.tensor([r2t[i + 1, lens[i] - 1].item() for i in range(B)], dtype=torch.int64)
        res = torch.zeros(B, HQ, D, dtype=dt)
        attn_logits = torch.zeros(B, HQ, 8, D + 1)
        kb_in, vb_in = kb.clone(), vb.clone()
        m.decode_attention_cpu(q, kb, vb, res, newk, newv, loc, attn_logits, r2t, rpi, lens,
                               D ** -0.5, 0.0, False, 0, None, None)
        t = f"dec{ci}."
        out.update({t + "q": bits(q), t + "kb_in": bits(kb_in), t + "vb_in": bits(vb_in),
                    t + "kb_out": bits(kb), t + "vb_out": bits(vb), t + "new_k": bits(newk),
                    t + "new_v": bits(newv), t + "loc": loc.numpy(), t + "req_to_token": r2t.numpy(),
                    t + "req_pool_indices": rpi.numpy(), t + "seq_lens": lens.numpy(),
                    t + "sm_scale": np.asarray(D ** -0.5), t + "o": bits(res)})
    for ci, (HQ, HKV, D, pre, ext) in enumerate([(12, 4, 128, [16, 33, 0], [5, 20, 9]),
                                                 (32, 8, 128, [64, 5], [3, 40])]):
        pre = torch.tensor(pre, dtype=torch.int64); ext = torch.tensor(ext, dtype=torch.int64)
        seq = pre + ext
        B, T = len(pre), int(ext.sum())
        pool = int(seq.sum()) + 7
        kb = torch.randn(pool, HKV, D).to(dt); vb = torch.randn(pool, HKV, D).to(dt)
        q = torch.randn(T, HQ, D).to(dt)
        r2t = torch.zeros(B + 1, int(seq.max()) + 2, dtype=torch.int32)
        perm = torch.randperm(pool - 1) + 1
        o = 0
        for i, n in enumerate(seq.tolist()):
            r2t[i + 1, :n] = perm[o:o + n].int(); o += n
        rpi = torch.arange(1, B + 1, dtype=torch.int64)
        start = torch.zeros(B, dtype=torch.int64); start[1:] = torch.cumsum(ext[:-1], 0)
        k_ext = torch.cat([kb[r2t[i + 1, pre[i]:seq[i]].long()] for i in range(B)])
        v_ext = torch.cat([vb[r2t[i + 1, pre[i]:seq[i]].long()] for i in range(B)])
        res = torch.zeros(T, HQ, D, dtype=dt)
        m.extend_attention_cpu(q, k_ext, v_ext, res, kb, vb, r2t, rpi, seq, ext.int(), start.int(), int(ext.max()),
                               D ** -0.5, 0.0, False, 0, None, None, None)
        t = f"ext{ci}."
        out.update({t + "q": bits(q), t + "k_ext": bits(k_ext), t + "v_ext": bits(v_ext), t + "kb": bits(kb),
                    t + "vb": bits(vb), t + "req_to_token": r2t.numpy(), t + "req_pool_indices": rpi.numpy(),
                    t + "seq_lens": seq.numpy(), t + "extend_seq_lens": ext.numpy(),
                    t + "extend_prefix_lens": pre.numpy(), t + "sm_scale": np.asarray(D ** -0.5),
                    t + "o": bits(res)})
    save("cpu_native.npz", **out)


def f9():
    """F9 extend with the speculative-decoding tree mask (custom_mask / mask_indptr, with and without
    the prefix part masked), sliding window + window_kv_offsets, and Grok's xai temperature ->
    extend_mask.npz.  Reference kernel under TRITON_INTERPRET=1 (fp16)."""
    torch.manual_seed(7)
    dtype = torch.float16
    rng = np.random.default_rng(19)
    cfgs = [
        # name, Hq, Hkv, D, prefix, extend, skip_prefix_mask, window, woff, xai_len
        ("tree", 8, 2, 128, [40, 7], [6, 6], True, -1, None, -1),
        ("tree_prefix", 4, 4, 64, [33, 5], [5, 9], False, -1, None, -1),
        ("tree_swa", 8, 2, 128, [20, 12], [4, 4], True, 16, [3, 0], -1),
        ("xai", 8, 2, 128, [50, 3], [9, 17], None, -1, None, 16),
        ("swa", 4, 2, 128, [70, 10], [12, 40], None, 24, None, -1),
    ]
    flat = {}
    for name, HQ, HKV, D, pre, ext, skipm, window, woff, xai in cfgs:
        pre = np.array(pre, dtype=np.int32); ext = np.array(ext, dtype=np.int32)
        B = len(pre)
        T = int(ext.sum())
        pool = int(pre.sum()) + 9
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(T, HQ, D).to(dtype)
        k_ext = torch.randn(T, HKV, D).to(dtype)
        v_ext = torch.randn(T, HKV, D).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(pre))
        qo_indptr = torch.zeros(B + 1, dtype=torch.int64)
        qo_indptr[1:] = torch.from_numpy(np.cumsum(ext))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(pre.sum())] + 1).to(torch.int64)
        custom_mask = mask_indptr = wko = None
        if skipm is not None:
            wo = np.array(woff if woff is not None else [0] * B, dtype=np.int32)
            rows = []
            for i in range(B):
                L = int(pre[i] + ext[i] + wo[i])
                m = rng.random((int(ext[i]), L)) < 0.5
                # a draft token always sees itself; tree masks never look ahead
                for r in range(int(ext[i])):
                    m[r, wo[i] + pre[i] + r] = True
                    m[r, wo[i] + pre[i] + r + 1:] = False
                    if skipm:
                        m[r, : wo[i] + pre[i]] = True
                    else:
                        m[r, wo[i]] = True  # keep every row non-empty in the prefix part too
                rows.append(m.reshape(-1))
            custom_mask = torch.from_numpy(np.concatenate(rows))
            mask_indptr = torch.zeros(B + 1, dtype=torch.int64)
            mask_indptr[1:] = torch.from_numpy(np.cumsum([r.size for r in rows]))
            if woff is not None:
                wko = torch.from_numpy(wo)
        o = torch.zeros(T, HQ, D, dtype=dtype)
        extend_attention_fwd(q, k_ext, v_ext, o, kb, vb, qo_indptr, kv_indptr, kv_indices, custom_mask,
                             True, mask_indptr, int(ext.max()), 1.0, 1.0, sm_scale=1.0 / D**0.5,
                             skip_prefix_custom_mask=bool(skipm) if skipm is not None else True,
                             sliding_window_size=window, window_kv_offsets=wko, xai_temperature_len=xai)
        c = dict(q=q, k_ext=k_ext, v_ext=v_ext, kb=kb, vb=vb, qo_indptr=qo_indptr, kv_indptr=kv_indptr,
                 kv_indices=kv_indices, sm_scale=1.0 / D**0.5, window=window, xai=xai,
                 skip_prefix_mask=-1 if skipm is None else int(skipm), o=o)
        if custom_mask is not None:
            c["custom_mask"] = custom_mask.to(torch.uint8)
            c["mask_indptr"] = mask_indptr
        if wko is not None:
            c["window_kv_offsets"] = wko
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("extend_mask.npz", **flat)


def f10():
    """F10 decode with the xai temperature (grouped and MHA stage-1 kernels) -> decode_xai.npz."""
    torch.manual_seed(3)
    dtype = torch.float16
    rng = np.random.default_rng(5)
    flat = {}
    for name, HQ, HKV, D, lens, xai in [("gqa", 8, 2, 128, [5, 40, 200, 17], 16), ("mha", 4, 4, 64, [100, 3], 8)]:
        lens = np.array(lens, dtype=np.int32)
        B = len(lens)
        pool = int(lens.sum()) + 5
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(B, HQ, D).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(lens))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(lens.sum())] + 1).to(torch.int64)
        S = 4
        nsplit = torch.full((B,), 2, dtype=torch.int32)
        o = torch.zeros(B, HQ, D, dtype=dtype)
        al = torch.zeros(B, HQ, S, D, dtype=torch.float32)
        lse = torch.zeros(B, HQ, S, dtype=torch.float32)
        decode_attention_fwd(q, kb, vb, o, kv_indptr, kv_indices, al, lse, nsplit, S, 1.0 / D**0.5, 1.0, 1.0,
                             xai_temperature_len=xai)
        c = dict(q=q, kb=kb, vb=vb, kv_indptr=kv_indptr, kv_indices=kv_indices, nsplit=nsplit, max_splits=S,
                 sm_scale=1.0 / D**0.5, xai=xai, o=o)
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("decode_xai.npz", **flat)


def f11():
    """F11 rotary embedding: the reference's torch-native apply_rotary_emb
    (srt/layers/rotary_embedding/utils.py:36-62) on fp32 copies of fp16 q/k with the cos_sin_cache of
    RotaryEmbedding._compute_cos_sin_cache (base.py:171-181) -> rope.npz."""
    from sglang.srt.layers.rotary_embedding.utils import apply_rotary_emb

    torch.manual_seed(13)
    flat = {}
    for name, D, rot, neox, base in [("neox128", 128, 128, True, 10000.0), ("gptj64", 64, 64, False, 10000.0),
                                     ("partial", 128, 64, True, 500000.0)]:
        n, HQ, HKV, max_pos = 9, 4, 2, 4096
        inv_freq = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))
        freqs = torch.einsum("i,j -> ij", torch.arange(max_pos, dtype=torch.float), inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1)
        pos = torch.tensor([0, 1, 2, 77, 1000, 4095, 5, 5, 300])
        q = torch.randn(n, HQ, D).half()
        k = torch.randn(n, HKV, D).half()
        cos, sin = cache.index_select(0, pos).chunk(2, dim=-1)
        outs = []
        for x in (q, k):
            xr = apply_rotary_emb(x[..., :rot].float(), cos, sin, neox)
            outs.append(torch.cat((xr, x[..., rot:].float()), dim=-1))
        # the fixture keeps only the cache rows that are used (positions remapped to them): the values
        # every consumer reads are unchanged, the file is 100x smaller
        uniq, remap = torch.unique(pos, return_inverse=True)
        c = dict(q=q, k=k, positions=remap, true_positions=pos, cos_sin_cache=cache.index_select(0, uniq),
                 rotary_dim=rot, is_neox=int(neox), q_out=outs[0], k_out=outs[1])
        for kk, v in c.items():
            flat[f"{name}.{kk}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("rope.npz", **flat)


def f12():
    """F12 the unified one-stage extend of deterministic inference (extend_attention_fwd_unified): plain
    causal, sliding window, tree mask, xai temperature -> extend_unified.npz (fp16, interpreter)."""
    from sglang.kernels.ops.attention.extend_attention import extend_attention_fwd_unified

    torch.manual_seed(21)
    dtype = torch.float16
    rng = np.random.default_rng(31)
    flat = {}
    for name, HQ, HKV, D, pre, ext, window, masked, xai in [
            ("causal", 8, 2, 128, [40, 0, 7], [6, 20, 9], -1, False, -1),
            ("mha64", 4, 4, 64, [33, 5], [5, 9], -1, False, -1),
            ("swa", 8, 2, 128, [70, 10], [12, 40], 24, False, -1),
            ("tree", 8, 2, 128, [50, 12], [6, 6], -1, True, -1),
            ("xai", 8, 2, 128, [50, 3], [9, 17], -1, False, 16)]:
        pre = np.array(pre, dtype=np.int32); ext = np.array(ext, dtype=np.int32)
        B = len(pre)
        T = int(ext.sum())
        tot = pre + ext
        pool = int(tot.sum()) + 9
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(T, HQ, D).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(tot))
        qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
        qo_indptr[1:] = torch.from_numpy(np.cumsum(ext))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(tot.sum())] + 1).to(torch.int64)
        cm = mi = None
        if masked:
            rows = []
            for i in range(B):
                m = rng.random((int(ext[i]), int(tot[i]))) < 0.5
                for r in range(int(ext[i])):
                    m[r, pre[i] + r] = True
                    m[r, pre[i] + r + 1:] = False
                rows.append(m.reshape(-1))
            cm = torch.from_numpy(np.concatenate(rows))
            mi = torch.zeros(B + 1, dtype=torch.int64)
            mi[1:] = torch.from_numpy(np.cumsum([r.size for r in rows]))
        o = torch.zeros(T, HQ, D, dtype=dtype)
        extend_attention_fwd_unified(q, o, kb, vb, 1.0, 1.0, qo_indptr, kv_indptr, kv_indices, torch.from_numpy(pre),
                                     int(ext.max()), custom_mask=cm, mask_indptr=mi, sm_scale=1.0 / D**0.5,
                                     is_causal=True, sliding_window_size=window, xai_temperature_len=xai)
        c = dict(q=q, kb=kb, vb=vb, qo_indptr=qo_indptr, kv_indptr=kv_indptr, kv_indices=kv_indices,
                 prefix_lens=torch.from_numpy(pre), sm_scale=1.0 / D**0.5, window=window, xai=xai, o=o)
        if cm is not None:
            c["custom_mask"] = cm.to(torch.uint8)
            c["mask_indptr"] = mi
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("extend_unified.npz", **flat)


# ------------------------------------------------------------------ F13
def gen_flow(page_size, seed):
    """A small scheduler loop on the REFERENCE's RadixCache + its CPU allocator + a req_to_token table: admit a
    request (match_prefix, lock, allocate the new tokens' slots, write its row, cache_unfinished_req), decode steps
    (alloc_decode / alloc + row write), re-caching between batches, finishing (cache_finished_req), explicit evictions.
    After every op: the request's row, its cache_protected_len / prefix_indices, both allocator lists and the tree
    sizes.  This pins the ORCHESTRATION (SURVEY a6 + the request hooks of a16): tests/test_gpu_radix_flow.py replays it
    through sglang_amd.mem_cache.{allocation, radix_cache, allocator} on the GPU."""
    from array import array

    from sglang.srt.mem_cache.allocator.paged import PagedTokenToKVPoolAllocator
    from sglang.srt.mem_cache.allocator.token import TokenToKVPoolAllocator

    rc, bpc, CacheInitParams = _import_ref_radix()
    rng = np.random.default_rng(seed)
    size, rows, ctx = 96 * max(page_size, 4), 12, 160
    alloc = (TokenToKVPoolAllocator(size, torch.bfloat16, "cpu", None, False) if page_size == 1 else
             PagedTokenToKVPoolAllocator(size, page_size, torch.bfloat16, "cpu", None, False))

    class Pool:
        def __init__(self):
            self.req_to_token = torch.zeros((rows, ctx), dtype=torch.int32)

        def write(self, indices, values):
            self.req_to_token[indices] = values.to(torch.int32)

    pool = Pool()
    cache = rc.RadixCache(CacheInitParams(disable=False, req_to_token_pool=pool, token_to_kv_pool_allocator=alloc,
                                          page_size=page_size, eviction_policy="lru"))

    class Req:
        def __init__(self, rid, toks, row):
            self.rid, self.origin_input_ids, self.output_ids = rid, list(toks), []
            self.req_pool_idx, self.extra_key, self.priority = row, None, 0
            self.prefix_indices, self.last_node, self.cache_protected_len = None, None, 0

        def get_fill_ids(self):
            return self.origin_input_ids + self.output_ids

    live, free_rows, log, next_rid = {}, list(range(1, rows)), [], [0]
    stems = [[int(x) for x in rng.integers(0, 50, size=3 * page_size + 5)] for _ in range(3)]

    def snap(req=None):
        d = dict(free=alloc.free_pages.tolist(), release=alloc.release_pages.tolist(),
                 sizes=[cache.evictable_size(), cache.protected_size(), cache.total_size()])
        if req is not None:
            n = len(req.get_fill_ids())
            d.update(row=pool.req_to_token[req.req_pool_idx, :n].tolist(), protected=req.cache_protected_len,
                     prefix_indices=None if req.prefix_indices is None else req.prefix_indices.tolist())
        return d

    for _ in range(70):
        op = rng.choice(["new", "decode", "recache", "finish", "evict"], p=[0.3, 0.35, 0.1, 0.17, 0.08])
        if op == "new" and free_rows:
            stem = stems[int(rng.integers(0, len(stems)))]
            toks = stem[: int(rng.integers(1, len(stem) + 1))] + [int(x) for x in rng.integers(50, 60, size=int(rng.integers(1, 2 * page_size + 3)))]
            req = Req(next_rid[0], toks, free_rows.pop(0))
            next_rid[0] += 1
            m = cache.match_prefix(bpc.MatchPrefixParams(key=rc.RadixKey(list(toks), None)))
            req.prefix_indices, req.last_node = m.device_indices, m.last_device_node
            cache.inc_lock_ref(req.last_node)
            pre = len(req.prefix_indices)
            req.cache_protected_len = pre
            pool.req_to_token[req.req_pool_idx, :pre] = req.prefix_indices.to(torch.int32)
            ext = len(toks) - pre
            if page_size == 1:
                out = alloc.alloc(ext)
            else:
                pl, sl = torch.tensor([pre]), torch.tensor([len(toks)])
                last = torch.tensor([int(req.prefix_indices[-1]) if pre else -1])
                out = alloc.alloc_extend(pl, pl, sl, sl, last, ext)
            assert out is not None
            pool.req_to_token[req.req_pool_idx, pre: len(toks)] = out.to(torch.int32)
            cache.cache_unfinished_req(req)
            live[req.rid] = req
            log.append(dict(op="new", rid=req.rid, row=req.req_pool_idx, tokens=toks, matched=pre, out=out.tolist(),
                            after=snap(req)))
        elif op == "decode" and live:
            req = live[int(rng.choice(list(live)))]
            n = len(req.get_fill_ids())
            if n + 1 >= ctx:
                continue
            if page_size == 1:
                loc = alloc.alloc(1)
            else:
                sl = torch.tensor([n + 1])
                loc = alloc.alloc_decode(sl, sl, torch.tensor([int(pool.req_to_token[req.req_pool_idx, n - 1])]))
            assert loc is not None
            pool.req_to_token[req.req_pool_idx, n] = int(loc[0])
            req.output_ids.append(int(rng.integers(60, 70)))
            log.append(dict(op="decode", rid=req.rid, loc=loc.tolist(), token=req.output_ids[-1], after=snap(req)))
        elif op == "recache" and live:
            req = live[int(rng.choice(list(live)))]
            cache.cache_unfinished_req(req)
            log.append(dict(op="recache", rid=req.rid, after=snap(req)))
        elif op == "finish" and live:
            req = live.pop(int(rng.choice(list(live))))
            insert = bool(rng.random() < 0.8)
            cache.cache_finished_req(req, is_insert=insert, kv_len_to_handle=len(req.get_fill_ids()))
            free_rows.append(req.req_pool_idx)
            log.append(dict(op="finish", rid=req.rid, insert=insert, after=snap()))
        elif op == "evict":
            n = int(rng.integers(1, 6 * page_size))
            r = cache.evict(bpc.EvictParams(num_tokens=n))
            log.append(dict(op="evict", num_tokens=n, evicted=r.num_tokens_evicted, after=snap()))
    return dict(page_size=page_size, size=size, rows=rows, ctx=ctx, log=log)


def f13():
    cases = [gen_flow(ps, seed) for ps, seed in [(1, 1), (4, 2), (16, 3), (16, 4)]]
    with open(os.path.join(HERE, "scheduler_flow.json"), "w") as f:
        json.dump(cases, f)
    print("wrote scheduler_flow.json", [len(c["log"]) for c in cases])


# ------------------------------------------------------------------ F14
def _ref_torch_native():
    """The two SDPA helpers of the reference's TorchNativeAttnBackend (torch_native_backend.py:36-277) as a class that
    can be instantiated here: the module itself does not import (forward_batch_info -> configs -> torchvision), but the
    three methods are torch-only, so they are cut out of the reference FILE with ast at generation time and executed --
    nothing of their text is stored, only the vectors they produce."""
    import ast
    import typing

    path = os.path.join(_ref_import.REF_PY, "sglang/srt/layers/attention/torch_native_backend.py")
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TorchNativeAttnBackend")
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and
            n.name in ("_make_sliding_window_mask", "_run_sdpa_forward_extend", "_run_sdpa_forward_decode")]
    assert len(keep) == 3, [n.name for n in keep]
    mod = ast.Module(body=[ast.ClassDef(name="RefTorchNative", bases=[], keywords=[], body=keep, decorator_list=[])],
                     type_ignores=[])
    ns = {"torch": torch, "Optional": typing.Optional,
          "scaled_dot_product_attention": torch.nn.functional.scaled_dot_product_attention}
    exec(compile(ast.fix_missing_locations(mod), path, "exec"), ns)
    return ns["RefTorchNative"]()


def f14():
    """a14: outputs of the reference's torch-native SDPA helpers on fp32 CPU tensors (so that the comparison with the
    fp64 oracle is tight): ragged extend over cached prefixes with GQA, decode, and both under a sliding window."""
    ref = _ref_torch_native()
    g = torch.Generator().manual_seed(14)
    flat = {}
    for name, hq, hkv, d, window in (("gqa", 8, 2, 64, None), ("mha", 4, 4, 32, None), ("mqa_window", 4, 1, 64, 5),
                                     ("gqa_window", 8, 2, 64, 17)):
        prefix = [0, 7, 33, 16]
        ext = [5, 1, 20, 16]
        seq = [p + e for p, e in zip(prefix, ext)]
        pool = sum(seq) + 9
        perm = torch.randperm(pool - 1, generator=g) + 1
        r2t = torch.zeros(6, 64, dtype=torch.int64)
        rows, o = [3, 1, 4, 2], 0
        for r, n in zip(rows, seq):
            r2t[r, :n] = perm[o: o + n]
            o += n
        kc = torch.randn(pool, hkv, d, generator=g)
        vc = torch.randn(pool, hkv, d, generator=g)
        T = sum(ext)
        q = torch.randn(T, hq, d, generator=g)
        rpi = torch.tensor(rows)
        out = ref._run_sdpa_forward_extend(q, torch.zeros(T, hq, d), kc, vc, r2t, rpi, torch.tensor(seq),
                                           torch.tensor(prefix), torch.tensor(ext), scaling=d ** -0.5,
                                           enable_gqa=hq != hkv, causal=True, sliding_window_size=window)
        qd = torch.randn(len(seq), hq, d, generator=g)
        outd = ref._run_sdpa_forward_decode(qd, torch.zeros(len(seq), hq, d), kc, vc, r2t, rpi, torch.tensor(seq),
                                            scaling=d ** -0.5, enable_gqa=hq != hkv, causal=False,
                                            sliding_window_size=window)
        c = dict(q=q, qd=qd, kc=kc, vc=vc, r2t=r2t, rpi=rpi, seq=torch.tensor(seq), prefix=torch.tensor(prefix),
                 ext=torch.tensor(ext), window=-1 if window is None else window, o_extend=out, o_decode=outd)
        for k, v in c.items():
            flat[f"{name}.{k}"] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    save("torch_native.npz", **flat)


# ------------------------------------------------------------------ F15
def _ref_funcs(rel_path, names, ns):
    """Functions cut out of a reference FILE with ast at generation time and executed (the module around them does not
    import here); only the vectors they produce are stored."""
    import ast

    path = os.path.join(_ref_import.REF_PY, rel_path)
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), [n.name for n in keep]
    exec(compile(ast.fix_missing_locations(ast.Module(body=keep, type_ignores=[])), path, "exec"), ns)
    return [ns[n] for n in names]


def f15():
    """F15 decode context parallel: get_dcp_lens (srt/layers/dcp/layout.py), the per-rank kv-index kernel
    (create_triton_kv_indices_for_dcp_triton, under the Triton interpreter) and cp_lse_ag_out_rs_mha
    (srt/layers/dcp/comm.py) run rank by rank against a recording stand-in for the process group -> dcp.npz."""
    import typing

    from sglang.kernels.ops.attention.dcp_kernels import create_triton_kv_indices_for_dcp_triton

    (get_dcp_lens,) = _ref_funcs("sglang/srt/layers/dcp/layout.py", ["get_dcp_lens"], {"torch": torch})
    rng = np.random.default_rng(15)
    out = {}
    ci = 0
    for dcp in (2, 3, 8):
        batch, max_batch, ctx = 9, 16, 400
        r2t = torch.from_numpy(rng.integers(0, 1 << 20, size=(max_batch, ctx)).astype(np.int32))
        rpi = torch.from_numpy(rng.choice(max_batch, size=batch, replace=False).astype(np.int32))
        lens = torch.from_numpy(np.concatenate([[0, 1, dcp - 1, dcp, dcp + 1],
                                                rng.integers(0, ctx // 2, size=batch - 5)]).astype(np.int32))
        start = torch.from_numpy(rng.integers(0, ctx // 2, size=batch).astype(np.int32))
        for use_start in (False, True):
            for rank in range(dcp):
                dl = get_dcp_lens(lens, dcp, rank, start if use_start else None).to(torch.int32)
                kv_indptr = torch.zeros((batch + 1,), dtype=torch.int32)
                kv_indptr[1:] = torch.cumsum(dl, 0)
                kv_indices = torch.full((int(kv_indptr[-1]),), -1, dtype=torch.int64)
                create_triton_kv_indices_for_dcp_triton[(batch,)](
                    r2t, rpi, dl, kv_indptr, start if use_start else None, kv_indices, r2t.stride(0), dcp, rank)
                tag = f"idx{ci}"
                ci += 1
                out.update({f"{tag}.req_to_token": r2t.numpy(), f"{tag}.req_pool_indices": rpi.numpy(),
                            f"{tag}.lens": lens.numpy(), f"{tag}.start": start.numpy(),
                            f"{tag}.use_start": np.int32(use_start), f"{tag}.dcp": np.int32(dcp),
                            f"{tag}.rank": np.int32(rank), f"{tag}.dcp_lens": dl.numpy(),
                            f"{tag}.kv_indptr": kv_indptr.numpy(), f"{tag}.kv_indices": kv_indices.numpy()})
    out["idx.count"] = np.int32(ci)

    # cp_lse_ag_out_rs_mha: every rank's call sees the same all-gathered LSEs; the stand-in's all_reduce records the
    # rank's scaled output and returns the sum over ALL ranks (computed from the recorded ones on a second pass)
    class Group:
        def __init__(self, world, rank, lses, summed):
            self.world_size, self.rank_in_group, self._lses, self._summed, self.scaled = world, rank, lses, summed, None

        def all_gather(self, x, dim=0):
            assert dim == 0
            return torch.cat(list(self._lses), dim=0)

        def all_reduce(self, x):
            self.scaled = x.clone()
            return self._summed if self._summed is not None else x

    ns = {"torch": torch, "GroupCoordinator": object, "Optional": typing.Optional}
    _ag_lse, merge = _ref_funcs("sglang/srt/layers/dcp/comm.py", ["_ag_lse", "cp_lse_ag_out_rs_mha"], ns)
    g = torch.Generator().manual_seed(15)
    for mi, (world, T, H, D) in enumerate([(2, 5, 4, 16), (4, 3, 8, 32)]):
        outs = [torch.randn(T, H, D, generator=g) for _ in range(world)]
        lses = [torch.randn(T, H, generator=g) * 3 for _ in range(world)]
        lses[0][0, :] = float("-inf")            # a rank without tokens for one request
        outs[0][0, :] = float("nan")             # ... whose output row is undefined
        for r in range(world):
            lses[r][1, 0] = float("-inf")        # a row empty on every rank
        scaled = []
        for r in range(world):
            grp = Group(world, r, lses, None)
            merge(outs[r].clone(), lses[r].clone(), grp)
            scaled.append(grp.scaled)
        summed = torch.stack(scaled).sum(0)
        finals, glses = [], []
        for r in range(world):
            o, l = merge(outs[r].clone(), lses[r].clone(), Group(world, r, lses, summed.clone()), return_lse=True)
            finals.append(o)
            glses.append(l)
        out.update({f"merge{mi}.outs": torch.stack(outs).numpy(), f"merge{mi}.lses": torch.stack(lses).numpy(),
                    f"merge{mi}.scaled": torch.stack(scaled).numpy(), f"merge{mi}.final": torch.stack(finals).numpy(),
                    f"merge{mi}.global_lse": torch.stack(glses).numpy()})
    out["merge.count"] = np.int32(2)
    save("dcp.npz", **out)


def f16():
    """F16 the ROCm MLA decode with fused RoPE -> mla_rope.npz.  The reference's stage-1 kernel
    _fwd_grouped_kernel_stage1_rope (kernels/ops/attention/rocm_mla_decode_rope.py:45-315) under TRITON_INTERPRET=1
    (fp16), both rotation styles; its public wrapper decode_attention_fwd_grouped_rope calls the stage-2 reduce with a
    stale signature in this tree (:439 vs decode_attention.py's _decode_softmax_reducev_fwd) and cannot run, so the
    stage-1 partials [bs, Hq, splits, c + 1] (normalised acc | lse, :294-315) are merged here by their LSEs -- what
    stage 2 computes (decode_attention.py:731-805).  The module asks Triton's active driver for its backend at import
    time; a CPU box has none, so a stub driver answers (the interpreter never uses it)."""
    import triton

    class _Target:
        backend = "cuda"

    class _Drv:
        def get_current_target(self):
            return _Target()

        def get_current_device(self):
            return 0

    triton.runtime.driver.set_active(_Drv())
    from sglang.kernels.ops.attention.rocm_mla_decode_rope import _decode_grouped_att_m_fwd_rope

    torch.manual_seed(16)
    rng = np.random.default_rng(16)
    dtype = torch.float16
    C, R, maxpos = 512, 64, 256
    inv = 1.0 / (10000 ** (torch.arange(0, R, 2).float() / R))
    fr = torch.outer(torch.arange(maxpos).float(), inv)
    cache = torch.cat((fr.cos(), fr.sin()), dim=-1)  # fp32 [maxpos, 64]
    flat = {}
    for name, H, lens, S, neox in [("neox", 16, [5, 70, 33, 1], 2, True), ("gptj", 16, [40, 129, 2], 4, False),
                                   ("neox_h20", 20, [64, 9], 1, True)]:
        lens = np.array(lens, dtype=np.int32)
        B = len(lens)
        pool = int(lens.sum()) + 3
        kb = (torch.randn(pool, 1, C + R) * 0.5).to(dtype)
        q = (torch.randn(B, H, C + R) * 0.5).to(dtype)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(lens))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(lens.sum())] + 1).to(torch.int64)
        positions = torch.from_numpy(lens.astype(np.int64) - 1 + rng.integers(0, 50, size=B))
        att = torch.zeros(B, H, S, C + 1, dtype=torch.float32)
        kpe = torch.zeros(B, 1, R, dtype=dtype)
        sm = 1.0 / (192 ** 0.5)
        _decode_grouped_att_m_fwd_rope(q, kb, kb[..., :C], att, kpe, C, cache, positions, R, kv_indptr, kv_indices, S,
                                       sm, 0.0, True, neox)
        # stage 2: LSE merge of the splits that hold tokens (split s covers [s * ceil(len / S), ...): :112-114)
        o = torch.zeros(B, H, C, dtype=torch.float64)
        for b in range(B):
            per = -(-int(lens[b]) // S)
            live = [s_ for s_ in range(S) if per * s_ < int(lens[b])]
            l = att[b, :, live, C].double()                     # [H, live]
            w = torch.exp(l - l.max(dim=1, keepdim=True).values)
            o[b] = ((att[b, :, live, :C].double() * w[..., None]).sum(1) / w.sum(1, keepdim=True))
        c = dict(q=q, kb=kb, kv_indptr=kv_indptr, kv_indices=kv_indices, positions=positions, cos_sin=cache,
                 sm_scale=sm, neox=int(neox), splits=S, o=o.to(dtype), k_pe_out=kpe)
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("mla_rope.npz", **flat)


def f17():
    """F17 fused QK-norm + RoPE -> qknorm_rope.npz.  The reference's fused kernel (kernels/jit/csrc/elementwise/
    fused_qknorm_rope.cuh) is CUDA-only; its own test (kernels/aot/tests/test_fused_qk_norm_rope.py:31-128) checks it
    against RMSNorm + RotaryEmbedding, and that pair is what runs here, in fp32 (the fused kernel rounds once, at the
    end): RMSNorm.forward_native (srt/layers/layernorm.py:644-692) -- the module itself does not import in this container
    (transformers' image-processing chain), so the METHOD is taken from the file with `ast` at run time and executed on a
    stand-in object carrying the attributes it reads; nothing of its text is stored -- then apply_rotary_emb
    (srt/layers/rotary_embedding/utils.py:36-62) with the cos / sin of RotaryEmbedding._compute_cos_sin_cache
    (base.py:171-181) at the positions.  Cases: neox / interleaved, partial rotary, head dims 64 / 128 / 256, bf16 inputs
    (+ one fp16), the reference test's weights (randn * 5) and positions (+100)."""
    import ast
    import types

    from sglang.srt.layers.rotary_embedding.utils import apply_rotary_emb

    path = "/root/reference/python/sglang/srt/layers/layernorm.py"
    tree = ast.parse(open(path).read())
    fn = next(b for node in tree.body if isinstance(node, ast.ClassDef) and node.name == "RMSNorm"
              for b in node.body if isinstance(b, ast.FunctionDef) and b.name == "forward_native")
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "Optional": __import__("typing").Optional, "Union": __import__("typing").Union,
          "Tuple": __import__("typing").Tuple}
    exec(compile(mod, path, "exec"), ns)
    forward_native = ns["forward_native"]

    def rmsnorm_fp32(x, w, eps):
        self = types.SimpleNamespace(override_orig_dtype=torch.float32, fp32_residual=False, hidden_size=x.shape[-1],
                                     variance_size_override=None, variance_epsilon=eps, cast_x_before_out_mul=False,
                                     weight=w.float())
        return forward_native(self, x)

    torch.manual_seed(17)
    flat = {}
    cases = [("neox128", 128, 128, True, torch.bfloat16, 10000.0, 1e-5), ("gptj128", 128, 128, False, torch.bfloat16, 10000.0, 1e-5),
             ("partial64of128", 128, 64, True, torch.bfloat16, 500000.0, 1e-6), ("gptj64", 64, 64, False, torch.bfloat16, 10000.0, 1e-5),
             ("neox256", 256, 256, True, torch.bfloat16, 1000000.0, 1e-6), ("partial_gptj32of64", 64, 32, False, torch.float16, 10000.0, 1e-5)]
    for name, D, rot, neox, dt, base, eps in cases:
        n, HQ, HKV = 7, 4, 2
        pos = torch.tensor([100, 101, 102, 0, 1, 4095, 777])
        qkv = torch.randn(n, (HQ + 2 * HKV) * D).to(dt)
        qw, kw = (torch.randn(D) * 5.0).to(dt), (torch.randn(D) * 5.0).to(dt)
        inv_freq = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))
        freqs = torch.einsum("i,j -> ij", pos.float(), inv_freq)
        cos, sin = freqs.cos(), freqs.sin()
        outs = []
        for x, w, H in ((qkv[:, : HQ * D], qw, HQ), (qkv[:, HQ * D: (HQ + HKV) * D], kw, HKV)):
            y = rmsnorm_fp32(x.reshape(-1, D), w, eps).view(n, H, D)
            yr = apply_rotary_emb(y[..., :rot].float(), cos, sin, neox)
            outs.append(torch.cat((yr, y[..., rot:].float()), dim=-1))
        c = dict(qkv=qkv, q_weight=qw, k_weight=kw, positions=pos, eps=np.float64(eps), base=np.float64(base), rotary_dim=rot,
                 is_neox=int(neox), hq=HQ, hkv=HKV, head_dim=D, q_out=outs[0], k_out=outs[1], cos_sin=torch.cat((cos, sin), dim=-1))
        for kk, v in c.items():
            flat[f"{name}.{kk}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("qknorm_rope.npz", **flat)


def f18():
    """F18 multi-step draft decode indices -> draft_kv_indices.npz.  The reference's Triton kernel
    generate_draft_decode_kv_indices (kernels/ops/speculative/cache_locs.py:56-141) under TRITON_INTERPRET=1, launched as
    TritonMultiStepDraftBackend.common_template does (triton_backend.py:1929-1945): grid (speculative_num_steps, num_seqs,
    topk), kv_indices [steps, num_seqs * topk * max_context_len] int64, kv_indptr [steps, max_bs * topk + 1] int32,
    positions = every branch's sequence length (what the draft worker passes in its first step).  Cases: top-k 1 / 4 / 8,
    page size 1 / 16 / 64 (the paged branch layout of top-k > 1), 1-5 steps, ragged lengths incl. page-aligned ones."""
    from sglang.kernels.ops.speculative.cache_locs import generate_draft_decode_kv_indices
    from sglang.srt.utils import next_power_of_2

    rng = np.random.default_rng(18)
    flat = {}
    cases = [("topk1_ps1", 1, 1, 3, [5, 17, 1]), ("topk4_ps1", 4, 1, 4, [33, 2, 100, 64, 7]), ("topk4_ps16", 4, 16, 3, [33, 16, 100, 64, 7]),
             ("topk8_ps64", 8, 64, 5, [130, 64, 1, 200]), ("topk1_ps16", 1, 16, 5, [40, 16]), ("topk2_ps16_step1", 2, 16, 1, [31, 32, 33])]
    for name, topk, ps, steps, lens in cases:
        num_seqs = len(lens)
        max_ctx = max(lens) + (steps + ps) * topk + 8
        pool_rows = num_seqs + 3
        r2t = torch.from_numpy(rng.integers(1, 10 ** 6, size=(pool_rows, max_ctx)).astype(np.int32))
        rpi = torch.from_numpy(rng.permutation(pool_rows)[:num_seqs].astype(np.int64))
        seq = torch.tensor(lens, dtype=torch.int64)
        bs = num_seqs * topk
        positions = seq.repeat_interleave(topk)
        width = num_seqs * topk * max_ctx
        kv_indices = torch.full((steps, width), -1, dtype=torch.int64)
        kv_indptr = torch.zeros((steps, pool_rows * topk + 1), dtype=torch.int32)
        generate_draft_decode_kv_indices[(steps, num_seqs, topk)](
            rpi, r2t, seq, kv_indices, kv_indptr, positions, r2t.shape[1], kv_indices.shape[1], kv_indptr.shape[1],
            next_power_of_2(num_seqs), next_power_of_2(steps), next_power_of_2(bs), ps)
        c = dict(req_to_token=r2t, req_pool_indices=rpi, seq_lens=seq, positions=positions, topk=topk, page_size=ps,
                 num_steps=steps, kv_indices=kv_indices, kv_indptr=kv_indptr)
        for kk, v in c.items():
            flat[f"{name}.{kk}"] = bits(v) if isinstance(v, torch.Tensor) else np.asarray(v)
    save("draft_kv_indices.npz", **flat)


def f19():
    """F19 score_mod = relative_bias_score_mod with aux_tensors = [rel_logits] (kernels/ops/attention/score_mod.py:44-56)
    through the reference's Triton kernels under TRITON_INTERPRET=1 (fp16): extend_attention_fwd (both stages, with a
    prefix longer and shorter than the bias extent, a sliding window, a logit cap), extend_attention_fwd_unified, and
    decode_attention_fwd (grouped and MHA stage-1 kernels, split KV) -> score_bias.npz.  The aux tensor is fp32 in half
    of the cases and fp16 in the others (the kernels add it to the fp32 logits either way)."""
    from sglang.kernels.ops.attention.extend_attention import extend_attention_fwd_unified
    from sglang.kernels.ops.attention.score_mod import relative_bias_score_mod

    torch.manual_seed(19)
    dtype = torch.float16
    rng = np.random.default_rng(190)
    flat = {}

    def put(name, c):
        for k, v in c.items():
            flat[f"{name}.{k}"] = bits(v) if isinstance(v, torch.Tensor) and v.dtype == torch.float16 else (
                v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))

    # ---- two-stage extend: name, Hq, Hkv, D, prefix, extend, extent, aux dtype, window, logit cap
    for name, HQ, HKV, D, pre, ext, extent, adt, window, cap in [
            ("ext_gqa", 8, 2, 128, [40, 0, 7], [6, 20, 9], 16, torch.float32, -1, 0.0),
            ("ext_long", 4, 1, 128, [150, 3], [70, 33], 48, torch.float16, -1, 0.0),
            ("ext_mha64", 4, 4, 64, [33, 5], [5, 9], 64, torch.float32, -1, 0.0),
            ("ext_swa_cap", 8, 2, 128, [70, 10], [12, 40], 20, torch.float16, 24, 30.0)]:
        pre = np.array(pre, dtype=np.int32); ext = np.array(ext, dtype=np.int32)
        B, T = len(pre), int(ext.sum())
        pool = int(pre.sum()) + 9
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(T, HQ, D).to(dtype)
        k_ext = torch.randn(T, HKV, D).to(dtype)
        v_ext = torch.randn(T, HKV, D).to(dtype)
        aux = (2.0 * torch.randn(T, HQ, extent)).to(adt)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(pre))
        qo_indptr = torch.zeros(B + 1, dtype=torch.int64)
        qo_indptr[1:] = torch.from_numpy(np.cumsum(ext))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(pre.sum())] + 1).to(torch.int64)
        o = torch.zeros(T, HQ, D, dtype=dtype)
        extend_attention_fwd(q, k_ext, v_ext, o, kb, vb, qo_indptr, kv_indptr, kv_indices, None, True, None,
                             int(ext.max()), 1.0, 1.0, sm_scale=1.0 / D**0.5, logit_cap=cap, sliding_window_size=window,
                             score_mod=relative_bias_score_mod, aux_tensors=[aux])
        put(name, dict(q=q, k_ext=k_ext, v_ext=v_ext, kb=kb, vb=vb, qo_indptr=qo_indptr, kv_indptr=kv_indptr,
                       kv_indices=kv_indices, sm_scale=1.0 / D**0.5, window=window, cap=cap, aux=aux,
                       aux_f32=int(adt == torch.float32), o=o))
    # ---- unified one-stage extend
    for name, HQ, HKV, D, pre, ext, extent, adt in [("uni_gqa", 8, 2, 128, [40, 0, 7], [6, 20, 9], 24, torch.float16),
                                                    ("uni_mha64", 4, 4, 64, [33, 5], [5, 9], 8, torch.float32)]:
        pre = np.array(pre, dtype=np.int32); ext = np.array(ext, dtype=np.int32)
        B, T = len(pre), int(ext.sum())
        tot = pre + ext
        pool = int(tot.sum()) + 9
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(T, HQ, D).to(dtype)
        aux = (2.0 * torch.randn(T, HQ, extent)).to(adt)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(tot))
        qo_indptr = torch.zeros(B + 1, dtype=torch.int32)
        qo_indptr[1:] = torch.from_numpy(np.cumsum(ext))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(tot.sum())] + 1).to(torch.int64)
        o = torch.zeros(T, HQ, D, dtype=dtype)
        extend_attention_fwd_unified(q, o, kb, vb, 1.0, 1.0, qo_indptr, kv_indptr, kv_indices, torch.from_numpy(pre),
                                     int(ext.max()), sm_scale=1.0 / D**0.5, is_causal=True,
                                     score_mod=relative_bias_score_mod, aux_tensors=[aux])
        put(name, dict(q=q, kb=kb, vb=vb, qo_indptr=qo_indptr, kv_indptr=kv_indptr, kv_indices=kv_indices,
                       prefix_lens=torch.from_numpy(pre), sm_scale=1.0 / D**0.5, aux=aux, aux_f32=int(adt == torch.float32), o=o))
    # ---- decode (grouped and MHA stage 1), two kv splits
    for name, HQ, HKV, D, lens, extent, adt in [("dec_gqa", 8, 2, 128, [5, 40, 200, 17], 32, torch.float32),
                                                ("dec_mha64", 4, 4, 64, [100, 3], 8, torch.float16),
                                                ("dec_wide", 8, 1, 128, [70, 9], 128, torch.float16)]:
        lens = np.array(lens, dtype=np.int32)
        B = len(lens)
        pool = int(lens.sum()) + 5
        kb = torch.randn(pool, HKV, D).to(dtype)
        vb = torch.randn(pool, HKV, D).to(dtype)
        q = torch.randn(B, HQ, D).to(dtype)
        aux = (2.0 * torch.randn(B, HQ, extent)).to(adt)
        kv_indptr = torch.zeros(B + 1, dtype=torch.int32)
        kv_indptr[1:] = torch.from_numpy(np.cumsum(lens))
        kv_indices = torch.from_numpy(rng.permutation(pool - 1)[: int(lens.sum())] + 1).to(torch.int64)
        S = 4
        nsplit = torch.full((B,), 2, dtype=torch.int32)
        o = torch.zeros(B, HQ, D, dtype=dtype)
        al = torch.zeros(B, HQ, S, D, dtype=torch.float32)
        lse = torch.zeros(B, HQ, S, dtype=torch.float32)
        decode_attention_fwd(q, kb, vb, o, kv_indptr, kv_indices, al, lse, nsplit, S, 1.0 / D**0.5, 1.0, 1.0,
                             score_mod=relative_bias_score_mod, aux_tensors=[aux])
        put(name, dict(q=q, kb=kb, vb=vb, kv_indptr=kv_indptr, kv_indices=kv_indices, nsplit=nsplit, max_splits=S,
                       sm_scale=1.0 / D**0.5, aux=aux, aux_f32=int(adt == torch.float32), o=o))
    save("score_bias.npz", **flat)


def f20():
    """F20 build_unified_kv_indices (kernels/ops/attention/extend_attention.py:193-238, Triton copy kernel under
    TRITON_INTERPRET=1) -> unified_kv_indices.npz: ragged prefixes incl. empty ones, extends of 1 .. 300 tokens, int64
    prefix indices + int64 out_cache_loc (what TritonAttnBackend._forward_extend_unified passes), and a case with
    int32 extend_seq_lens / extend_start_loc as ForwardBatch carries them."""
    from sglang.kernels.ops.attention.extend_attention import build_unified_kv_indices

    rng = np.random.default_rng(20)
    flat = {}
    for name, pre, ext, ldt in [("ragged", [40, 0, 7, 300], [6, 20, 9, 1], torch.int32),
                                ("no_prefix", [0, 0], [129, 300], torch.int64),
                                ("one", [513], [64], torch.int32),
                                ("many", list(rng.integers(0, 90, size=37)), list(rng.integers(1, 50, size=37)), torch.int32)]:
        pre = np.array(pre, dtype=np.int64); ext = np.array(ext, dtype=np.int64)
        bs = len(pre)
        pool = int(pre.sum() + ext.sum()) + 11
        perm = rng.permutation(pool - 1) + 1
        prefix_idx = torch.from_numpy(perm[: int(pre.sum())].astype(np.int64))
        ext_idx = torch.from_numpy(perm[int(pre.sum()): int(pre.sum() + ext.sum())].astype(np.int64))
        indptr = torch.zeros(bs + 1, dtype=torch.int32)
        indptr[1:] = torch.from_numpy(np.cumsum(pre))
        start = torch.from_numpy(np.concatenate([[0], np.cumsum(ext)[:-1]])).to(ldt)
        lens = torch.from_numpy(ext).to(ldt)
        u_indptr, u_idx, p_lens = build_unified_kv_indices(indptr, prefix_idx, start, lens, ext_idx, bs)
        total = int(u_indptr[-1])
        c = dict(prefix_kv_indptr=indptr, prefix_kv_indices=prefix_idx, extend_start_loc=start, extend_seq_lens=lens,
                 extend_kv_indices=ext_idx, unified_kv_indptr=u_indptr.to(torch.int32), unified_kv_indices=u_idx[:total],
                 prefix_lens=p_lens.to(torch.int32))
        for k, v in c.items():
            flat[f"{name}.{k}"] = v.numpy()
    save("unified_kv_indices.npz", **flat)



def f21():
    """F21 fused_fp8_qkv_kv_cache (kernels/ops/kvcache/fused_fp8_qkv_kv_cache.py:35-80) -> fused_fp8_qkv.npz.  The operator
    is a CUDA JIT kernel (kernels/jit/csrc/attention/fused_fp8_qkv_kv_cache.cuh) that cannot run in this container; the
    expected bytes are the reference TEST's own expectation (test/registered/kernels/ops/kvcache/
    test_fused_fp8_qkv_kv_cache.py:13-16,76-90): ``(x.float() * inv_scale).clamp(-448, 448).to(float8_e4m3fn)`` with
    ``inv_scale = 1 / float(scale)`` for K / V and ``q.to(float8_e4m3fn)`` for q, computed with torch on the CPU.  Cases: the
    test's (hq, hkv, head_dim) table x {bf16, fp16} x scale {None, 0.5, 2.0} at 1 / 5 / 33 tokens, rows sliced from a fused
    qkv tensor, plus one case of edge values (+-448, values that saturate, ties, subnormals, -0.0)."""
    FP8 = torch.float8_e4m3fn
    g = torch.Generator().manual_seed(21)
    flat = {}
    n_case = 0
    for dtype, dn in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        for hq, hkv, hd in ((8, 1, 128), (8, 8, 128), (4, 2, 64), (64, 2, 128)):
            for n in ((1, 5) if hq == 64 else (1, 5, 33)):
                for scale in (None, 0.5, 2.0):
                    q_dim, kv_dim = hq * hd, hkv * hd
                    qkv = torch.randn(n, q_dim + 2 * kv_dim, generator=g).to(dtype)
                    q, k, v = qkv[:, :q_dim], qkv[:, q_dim: q_dim + kv_dim], qkv[:, q_dim + kv_dim:]
                    slots = n + 4
                    loc = torch.randperm(slots, generator=g)[:n].to(torch.int64)
                    inv_k = 1.0 if scale is None else 1.0 / float(torch.tensor(scale, dtype=torch.float32))
                    inv_v = 1.0 if scale is None else 1.0 / float(torch.tensor(scale * 1.5, dtype=torch.float32))
                    name = f"c{n_case}"
                    n_case += 1
                    flat[name + ".meta"] = np.array([hq, hkv, hd, n, slots, 0 if scale is None else 1, dtype == torch.bfloat16], dtype=np.int64)
                    flat[name + ".scale"] = np.array([1.0 if scale is None else scale, 1.0 if scale is None else scale * 1.5], dtype=np.float32)
                    flat[name + ".qkv"] = bits(qkv) if dtype == torch.bfloat16 else qkv.numpy().copy()
                    flat[name + ".loc"] = loc.numpy()
                    flat[name + ".q_fp8"] = q.to(FP8).view(torch.uint8).numpy().copy()
                    flat[name + ".k_fp8"] = (k.float() * inv_k).clamp(-448.0, 448.0).to(FP8).view(torch.uint8).numpy().copy()
                    flat[name + ".v_fp8"] = (v.float() * inv_v).clamp(-448.0, 448.0).to(FP8).view(torch.uint8).numpy().copy()
    # edge values through the K / V formula (clamped) -- q is kept inside the finite range (torch's q.to(FP8) makes NaN
    # beyond 448 where the kernel's satfinite cast gives 448; the reference test never leaves the range)
    edge = torch.tensor([0.0, -0.0, 448.0, -448.0, 449.0, 464.0, 465.0, 1000.0, -1e5, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10,
                         0.0009765625 * 1.5, 17.0, 18.0, 19.0, 20.0, 21.0, 22.0, 23.0, 24.0, 25.0, 26.0, 27.0, 28.0, 0.4375, 0.46875,
                         0.40625, 240.0, 232.0, 248.0, 1.0625], dtype=torch.float32)
    for dtype, dn in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        x = edge.to(dtype)
        for sc in (1.0, 0.5, 3.0, 0.3):
            inv = 1.0 / float(torch.tensor(sc, dtype=torch.float32))
            flat[f"edge_{dn}_{sc}.x"] = bits(x) if dtype == torch.bfloat16 else x.numpy().copy()
            flat[f"edge_{dn}_{sc}.fp8"] = (x.float() * inv).clamp(-448.0, 448.0).to(FP8).view(torch.uint8).numpy().copy()
    flat["n_cases"] = np.array([n_case])
    save("fused_fp8_qkv.npz", **flat)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7", "f8", "f9", "f10", "f11", "f12", "f13", "f14",
                             "f15", "f16", "f17", "f18", "f19", "f20", "f21"]
    for w in which:
        globals()[w]()
