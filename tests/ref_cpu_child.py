"""CPU-only child of tests/test_gpu_vs_reference_cpu.py (and bench.py's parity figure): builds one seeded
full-size case on the host, runs the REFERENCE's own compiled CPU kernel on it (oracle/_ref: decode_attention_cpu /
extend_attention_cpu, aot/csrc/cpu/decode.cpp:1586, extend.cpp:425, built by oracle/build_ref.py) and leaves inputs and
the kernel's output as .npy files in --out.  A child because (a) the reference build may use ISA the host lacks (SIGILL
must not take pytest down), (b) its OpenMP pool stays out of the GPU process.  TEST INFRASTRUCTURE: never imported by
sglang_amd/.

bf16 tensors travel as uint16 bit patterns.  K/V pools are written in the reference's NHD shape [slots, Hkv, D]; the
parent re-lays them out as HND pages for the HIP pool (same values)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bits(t):
    return t.contiguous().view(torch.uint16).numpy()


def normal_bf16(g, *shape):
    """N(0, 1) bf16 values: a 16 Mi-element base drawn once and tiled (generating 2 x 2 GiB of normals costs ~10 s of the GPU
    suite; both sides read the SAME tensor, which is all a parity test needs -- a request's shuffled slots still hold
    distinct rows)."""
    n = 1
    for d in shape:
        n *= d
    base = torch.empty(min(n, 1 << 24), dtype=torch.bfloat16).normal_(generator=g)
    reps = -(-n // base.numel())
    return (base if reps == 1 else base.repeat(reps)[:n]).view(*shape).contiguous()


def shuffled_slots(rng, bs, pages_per_req, ps, first_page=1):
    perm = rng.permutation(np.arange(first_page, first_page + bs * pages_per_req))
    return (perm.reshape(bs, pages_per_req)[:, :, None] * ps + np.arange(ps)[None, None, :]).reshape(bs, -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", choices=["decode", "extend"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--min-ctx", type=int, default=0, help="decode: ragged lengths uniform in [min-ctx, ctx]")
    ap.add_argument("--hq", type=int, default=32)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--ps", type=int, default=16)
    ap.add_argument("--prefix", type=int, default=3584, help="extend: shared cached prefix length")
    ap.add_argument("--extend", type=int, default=512, help="extend: new tokens per request")
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    from oracle import build_ref

    m = build_ref.load()
    if m is None:
        print(json.dumps({"error": "oracle/_ref not built"}))
        return 3
    torch.set_num_threads(os.cpu_count())
    g = torch.Generator().manual_seed(a.seed)
    rng = np.random.default_rng(a.seed)
    HQ, HKV, D, ps, bs = a.hq, a.hkv, a.d, a.ps, a.bs
    os.makedirs(a.out, exist_ok=True)
    save = lambda name, arr: np.save(os.path.join(a.out, name + ".npy"), arr)  # noqa: E731
    if a.kind == "decode":
        ctx = a.ctx
        ppr = (ctx + ps - 1) // ps
        slots_n = (bs * ppr + 1) * ps
        kb = normal_bf16(g, slots_n, HKV, D)
        vb = normal_bf16(g, slots_n, HKV, D)
        q = torch.randn(bs, HQ, D, generator=g).to(torch.bfloat16)
        k_new = torch.randn(bs, HKV, D, generator=g).to(torch.bfloat16)
        v_new = torch.randn(bs, HKV, D, generator=g).to(torch.bfloat16)
        slots = shuffled_slots(rng, bs, ppr, ps)
        r2t = torch.zeros(bs + 1, ppr * ps, dtype=torch.int32)
        r2t[1:] = torch.from_numpy(slots.astype(np.int32))
        rpi = torch.arange(1, bs + 1, dtype=torch.int64)
        lens_np = (rng.integers(a.min_ctx, ctx + 1, size=bs) if a.min_ctx else np.full(bs, ctx)).astype(np.int64)
        lens = torch.from_numpy(lens_np)
        loc = r2t[1:].gather(1, (lens - 1).view(-1, 1)).view(-1).to(torch.int64)
        # inputs BEFORE the call: the kernel writes k_new / v_new into the pools at loc (the step's store)
        save("k_buffer", bits(kb)); save("v_buffer", bits(vb)); save("q", bits(q))
        save("k_new", bits(k_new)); save("v_new", bits(v_new)); save("req_to_token", r2t.numpy())
        save("seq_lens", lens_np); save("loc", loc.numpy())
        out = torch.zeros(bs, HQ, D, dtype=torch.bfloat16)
        attn_logits = torch.zeros(bs, HQ, 8, D + 1)
        t0 = time.perf_counter()
        m.decode_attention_cpu(q, kb, vb, out, k_new, v_new, loc, attn_logits, r2t, rpi, lens, D ** -0.5, 0.0, False, 0,
                               None, None)
        dt = time.perf_counter() - t0
        save("out", bits(out))
        # the rows the kernel stored (the parent checks the fused store's bytes against them)
        save("k_stored", bits(kb[loc])); save("v_stored", bits(vb[loc]))
    else:
        P, E, chunk = a.prefix, a.extend, bs
        npp, npe = (P + ps - 1) // ps, (E + ps - 1) // ps
        n_pages = npp + chunk * npe + 1
        kb = torch.empty(n_pages * ps, HKV, D, dtype=torch.bfloat16).normal_(generator=g)
        vb = torch.empty(n_pages * ps, HKV, D, dtype=torch.bfloat16).normal_(generator=g)
        T = chunk * E
        q = torch.randn(T, HQ, D, generator=g).to(torch.bfloat16)
        k_ext = torch.randn(T, HKV, D, generator=g).to(torch.bfloat16)
        v_ext = torch.randn(T, HKV, D, generator=g).to(torch.bfloat16)
        perm = rng.permutation(np.arange(1, n_pages))
        pre_slots = (perm[:npp, None] * ps + np.arange(ps)[None]).reshape(-1)[:P]
        r2t = torch.zeros(chunk + 1, P + E + ps, dtype=torch.int32)
        for i in range(chunk):
            own = (perm[npp + i * npe: npp + (i + 1) * npe, None] * ps + np.arange(ps)[None]).reshape(-1)[:E]
            r2t[i + 1, :P] = torch.from_numpy(pre_slots.astype(np.int32))
            r2t[i + 1, P: P + E] = torch.from_numpy(own.astype(np.int32))
        rpi = torch.arange(1, chunk + 1, dtype=torch.int64)
        seq = torch.full((chunk,), P + E, dtype=torch.int64)
        ext = torch.full((chunk,), E, dtype=torch.int32)
        start = (torch.arange(chunk, dtype=torch.int32) * E)
        save("k_buffer", bits(kb)); save("v_buffer", bits(vb)); save("q", bits(q))
        save("k_extend", bits(k_ext)); save("v_extend", bits(v_ext)); save("req_to_token", r2t.numpy())
        out = torch.zeros(T, HQ, D, dtype=torch.bfloat16)
        t0 = time.perf_counter()
        m.extend_attention_cpu(q, k_ext, v_ext, out, kb, vb, r2t, rpi, seq, ext, start, E, D ** -0.5, 0.0, False, 0, None,
                               None, None)
        dt = time.perf_counter() - t0
        save("out", bits(out))
    print(json.dumps({"ok": True, "seconds": dt, "threads": torch.get_num_threads()}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
