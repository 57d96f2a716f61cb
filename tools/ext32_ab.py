#!/usr/bin/env python3
"""A/B timing of rx::extend_mfma32_kernel variants on the config-3 chunk (bench.py's extend leg), interleaved in ONE
process (cdna_hip_programming.md rule 24):
    VARIANTS=0,1,2 python tools/ext32_ab.py            # builds tools/probe/libext32_dev.so if needed (here, with hipcc)
    ZERO=1 ...                                         # all-zero operands: same instruction stream, minimal switching power
Prints per variant: median / min ms per chunk, TFLOP/s, and max |o - o_variant0| (the variants must agree)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.environ.get("DEV", "32")  # DEV=64: rx::extend_mfma64_kernel's dev variants (tools/probe/ext64_dev.hip: VARIANTS = its bit masks)
LIB = os.path.join(ROOT, "tools", "probe", f"libext{DEV}_dev.so")
SRC = os.path.join(ROOT, "tools", "probe", f"ext{DEV}_dev.hip")
VARS = [int(x) for x in os.environ.get("VARIANTS", "0").split(",")]


def build():
    from sglang_amd import build as b
    deps = [SRC, os.path.join(b.CSRC, "rx_extend32_kernel.inc"), os.path.join(b.CSRC, "rx_extend64_kernel.inc"),
            os.path.join(b.CSRC, "rx_common.h")]
    tag = LIB + ".vars"
    want = ",".join(str(v) for v in sorted(set(VARS)))
    if (os.path.exists(LIB) and os.path.exists(tag) and open(tag).read() == want
            and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps)):
        return
    cases = " ".join(f"RX_V({v})" for v in sorted(set(VARS)) if v != 0)
    cmd = [b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           *b.EXTRA_FLAGS[f"rx_extend{DEV}.hip"], "-I", os.path.join(ROOT, "include"), "-I", b.CSRC,
           f"-DRX_DEV_VARIANT_CASES={cases}", SRC, "-o", LIB]
    print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    open(tag, "w").write(want)


def main():
    build()
    if os.environ.get("BUILD_ONLY"):
        return
    import torch
    from sglang_amd import ops
    from sglang_amd import lib as rxlib

    dev = torch.device("cuda:0")
    HQ, HKV, D = 32, 8, 128
    P, E, chunk, ps = (int(x) for x in os.environ.get("SHAPE", "3584,512,32,16").split(","))
    g = torch.Generator(device=dev).manual_seed(1)
    n_pages = (P + ps - 1) // ps + 1
    kb = torch.randn((n_pages, HKV, ps, D), device=dev, generator=g).to(torch.bfloat16)
    vb = torch.randn((n_pages, HKV, ps, D), device=dev, generator=g).to(torch.bfloat16)
    lay = ops.kv_layout_hnd(kb, vb)
    T = chunk * E
    q = torch.randn(T, HQ, D, device=dev, generator=g).to(torch.bfloat16)
    ke = torch.randn(T, HKV, D, device=dev, generator=g).to(torch.bfloat16)
    ve = torch.randn(T, HKV, D, device=dev, generator=g).to(torch.bfloat16)
    if os.environ.get("ZERO"):
        for t_ in (kb, vb, q, ke, ve):
            t_.zero_()
    pages = torch.randperm(n_pages - 1, device=dev, generator=g)[: (P + ps - 1) // ps] + 1
    slots = (pages[:, None] * ps + torch.arange(ps, device=dev)[None, :]).reshape(-1)[:P].to(torch.int64)
    kvi = slots.repeat(chunk)
    kvp = (torch.arange(chunk + 1, device=dev) * P).to(torch.int32)
    qo = (torch.arange(chunk + 1, device=dev) * E).to(torch.int64)
    outs = {v: torch.zeros(T, HQ, D, device=dev, dtype=torch.bfloat16) for v in VARS}
    dl = C.CDLL(LIB)
    entry = getattr(dl, f"rx_dev_extend{DEV}")
    entry.restype = C.c_int
    entry.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    causal = not os.environ.get("NONCAUSAL")  # NONCAUSAL=1: every new-token tile is a full tile (what the diagonal costs: compare the times)
    params = {v: ops._extend_params(q, ke, ve, outs[v], kb, vb, qo, kvp, kvi, None, causal, None, E, 1.0, 1.0,
                                    sm_scale=D ** -0.5, page_size=ps, kv_layout=lay) for v in VARS}
    stream = torch.cuda.current_stream().cuda_stream

    def run(v):
        rc = entry(C.byref(params[v]), v, stream)
        assert rc == 0, (v, rc)

    # production kernel as the reference output
    o_ref = torch.zeros_like(outs[VARS[0]])
    ops.extend_attention_fwd(q, ke, ve, o_ref, kb, vb, qo, kvp, kvi, None, causal, None, E, 1.0, 1.0, sm_scale=D ** -0.5,
                             page_size=ps, kv_layout=lay)
    print("production instance:", rxlib.last_dispatch())
    for v in VARS:
        run(v)
    torch.cuda.synchronize()
    for v in VARS:
        d = (outs[v].float() - o_ref.float()).abs().max().item()
        print(f"variant {v}: max |o - production| = {d:.3e}  nan={bool(torch.isnan(outs[v].float()).any())}")
    rounds, reps = int(os.environ.get("ROUNDS", "12")), int(os.environ.get("REPS", "4"))
    times = {v: [] for v in VARS}
    for _ in range(3):
        for v in VARS:
            run(v)
    torch.cuda.synchronize()
    for r in range(rounds):
        for v in (VARS if r % 2 == 0 else VARS[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / reps)
    flops = 4.0 * HQ * D * chunk * (E * P + (E * (E + 1) / 2 if causal else E * E))
    for v in VARS:
        t = sorted(times[v])
        med, mn = t[len(t) // 2], t[0]
        print(f"variant {v}: median {med:.4f} ms  min {mn:.4f} ms  ->  {flops / med / 1e9:.1f} TFLOP/s (median)  {flops / mn / 1e9:.1f} (best)")


if __name__ == "__main__":
    main()
