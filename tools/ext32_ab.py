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
STAMPS = bool(os.environ.get("STAMPS"))  # STAMPS=1 (DEV=32): per-block clock stamps of ONE launch (rx_extend32_kernel.inc RX_EXT32_TIMELINE)
LIB = os.path.join(ROOT, "tools", "probe", f"libext{DEV}_{'stamps' if STAMPS else 'dev'}.so")
SRC = os.path.join(ROOT, "tools", "probe", f"ext{DEV}_dev.hip")
VARS = [int(x) for x in os.environ.get("VARIANTS", "0").split(",")]


def build():
    from sglang_amd import build as b
    deps = [SRC, os.path.join(b.CSRC, "rx_extend32_kernel.inc"), os.path.join(ROOT, "tools", "probe", "rx_extend64_kernel.inc"),
            os.path.join(b.CSRC, "rx_common.h")]
    tag = LIB + ".vars"
    want = ",".join(str(v) for v in sorted(set(VARS)))
    if (os.path.exists(LIB) and os.path.exists(tag) and open(tag).read() == want
            and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps)):
        return
    cases = " ".join(f"RX_V({v})" for v in sorted(set(VARS)) if v != 0)
    cmd = [b._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           *b.EXTRA_FLAGS[f"rx_extend{DEV}.hip"], "-I", os.path.join(ROOT, "include"), "-I", b.CSRC,
           f"-DRX_DEV_VARIANT_CASES={cases}", *(["-DRX_EXT32_TIMELINE"] if STAMPS else []), SRC, "-o", LIB]
    print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    open(tag, "w").write(want)


def stamps_report(dl, run, v, nblocks):
    """One launch's per-block stamps (10-ns ticks of s_memrealtime): where a block's time goes and how the CUs' timelines end."""
    import numpy as np
    import torch
    for _ in range(8):
        run(v)
    torch.cuda.synchronize()
    buf = np.zeros((nblocks, 8), dtype=np.uint64)
    assert dl.rx_dev_stamps(buf.ctypes.data_as(C.c_void_p), nblocks) == 0
    st = buf[:, :5].astype(np.int64)
    cyc = (buf[:, 6].astype(np.int64) - st[:, 4])
    st[:, 4] = st[:, 3]
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0  # ticks -> us
    pro, loop, epi, drain = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 4] - us[:, 3]
    nt = buf[:, 7].astype(np.int64)
    hw, xcc = buf[:, 5].astype(np.int64) & 0xFFFFFFFF, (buf[:, 5].astype(np.int64) >> 32) & 0xF
    mhz = cyc / np.maximum(us[:, 3] - us[:, 0], 1e-3)
    print(f"shader clock over a block (s_memtime cycles / s_memrealtime us), MHz p5 / p50 / p95: {np.percentile(mhz, 5):.0f} / {np.percentile(mhz, 50):.0f} / {np.percentile(mhz, 95):.0f}")
    cu = (xcc << 8) | (((hw >> 12) & 0xF) << 4) | ((hw >> 8) & 0xF)  # (XCC, SE | SH, CU) of HW_REG_HW_ID
    print(f"blocks {nblocks}, distinct CUs {len(set(cu.tolist()))}, kernel span {us[:, 4].max():.1f} us")
    q = lambda a: " / ".join(f"{np.percentile(a, p):.2f}" for p in (5, 50, 95))  # noqa: E731
    print(f"per block, us (p5 / p50 / p95): prologue {q(pro)} | tile loop {q(loop)} ({q(loop / np.maximum(nt, 1))} per tile, {int(np.median(nt))} tiles) | "
          f"epilogue to last store issued {q(epi)} | stores complete {q(drain)} | whole block {q(us[:, 4] - us[:, 0])}")
    gaps, ends, firsts = [], [], []
    for c in set(cu.tolist()):
        rows = us[cu == c]
        rows = rows[np.argsort(rows[:, 0])]
        gaps += (rows[1:, 0] - rows[:-1, 4]).tolist()
        ends.append(rows[-1, 4])
        firsts.append(rows[0, 0])
    ends, firsts = np.array(ends), np.array(firsts)
    print(f"per CU: first block starts at {q(firsts)} us; gap between a block's end and the next block's start {q(np.array(gaps))} us; "
          f"last block ends at {q(ends)} us (min {ends.min():.1f}, max {ends.max():.1f}): idle tail mean {(ends.max() - ends).mean():.1f} us")
    busy = sum(float((us[cu == c][:, 4] - us[cu == c][:, 0]).sum()) for c in set(cu.tolist()))
    print(f"sum of block times / (CUs x span) = {busy / (len(set(cu.tolist())) * us[:, 4].max()):.3f}; blocks per CU min / max "
          f"{min(int((cu == c).sum()) for c in set(cu.tolist()))} / {max(int((cu == c).sum()) for c in set(cu.tolist()))}")


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
    if STAMPS:
        for _ in range(10):  # settle the clocks on the mix of variants first
            for v in VARS:
                for _ in range(4):
                    run(v)
        for v in VARS:
            print(f"--- variant {v}")
            stamps_report(dl, run, v, chunk * HKV * ((E * (HQ // HKV) + 255) // 256))
        return
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
