"""The peer-to-peer all-reduce kernels with all W "ranks" inside ONE process -- W regions, W contexts, W streams, the W
kernels of a call running side by side on one GPU:
    python tools/allreduce_loopback.py [W=2]
    cd /tmp && rocprofv3 --kernel-trace --stats -d out -o ar --output-format csv -- python3 <repo>/tools/allreduce_loopback.py
No IPC, no launcher, no child process: the one form of these collectives that rocprofv3 can wrap directly (its kernel
durations are what profiles/rNN_allreduce_loopback_kernel_stats.csv holds).  A rank's kernel spins until its peers' kernels
have staged / reduced, so a kernel's duration includes waiting for the LAST of the W launches: it is the latency of the
collective as one rank sees it, on one GPU (no link involved).  W = 2 is what this tool is for: HIP multiplexes a
process's streams onto a few hardware queues (4 by default), and two "ranks" whose streams share a queue deadlock -- the
first kernel spins on a flag that the kernel queued BEHIND it would raise (W = 4 hung until its spin limit; one process per
GPU, the real deployment, has no such coupling).  The tool lowers the spin limit so that such a hang ends in seconds.
(`--pmc` passes are NOT possible: counter collection serialises kernels, the two ranks' kernels can no longer run side by
side and every flag wait ends in its timeout -- tried, round 6.)  Results are checked: two-shot / one-shot / fused against the
fp32 rank-order sum, every quick level against the oracle."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import radix_oracle as O  # noqa: E402  (a tool, not the product: the checker)
from sglang_amd import lib as L  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    lib = L.load()
    L.set_option("ar_spin_log2", 21)   # (a flag wait that cannot be satisfied gives up after ~2 s, see the docstring)
    cp = C.c_void_p
    streams = [torch.cuda.Stream() for _ in range(W)]
    err = torch.zeros(1, dtype=torch.int32, device=dev)

    def regions(nbytes):
        out = []
        for _ in range(W):
            p = cp()
            L.check(lib.rx_ar_alloc_region(nbytes, C.byref(p)), "rx_ar_alloc_region")
            out.append(p)
        return out

    max_bytes = 8 << 20
    ar_regions = regions(lib.rx_ar_region_bytes(max_bytes))
    qr_regions = regions(lib.rx_qr_region_bytes())
    ar_ctx, qr_ctx = [], []
    for r in range(W):
        ptrs = (cp * W)(*[p.value for p in ar_regions])
        c = cp()
        L.check(lib.rx_ar_init(C.byref(c), r, W, ptrs, max_bytes, cp(err.data_ptr())), "rx_ar_init")
        ar_ctx.append(c)
        ptrs = (cp * W)(*[p.value for p in qr_regions])
        c = cp()
        L.check(lib.rx_qr_init(C.byref(c), r, W, ptrs, cp(err.data_ptr())), "rx_qr_init")
        qr_ctx.append(c)

    def bits(t):
        return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1)

    def timed(launch, reps=20):
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(W)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(W)]
        for r in range(W):
            e0[r].record(streams[r])
        for _ in range(reps):
            launch()
        for r in range(W):
            e1[r].record(streams[r])
        torch.cuda.synchronize()
        return max(e0[r].elapsed_time(e1[r]) for r in range(W)) / reps * 1e3

    res = []
    H = 4096
    for kib in (256, 2048):
        n = kib * 1024 // 2
        g = torch.Generator().manual_seed(kib)
        parts = [torch.randn(n, generator=g).to(torch.bfloat16) for _ in range(W)]
        want = sum(p.float() for p in parts).to(torch.bfloat16)
        xs = [p.to(dev) for p in parts]
        ys = [torch.empty_like(x) for x in xs]
        resid = [torch.randn(n // H, H, device=dev).to(torch.bfloat16) for _ in range(W)]
        wgt = torch.ones(H, device=dev, dtype=torch.bfloat16)
        outs = [torch.empty(n // H, H, device=dev, dtype=torch.bfloat16) for _ in range(W)]

        def two_shot(fn=lib.rx_allreduce):
            for r in range(W):
                L.check(fn(ar_ctx[r], cp(xs[r].data_ptr()), cp(ys[r].data_ptr()), n, L.RX_BF16, cp(streams[r].cuda_stream)), "ar")

        def fused():
            for r in range(W):
                L.check(lib.rx_allreduce_rmsnorm(ar_ctx[r], cp(xs[r].data_ptr()), cp(resid[r].data_ptr()), cp(wgt.data_ptr()),
                                                 cp(outs[r].data_ptr()), cp(resid[r].data_ptr()), n // H, H, 1e-6, L.RX_BF16,
                                                 cp(streams[r].cuda_stream)), "fused")

        for kind, fn in (("two_shot", two_shot), ("one_shot_det", lambda: two_shot(lib.rx_allreduce_det)), ("fused_rmsnorm", fused)):
            us = timed(fn)
            ok = True
            if kind != "fused_rmsnorm":
                ok = all(torch.equal(y.cpu(), want) for y in ys)
            res.append({"kind": kind, "world": W, "message_KiB": kib, "us_per_call": round(us, 2), "matches_fp32_rank_order_sum": ok})
    # the quick all-reduce: 64 MiB, every level, fp16
    n = 32 << 20
    g = torch.Generator().manual_seed(5)
    small = [torch.randn(16384 * 3 + 72, generator=g).to(torch.float16) for _ in range(W)]
    xs = [torch.randn(n, device=dev).to(torch.float16) for _ in range(W)]
    ys = [torch.empty_like(x) for x in xs]
    if W in (2, 4, 8):
        for level, name in ((0, "FP"), (1, "INT8"), (2, "INT6"), (3, "INT4")):
            sx = [p.to(dev) for p in small]
            sy = [torch.empty_like(p) for p in sx]
            for r in range(W):
                L.check(lib.rx_quick_allreduce(qr_ctx[r], cp(sx[r].data_ptr()), cp(sy[r].data_ptr()), sx[r].numel(), L.RX_F16, level, 0,
                                               cp(streams[r].cuda_stream)), "qr")
            torch.cuda.synchronize()
            want = O.quick_allreduce([bits(p) for p in small], False, level)
            ok = all(np.array_equal(bits(y), want) for y in sy)

            def quick():
                for r in range(W):
                    L.check(lib.rx_quick_allreduce(qr_ctx[r], cp(xs[r].data_ptr()), cp(ys[r].data_ptr()), n, L.RX_F16, level, 0,
                                                   cp(streams[r].cuda_stream)), "qr")
            us = timed(quick, reps=10)
            res.append({"kind": "quick_" + name, "world": W, "message_KiB": 65536, "us_per_call": round(us, 2),
                        "message_GB_per_s_per_rank": round(64 / 1024 * 1.073741824 / (us / 1e6), 1), "matches_oracle_bit_for_bit": ok})
    assert int(err.item()) == 0, "device error word set"
    for r in res:
        print(json.dumps(r))
    assert all(v for r in res for k, v in r.items() if k.startswith("matches"))
    for c in ar_ctx:
        lib.rx_ar_destroy(c)
    for c in qr_ctx:
        lib.rx_qr_destroy(c)
    for p in ar_regions + qr_regions:
        lib.rx_ar_free_region(p)


if __name__ == "__main__":
    main()
