"""SURVEY 8(d), CPU baseline item (1): in the BUILD container time (a) the reference's own compiled CPU decode
kernel (oracle/_ref, built from /root/reference/.../aot/csrc/cpu by oracle/build_ref.py) and (b) this repo's
C restatement (oracle/rx_oracle.c) on IDENTICAL inputs at the bench shape, check them against each other, and
write the results fixture tests/golden/cpu_baseline_container.json.  Test infrastructure only.

    python oracle/time_cpu_container.py            # ~1-2 minutes on the 8-vCPU container
"""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(kind, seconds=12, extra=()):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", kind, "--cpu-seconds",
                          str(seconds), *extra], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    return json.loads(out)


def _parity_extend():
    """extend_attention_cpu (extend.cpp:425) vs the C restatement on one small identical radix-hit input."""
    from oracle import build_ref, c_oracle

    m = build_ref.load()
    B, P, E, ps, HQ, HKV, D = 3, 200, 70, 16, 32, 8, 128
    g = torch.Generator().manual_seed(2)
    n_pages = (P + ps - 1) // ps + B * ((E + ps - 1) // ps) + 1
    kb = torch.randn(n_pages * ps, HKV, D, generator=g).to(torch.bfloat16)
    vb = torch.randn(n_pages * ps, HKV, D, generator=g).to(torch.bfloat16)
    T = B * E
    q = torch.randn(T, HQ, D, generator=g).to(torch.bfloat16)
    ke = torch.randn(T, HKV, D, generator=g).to(torch.bfloat16)
    ve = torch.randn(T, HKV, D, generator=g).to(torch.bfloat16)
    perm = np.random.default_rng(1).permutation(np.arange(1, n_pages))
    npp, npe = (P + ps - 1) // ps, (E + ps - 1) // ps
    pre = (perm[:npp, None] * ps + np.arange(ps)[None]).reshape(-1)[:P]
    r2t = torch.zeros(B + 1, P + E + ps, dtype=torch.int32)
    for i in range(B):
        own = (perm[npp + i * npe: npp + (i + 1) * npe, None] * ps + np.arange(ps)[None]).reshape(-1)[:E]
        r2t[i + 1, :P] = torch.from_numpy(pre.astype(np.int32))
        r2t[i + 1, P: P + E] = torch.from_numpy(own.astype(np.int32))
    out = torch.zeros(T, HQ, D, dtype=torch.bfloat16)
    m.extend_attention_cpu(q, ke, ve, out, kb, vb, r2t, torch.arange(1, B + 1, dtype=torch.int64),
                           torch.full((B,), P + E, dtype=torch.int64), torch.full((B,), E, dtype=torch.int32),
                           (torch.arange(B, dtype=torch.int32) * E), E, D ** -0.5, 0.0, False, 0, None, None, None)
    bits = lambda t: t.contiguous().view(torch.uint16).numpy()  # noqa: E731
    port = c_oracle.extend_bf16(bits(q), bits(ke), bits(ve), bits(kb), bits(vb), (np.arange(B + 1) * E).astype(np.int64),
                                (np.arange(B + 1) * P).astype(np.int32), np.tile(pre.astype(np.int64), B), D ** -0.5)
    port = torch.from_numpy(np.ascontiguousarray(port)).view(torch.bfloat16).float().numpy().astype(np.float64)
    return float(np.abs(out.float().numpy().astype(np.float64) - port.reshape(T, HQ, D)).max())


def _parity():
    """Reference kernel vs C restatement on one small identical input (bf16 bits in, bf16 / fp32 out)."""
    from oracle import build_ref, c_oracle

    m = build_ref.load()
    bs, ctx, ps, HQ, HKV, D = 8, 777, 16, 32, 8, 128
    g = torch.Generator().manual_seed(1)
    pages = bs * ((ctx + ps - 1) // ps)
    kb = torch.randn((pages + 1) * ps, HKV, D, generator=g).to(torch.bfloat16)
    vb = torch.randn((pages + 1) * ps, HKV, D, generator=g).to(torch.bfloat16)
    q = torch.randn(bs, HQ, D, generator=g).to(torch.bfloat16)
    perm = np.random.default_rng(0).permutation(np.arange(1, pages + 1))
    slots = (perm.reshape(bs, -1)[:, :, None] * ps + np.arange(ps)[None, None, :]).reshape(bs, -1)[:, :ctx]
    r2t = torch.zeros(bs + 1, ctx + ps, dtype=torch.int32)
    r2t[1:, :ctx] = torch.from_numpy(slots.astype(np.int32))
    rpi = torch.arange(1, bs + 1, dtype=torch.int64)
    lens = torch.full((bs,), ctx, dtype=torch.int64)
    out = torch.zeros(bs, HQ, D, dtype=torch.bfloat16)
    loc = r2t[1:, ctx - 1].to(torch.int64)
    # the reference kernel also writes the new token's K/V at loc: hand it the rows that are already there
    k_new, v_new = kb[loc].clone(), vb[loc].clone()
    m.decode_attention_cpu(q, kb, vb, out, k_new, v_new, loc, torch.zeros(bs, HQ, 8, D + 1), r2t, rpi, lens,
                           D ** -0.5, 0.0, False, 0, None, None)
    bits = lambda t: t.contiguous().view(torch.uint16).numpy()  # noqa: E731
    port = c_oracle.decode_bf16(bits(q), bits(kb), bits(vb), r2t.numpy(), rpi.numpy(), lens.numpy(), D ** -0.5)
    port = torch.from_numpy(np.ascontiguousarray(port)).view(torch.bfloat16).float().numpy().astype(np.float64)
    return float(np.abs(out.float().numpy().astype(np.float64) - port.reshape(bs, HQ, D)).max())


def main():
    res = {"where": "build container (no GPU)", "logical_cpus": os.cpu_count(),
           "shape": "one layer of the bench shape: bs=256, ctx=4096, Hq=32, Hkv=8, D=128, bf16, page 16 shuffled",
           "reference": _worker("reference"), "port": _worker("port"),
           "parity_max_abs_reference_vs_port_small_case": _parity(),
           "extend_shape": "one layer of a config-3 chunk: requests x (3584 shared cached + 512 new), Hq=32, Hkv=8, "
                           "D=128, bf16, page 16 shuffled (bench.py --cpu-worker reference-extend / port-extend)",
           "extend_reference": _worker("reference-extend", extra=("--cpu-chunk", "8")),
           "extend_port": _worker("port-extend", 8, extra=("--cpu-chunk", "1")),
           "extend_parity_max_abs_reference_vs_port_small_case": _parity_extend()}
    res["port_over_reference_time"] = res["port"]["ms_per_layer"] / res["reference"]["ms_per_layer"]
    path = os.path.join(ROOT, "tests", "golden", "cpu_baseline_container.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
