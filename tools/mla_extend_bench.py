#!/usr/bin/env python3
"""Latent (absorbed MLA) extend, q 576 / v 512 over one kv head: TFLOP/s of rx::extend_mla_kernel.

    python tools/mla_extend_bench.py                 # bs 32 x (3584 cached + 512 new), 16 q heads (DeepSeek TP8 shard)
    SHAPES=32x3584+512,8x8192+2048 HQ=16,128 python tools/mla_extend_bench.py
    OWNV=1  ... the new tokens' v is its own tensor (the kernel's two-image form)
    GENERIC=1 ... also time the scalar generic kernel (RX_OPT_EXTEND_MLA=0 in a child process) on the first shape

FLOPs = 2 * (576 + 512) * Hq * sum_i (E_i * P_i + E_i (E_i + 1) / 2)   (causal)."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
DK, DV = 576, 512


def run(bs, P, E, hq, own_v, iters):
    g = torch.Generator(device=DEV).manual_seed(1)
    pool = bs * (P + E) + 64
    latent = (torch.randn(pool, 1, DK, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    perm = torch.randperm(pool - 1, device=DEV, generator=g)[: bs * (P + E)] + 1
    if os.environ.get("CONTIG"):
        perm = torch.arange(1, bs * (P + E) + 1, device=DEV)
    perm = perm.view(bs, P + E)
    kv_indices = perm[:, :P].reshape(-1).contiguous()
    ext_slots = perm[:, P:].reshape(-1)
    kv_indptr = (torch.arange(bs + 1, device=DEV, dtype=torch.int32) * P).contiguous()
    qo = (torch.arange(bs + 1, device=DEV, dtype=torch.int64) * E).contiguous()
    q = torch.randn(bs * E, hq, DK, device=DEV, generator=g).to(torch.bfloat16)
    ke = latent[ext_slots].contiguous()
    ve = ke[..., :DV].contiguous() if own_v else ke[..., :DV]
    o = torch.empty(bs * E, hq, DV, device=DEV, dtype=torch.bfloat16)
    lse = None
    sm = 1.0 / (192 ** 0.5)

    def call():
        ops.extend_attention_fwd(q, ke, ve, o, latent, latent[..., :DV], qo, kv_indptr, kv_indices, None, True, None, E,
                                 1.0, 1.0, sm_scale=sm, lse_extend=lse)

    if os.environ.get("SPLIT"):  # the split-KV form (ops.VerifySplitKV, causal rule) for small batches of short extends
        vs = ops.VerifySplitKV(hq, 1, torch.bfloat16, DEV, head_dim=DK, v_head_dim=DV)
        vs.plan(qo, kv_indptr, kv_indices, None, None, E)
        print("chunks per request:", vs.num_chunks(bs, E))

        def call():  # noqa: F811
            vs(q, ke, ve, o, latent, latent[..., :DV], 1.0, 1.0, sm_scale=sm)

    call()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        call()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / iters
    flops = 2.0 * (DK + DV) * hq * bs * (E * P + E * (E + 1) / 2)
    return ms, flops / ms / 1e9, o


def main():
    shapes = os.environ.get("SHAPES", "32x3584+512").split(",")
    hqs = [int(x) for x in os.environ.get("HQ", "16").split(",")]
    own_v = bool(int(os.environ.get("OWNV", "0")))
    iters = int(os.environ.get("ITERS", "5"))
    for sh in shapes:
        bs, rest = sh.split("x")
        P, E = rest.split("+")
        for hq in hqs:
            ms, tf, _ = run(int(bs), int(P), int(E), hq, own_v, iters)
            print(json.dumps({"bs": int(bs), "prefix": int(P), "extend": int(E), "hq": hq, "own_v": own_v,
                              "kernel": "generic" if (os.environ.get("RX_OPT_EXTEND_MLA") == "0") else "extend_mla",
                              "ms": round(ms, 3), "tflops": round(tf, 1), "frac_of_2.5PF": round(tf / 2500, 3)}), flush=True)
    if os.environ.get("GENERIC") and not (os.environ.get("RX_OPT_EXTEND_MLA") == "0"):
        env = dict(os.environ, RX_OPT_EXTEND_MLA="0", SHAPES=shapes[0], HQ=str(hqs[0]), ITERS="1")
        env.pop("GENERIC")
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)


if __name__ == "__main__":
    main()
