"""A/B of rx::extend_mfma64_kernel (one wave per SIMD, 64 rows per wave; tools/probe/rx_extend64.hip) against
rx::extend_mfma32_kernel's eight-wave PLAIN instances on the same inputs: max |o| / |lse| difference per shape and the two timings.
NEEDS A DEV BUILD of the library: RX_WITH_EXT64=1 RX_LIB_NAME=libradix_hip_ext64.so python -m sglang_amd.build (the product
library does not carry the kernel; option ext64 is a no-op there).  env SHAPES="bs x P + E , ..." (default a few), HQ / HKV (32 / 8), PS (page size, 16)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import lib as rxlib, ops  # noqa: E402

dev = "cuda"
hq, hkv, d = int(os.environ.get("HQ", 32)), int(os.environ.get("HKV", 8)), 128
ps = int(os.environ.get("PS", 16))
shapes = os.environ.get("SHAPES", "4x3584+512,3x1000+700,2x0+2048,5x64+300,32x3584+512")
dt = torch.float16 if os.environ.get("FP16") else torch.bfloat16
for sh in shapes.split(","):
    bs, rest = sh.split("x"); P, E = (int(x) for x in rest.split("+")); bs = int(bs)
    g = torch.Generator(device=dev).manual_seed(bs * 7 + P + E)
    npg = bs * (-(-max(P, 1) // ps)) + 2
    kb = torch.randn(npg * ps, hkv, d, device=dev, generator=g).to(dt)
    vb = torch.randn(npg * ps, hkv, d, device=dev, generator=g).to(dt)
    perm = torch.randperm(npg - 1, device=dev, generator=g) + 1
    kvi = []
    for i in range(bs):
        pg = perm[i * (-(-max(P, 1) // ps)): (i + 1) * (-(-max(P, 1) // ps))]
        kvi.append((pg[:, None] * ps + torch.arange(ps, device=dev)[None]).reshape(-1)[:P])
    kvi = torch.cat(kvi).to(torch.int64) if P else torch.zeros(0, dtype=torch.int64, device=dev)
    kvp = (torch.arange(bs + 1, device=dev) * P).to(torch.int32)
    T = bs * E
    q = torch.randn(T, hq, d, device=dev, generator=g).to(dt)
    ke = torch.randn(T, hkv, d, device=dev, generator=g).to(dt)
    ve = torch.randn(T, hkv, d, device=dev, generator=g).to(dt)
    qo = (torch.arange(bs + 1, device=dev) * E).to(torch.int64)
    res = {}
    MODE = int(os.environ.get("EXT64", "1"))
    for mode in (int(os.environ.get("EXT64", "1")), 0):
        with rxlib.option("ext64", mode), rxlib.option("ext32_small_wg", 0):
            o = torch.zeros(T, hq, d, device=dev, dtype=dt)
            lse = torch.zeros(T, hq, dtype=torch.float32, device=dev)
            run = lambda: ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, kvi, None, True, None, E, 1.0, 1.0,  # noqa: E731
                                                   sm_scale=d ** -0.5, lse_extend=lse, page_size=ps, avg_kv_len_hint=P + 2048)
            run(); torch.cuda.synchronize()
            name = rxlib.last_dispatch()
            for _ in range(10): run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): run()
            b.record(); torch.cuda.synchronize()
            res[mode] = (o.float(), lse.clone(), a.elapsed_time(b) / 20, name)
    fl = 2.0 * hq * 2 * d * bs * (E * P + E * (E + 1) / 2)
    do = (res[MODE][0] - res[0][0]).abs().max().item(); dl = (res[MODE][1] - res[0][1]).abs().max().item()
    if do > 1e-3:
        bad = ((res[MODE][0] - res[0][0]).abs() > 1e-3).any(dim=2).nonzero()
        toks = sorted(set(int(x) % E for x in bad[:, 0].tolist()))
        print("   mismatching (token-in-request) positions:", toks[:12], "...", toks[-6:], "count", len(toks),
              "heads", sorted(set(bad[:, 1].tolist()))[:12], "requests", sorted(set(int(x) // E for x in bad[:, 0].tolist())))
    print(f"{sh}: max|do| {do:.3e} max|dlse| {dl:.3e} nan {int(torch.isnan(res[MODE][0]).any())} | "
          f"{res[MODE][3].split('<')[0]} {res[MODE][2]*1e3:.0f} us {fl/res[MODE][2]/1e9:.0f} TF | {res[0][3].split('<')[0]} {res[0][2]*1e3:.0f} us {fl/res[0][2]/1e9:.0f} TF")
