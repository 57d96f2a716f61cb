"""Dev: time ops.rope_store_kv (Llama-3-8B heads 32 / 8, D 128, bf16; cos_sin_cache) with and without the pool store."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops

dev = "cuda"
hq, hkv, d = 32, 8, 128
inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float, device=dev) / d))
fr = torch.einsum("i,j->ij", torch.arange(8192, dtype=torch.float, device=dev), inv)
cache = torch.cat((fr.cos(), fr.sin()), dim=-1).contiguous()
for n in (256, 4096, 16384):
    q = torch.randn(n, hq, d, device=dev).to(torch.bfloat16); k = torch.randn(n, hkv, d, device=dev).to(torch.bfloat16)
    v = torch.randn(n, hkv, d, device=dev).to(torch.bfloat16)
    pos = (torch.arange(n, device=dev) % 8192).to(torch.int64)
    kb = torch.zeros(n + 16, hkv, d, dtype=torch.bfloat16, device=dev); vb = torch.zeros_like(kb)
    lay = ops._kv_layout(kb, vb, 1)
    loc = torch.randperm(n, device=dev) + 1
    for store in (False, True):
        kwargs = dict(layout=lay, loc=loc, size_limit=n + 16) if store else {}
        run = lambda: ops.rope_store_kv(q, k, v if store else None, pos, cache, True, **kwargs)  # noqa: E731
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): run()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): run()
        for _ in range(3): gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        byt = n * (hq + hkv) * d * 2 * 2 + (n * hkv * d * 2 * 3 if store else 0)
        print(f"n={n} store={store}: {us:.1f} us  {byt / us / 1e3:.0f} GB/s ({byt / us / 1e3 / 8000:.2f} of 8 TB/s)", flush=True)
