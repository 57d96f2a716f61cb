"""Dev: time ops.fused_qk_norm_rope (Qwen3-8B heads 32 / 8 / 8, D 128, bf16) at prefill- and decode-sized token counts,
with and without the KV store in the same launch; GB/s of algorithmic bytes (q, k read + written; with the store also
v read and the k / v rows written to the pool).  Graph-replayed, 20 calls per graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops

dev = "cuda"
hq, hkv, d = 32, 8, 128
for n in (256, 4096, 16384, 65536):
    qkv = torch.randn(n, (hq + 2 * hkv) * d, device=dev).to(torch.bfloat16)
    qw = torch.randn(d, device=dev).to(torch.bfloat16); kw = torch.randn(d, device=dev).to(torch.bfloat16)
    pos = torch.arange(n, dtype=torch.int32, device=dev)
    kb = torch.zeros(n + 16, hkv, d, dtype=torch.bfloat16, device=dev); vb = torch.zeros_like(kb)
    lay = ops._kv_layout(kb, vb, 1)
    loc = torch.randperm(n, device=dev) + 1
    for store in (False, True):
        kwargs = dict(layout=lay, loc=loc, size_limit=n + 16) if store else {}
        run = lambda: ops.fused_qk_norm_rope(qkv, hq, hkv, hkv, d, 1e-6, qw, kw, 10000.0, True, pos, **kwargs)  # noqa: E731
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
