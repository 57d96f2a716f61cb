"""Dev: GB/s of the byte / index kernels at prefill-sized inputs (Llama-3-8B geometry: Hkv 8, D 128, bf16).
Graph-replayed (20 calls per graph); bytes = the algorithmic read + write of each kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops

dev = "cuda"


def timeit(run):
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
    return e0.elapsed_time(e1) / 100 * 1e3


def rep(name, us, byt):
    print(f"{name}: {us:.1f} us  {byt / us / 1e3:.0f} GB/s ({byt / us / 1e3 / 8000:.2f} of 8 TB/s)", flush=True)


hkv, hq, d, n, page = 8, 32, 128, 16384, 16
k = torch.randn(n, hkv, d, device=dev).to(torch.bfloat16); v = torch.randn_like(k)
slots = n + page
loc = torch.randperm(n, device=dev) + 1
kb = torch.zeros(slots, hkv, d, dtype=torch.bfloat16, device=dev); vb = torch.zeros_like(kb)
rep("store_cache NHD 16Ki tok", timeit(lambda: ops.store_cache(k.view(n, -1), v.view(n, -1), kb.view(slots, -1), vb.view(slots, -1), loc)), n * hkv * d * 2 * 4)
kh = torch.zeros(slots // page, hkv, page, d, dtype=torch.bfloat16, device=dev); vh = torch.zeros_like(kh)
lay = ops.kv_layout_hnd(kh, vh)
rep("store_cache_layout HND", timeit(lambda: ops.store_cache_layout(k, v, lay, loc, hkv, d, d, size_limit=slots)), n * hkv * d * 2 * 4)
k8 = torch.zeros(slots, hkv, d, dtype=torch.float8_e4m3fn, device=dev); v8 = torch.zeros_like(k8)
lay8 = ops._kv_layout(k8, v8, 1)
rep("store_cache_fp8", timeit(lambda: ops.store_cache_fp8(k, v, lay8, loc, hkv, d, d, size_limit=slots, k_scale=0.5, v_scale=2.0)), n * hkv * d * (2 + 1) * 2)
a = torch.randn(n, hq, d, device=dev).to(torch.bfloat16); b = torch.randn_like(a)
la = torch.randn(n, hq, device=dev); lb = torch.randn(n, hq, device=dev)
o = torch.empty_like(a); lo = torch.empty_like(la)
rep("merge_state 16Ki x 32 x 128", timeit(lambda: ops.merge_state(a, la, b, lb, o, lo)), n * hq * d * 2 * 3 + n * hq * 4 * 3)
bs, ctx = 256, 4096
r2t = torch.randint(1, 1 << 20, (bs + 1, ctx), dtype=torch.int32, device=dev)
rpi = torch.arange(1, bs + 1, device=dev); lens = torch.full((bs,), ctx, dtype=torch.int64, device=dev)
kvp = torch.zeros(bs + 1, dtype=torch.int32, device=dev); kvi = torch.empty(bs * ctx, dtype=torch.int64, device=dev)
rep("build_kv_indices 256 x 4096", timeit(lambda: ops.build_kv_indices(r2t, rpi, lens, kvp, kvi)), bs * ctx * 12)
lat = torch.randn(n, 1, 576, device=dev).to(torch.float8_e4m3fn)
rep("get_mla_kv fp8 -> bf16, 16Ki rows", timeit(lambda: ops.get_mla_kv(lat, loc - 1, 512, 64, torch.bfloat16, size_limit=n)), n * 576 * 3)
ptrs = torch.tensor([kb.data_ptr(), vb.data_ptr()], dtype=torch.int64, device=dev).view(torch.uint64) if hasattr(torch, "uint64") else None
rb = torch.tensor([hkv * d * 2, hkv * d * 2], dtype=torch.int64, device=dev)
src = torch.randperm(n // 2, device=dev) + 1; tgt = src + n // 2
rep("move_kv 8Ki rows x 2 buffers", timeit(lambda: ops.move_kv(ptrs, rb, tgt, src)), (n // 2) * hkv * d * 2 * 2 * 2)
