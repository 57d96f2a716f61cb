"""MLA fp8 decode (bench shape) over split counts, stage-2 launch vs in-kernel merge, 10 calls per HIP graph."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops, lib as L
dev = torch.device("cuda:0")
bs, ctx, hq, dk, dv, ps = 64, 8192, 16, 576, 512, 64
g = torch.Generator(device=dev).manual_seed(3)
pool = bs * ctx + ps
perm = torch.randperm(bs * ctx // ps, device=dev, generator=g) + 1
slots = (perm.view(bs, -1, 1) * ps + torch.arange(ps, device=dev)).view(bs, -1)[:, :ctx]
r2t = torch.zeros(bs + 1, ctx, dtype=torch.int32, device=dev); r2t[1:] = slots.int()
rpi = torch.arange(1, bs + 1, device=dev)
lens = torch.full((bs,), ctx, dtype=torch.int64, device=dev)
q = torch.randn(bs, hq, dk, device=dev, generator=g).to(torch.bfloat16)
o = torch.empty(bs, hq, dv, dtype=torch.bfloat16, device=dev)
kv = torch.empty(pool, 1, dk, dtype=torch.bfloat16, device=dev).normal_(generator=g).to(torch.float8_e4m3fn)
byt = bs * ctx * dk + bs * hq * (dk + dv) * 2
for S in (2, 4, 8, 16):
    for mc in (False, True):
        L.set_option("merge_in_kernel_max_mb_mla", 64 if mc else 0)
        nsplit = torch.full((bs,), S, dtype=torch.int32, device=dev)
        al = torch.empty(bs, hq, S, dv, dtype=torch.float32, device=dev)
        lse = torch.empty(bs, hq, S, dtype=torch.float32, device=dev)
        cnt = torch.zeros(bs * hq, dtype=torch.int32, device=dev)
        def run():
            ops.decode_attention_fwd_paged(q, kv, kv[..., :dv], o, r2t, rpi, lens, al, lse, nsplit, S, dk ** -0.5, page_size=ps,
                                           merge_counters=cnt if mc else None)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): run()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10): run()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        print(f"S={S:2d} in_kernel_merge={int(mc)} {us:6.1f} us  op frac {byt / us / 1e6 / 8000 * 1e3 / 1e3:.3f}  {L.last_dispatch()}", flush=True)
        del gr
