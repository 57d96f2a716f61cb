"""Dev sweep: MLA decode (config-5 shard shape, 16 q heads, latent 576 / 512) whole-op time vs forced kv-split count over
batch sizes; graph-replayed.  env: FP8=1 (fp8 rows), PS (page size, 64), SHAPES (bsxctx,...), SPLITS.
The last column is what the backend's native policy (rx_num_kv_splits_balanced, wg_target 512, min_tokens 128) picks."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sglang_amd import ops

dev = "cuda"
hq, dk, dv = 16, 576, 512
ps = int(os.environ.get("PS", "64"))
fp8 = bool(os.environ.get("FP8"))
shapes = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("SHAPES", "16x8192,32x8192,64x8192,128x8192,256x4096,512x2048").split(",")]
splits = [int(x) for x in os.environ.get("SPLITS", "1,2,4,8,16").split(",")]
for bs, ctx in shapes:
    g = torch.Generator(device=dev).manual_seed(3)
    perm = torch.randperm(bs * ctx // ps, device=dev, generator=g) + 1
    slots = (perm.view(bs, -1, 1) * ps + torch.arange(ps, device=dev)).view(bs, -1)[:, :ctx]
    r2t = torch.zeros(bs + 1, ctx, dtype=torch.int32, device=dev)
    r2t[1:] = slots.int()
    rpi = torch.arange(1, bs + 1, device=dev)
    lens = torch.full((bs,), ctx, dtype=torch.int64, device=dev)
    kv = torch.empty(bs * ctx + ps, 1, dk, dtype=torch.bfloat16, device=dev).normal_(generator=g)
    if fp8:
        kv = kv.to(torch.float8_e4m3fn)
    q = torch.randn(bs, hq, dk, device=dev, generator=g).to(torch.bfloat16)
    o = torch.empty(bs, hq, dv, dtype=torch.bfloat16, device=dev)
    row = []
    for S in splits:
        ns = torch.full((bs,), S, dtype=torch.int32, device=dev)
        S8 = max(8, S) if S > 1 else 1
        al = torch.empty(bs, hq, S8, dv, dtype=torch.float32, device=dev)
        lse = torch.empty(bs, hq, S8, dtype=torch.float32, device=dev)

        def run():
            if S == 1:
                ops.decode_attention_fwd_paged(q, kv, kv[..., :dv], o, r2t, rpi, lens, None, None, None, 1, dk ** -0.5, page_size=ps)
            else:
                ops.decode_attention_fwd_paged(q, kv, kv[..., :dv], o, r2t, rpi, lens, al, lse, ns, S8, dk ** -0.5, page_size=ps)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): run()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10): run()
        for _ in range(5): gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): gr.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        byt = bs * ctx * dk * (1 if fp8 else 2)
        row.append(f"S={S}: {us:.1f} us ({byt / us / 1e6 / 8:.3f})")
    pol = ops.balanced_kv_splits_host([ctx] * bs, hq, 1, 32, 512, 128, 0)
    print(f"bs={bs} ctx={ctx} {'fp8' if fp8 else 'bf16'}: " + " | ".join(row) + f" | native policy -> {int(pol.max())}", flush=True)
    del kv
