"""Dev: library-GEMM variants for the row-parallel o_proj shapes [256, K] x [K, 4096], K = 4096 / TP."""
import torch, torch.nn.functional as F
dev = "cuda"
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for K in (4096, 2048, 1024, 512):
    x = torch.randn(256, K, device=dev, dtype=torch.bfloat16)
    W = (torch.randn(K, 4096, device=dev) * 0.02).to(torch.bfloat16)
    Wt = W.t().contiguous()
    xt = x.t().contiguous()
    out = torch.empty(256, 4096, device=dev, dtype=torch.bfloat16)
    r = {"mm(x,W)": t(lambda: torch.mm(x, W)), "mm(x,Wt.t())": t(lambda: torch.mm(x, Wt.t())),
         "linear(x,Wt)": t(lambda: F.linear(x, Wt)), "mm(Wt,x.t()) [y^T]": t(lambda: torch.mm(Wt, x.t())),
         "mm(W.t(),xt)": t(lambda: torch.mm(W.t(), xt)), "empty kernel": t(lambda: out.zero_())}
    print(K, {k: round(v, 1) for k, v in r.items()})
