import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sglang_amd import ops
dev = torch.device("cuda:0")
HQ, HKV, D, P, E, chunk, ps = 32, 8, 128, 3584, 512, 32, 16
g = torch.Generator(device=dev).manual_seed(1)
for big in (False, True):
    n_pages = (P + ps - 1) // ps + (chunk * ((E + ps - 1) // ps) if big else 0) + 1
    kb = torch.randn((n_pages, HKV, ps, D), device=dev, generator=g).to(torch.bfloat16)
    vb = torch.randn((n_pages, HKV, ps, D), device=dev, generator=g).to(torch.bfloat16)
    lay = ops.kv_layout_hnd(kb, vb)
    T = chunk * E
    q = torch.randn(T, HQ, D, device=dev, generator=g).to(torch.bfloat16)
    ke = torch.randn(T, HKV, D, device=dev, generator=g).to(torch.bfloat16)
    ve = torch.randn(T, HKV, D, device=dev, generator=g).to(torch.bfloat16)
    o = torch.empty(T, HQ, D, device=dev, dtype=torch.bfloat16)
    pages = torch.randperm(n_pages - 1, device=dev, generator=g)[: (P + ps - 1) // ps] + 1
    slots = (pages[:, None] * ps + torch.arange(ps, device=dev)[None, :]).reshape(-1)[:P].to(torch.int64)
    kvi = slots.repeat(chunk)
    kvp = (torch.arange(chunk + 1, device=dev) * P).to(torch.int32)
    qo = (torch.arange(chunk + 1, device=dev) * E).to(torch.int64)
    def run():
        ops.extend_attention_fwd(q, ke, ve, o, kb, vb, qo, kvp, kvi, None, True, None, E, 1.0, 1.0, sm_scale=D ** -0.5, page_size=ps, kv_layout=lay)
    torch.cuda.synchronize(); time.sleep(3)
    ts = []
    for i in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("big pool" if big else "small pool", "per-launch ms from idle:", " ".join(f"{t:.3f}" for t in ts[:12]), "... last:", f"{ts[-1]:.3f}")
    # back-to-back timing like bench.py
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): run()
    e1.record(); torch.cuda.synchronize()
    print("   8 back-to-back:", e0.elapsed_time(e1) / 8)
