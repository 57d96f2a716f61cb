"""Speculative-verify shaped extend: bs requests with their own P-token cached prefix and ND draft tokens under
a (lower-triangular) tree mask; per-q-head launch vs GQA-packed rows.  env: BS (64) P (4096) ND (8) HQ (32) HKV (8)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import ops  # noqa: E402

E = lambda k, d: int(os.environ.get(k, d))  # noqa: E731
bs, P, nd, hq, hkv, d = E("BS", 64), E("P", 4096), E("ND", 8), E("HQ", 32), E("HKV", 8), 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
pool = bs * P + 16
kb = torch.randn(pool, hkv, d, device=dev, generator=g).to(torch.bfloat16)
vb = torch.randn(pool, hkv, d, device=dev, generator=g).to(torch.bfloat16)
T = bs * nd
q = torch.randn(T, hq, d, device=dev, generator=g).to(torch.bfloat16)
ke = torch.randn(T, hkv, d, device=dev, generator=g).to(torch.bfloat16)
ve = torch.randn(T, hkv, d, device=dev, generator=g).to(torch.bfloat16)
kv_indices = (torch.randperm(bs * P, device=dev, generator=g) + 8).to(torch.int64)
kv_indptr = (torch.arange(bs + 1, device=dev) * P).to(torch.int32)
qo = (torch.arange(bs + 1, device=dev) * nd).to(torch.int64)
rng = np.random.default_rng(0)
rows = []
for i in range(bs):
    m = np.ones((nd, P + nd), dtype=np.uint8)
    m[:, P:] = np.tril(rng.integers(0, 2, size=(nd, nd))) | np.eye(nd, dtype=np.int64)
    rows.append(m.reshape(-1))
mask = torch.from_numpy(np.concatenate(rows)).to(dev)
mi = torch.from_numpy(np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int64)).to(dev)
o1, o2 = torch.zeros_like(q), torch.zeros_like(q)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


args = (kb, vb, qo, kv_indptr, kv_indices, mask, True, mi, nd, 1.0, 1.0)
t1 = timed(lambda: ops.extend_attention_fwd(q, ke, ve, o1, *args))
t2 = timed(lambda: ops.extend_attention_fwd_gqa_packed(q, ke, ve, o2, *args))
vsk = ops.VerifySplitKV(hq, hkv, torch.bfloat16, dev)
S = vsk.num_chunks(bs)
o3 = torch.zeros_like(q)
vsk.plan(qo, kv_indptr, kv_indices, mask, mi, nd)
t3 = timed(lambda: vsk(q, ke, ve, o3, kb, vb, 1.0, 1.0))
t_plan = timed(lambda: vsk.plan(qo, kv_indptr, kv_indices, mask, mi, nd))
print(f"split-KV verify ({S} chunks): {t3:.0f} us ({t1 / t3:.2f}x the per-head launch); max |diff| vs packed {(o3.float() - o2.float()).abs().max().item():.4f}; plan {t_plan:.0f} us per forward")
# the same split with the causal rule instead of a tree mask: a short follow-up turn over a long conversation
vsk.plan(qo, kv_indptr, kv_indices, None, None, nd)
o4, o5 = torch.zeros_like(q), torch.zeros_like(q)
argc = (kb, vb, qo, kv_indptr, kv_indices, None, True, None, nd, 1.0, 1.0)
t4 = timed(lambda: ops.extend_attention_fwd_gqa_packed(q, ke, ve, o4, *argc))
t5 = timed(lambda: vsk(q, ke, ve, o5, kb, vb, 1.0, 1.0))
print(f"causal extend of {nd} tokens: packed {t4:.0f} us, split-KV {t5:.0f} us ({t4 / t5:.2f}x); max |diff| {(o4.float() - o5.float()).abs().max().item():.4f}")
byt = bs * P * hkv * d * 2 * 2
print(f"bs={bs} P={P} nd={nd}: per-head {t1:.0f} us, GQA-packed {t2:.0f} us ({t1 / t2:.2f}x); KV bytes once = "
      f"{byt / 1e6:.0f} MB -> {byt / t2 / 1e6:.2f} TB/s packed; max |diff| {(o1.float() - o2.float()).abs().max().item():.4f}")
