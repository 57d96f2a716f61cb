"""Decode context parallel (SURVEY 8e "alternative shardings"): the index / store / LSE kernels of csrc/rx_dcp.hip
against the reference-generated golden F15 and the oracle, and the whole path -- prefill, a second prefill chunk over
the sharded prefix, decode steps -- with TWO processes on the one GPU of the test box (gloo carries the exchanges
through the host), every rank checking its heads against the oracle's attention over the WHOLE sequence."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import parity_util as parity
from oracle import radix_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ops():
    from sglang_amd import ops as o
    return o


def _cases():
    npz = np.load(os.path.join(GOLD, "dcp.npz"))
    cases = {}
    for key in npz.files:
        case, field = key.split(".", 1)
        cases.setdefault(case, {})[field] = npz[key]
    return cases


@pytest.mark.parametrize("out_dtype", [torch.int64, torch.int32])
def test_dcp_kv_indices_golden_bit_exact(ops, out_dtype):
    """rx_dcp_kv_indices vs the reference's get_dcp_lens + create_triton_kv_indices_for_dcp_triton (F15)."""
    cases = _cases()
    for i in range(int(cases["idx"]["count"])):
        c = cases[f"idx{i}"]
        bs = len(c["lens"])
        r2t = torch.from_numpy(c["req_to_token"]).to(DEV)
        start = torch.from_numpy(c["start"]).to(DEV) if int(c["use_start"]) else None
        for lens_dt, rpi_dt in ((torch.int32, torch.int32), (torch.int64, torch.int64)):
            kv_indptr = torch.full((bs + 1,), -7, dtype=torch.int32, device=DEV)
            kv_indices = torch.full((len(c["kv_indices"]) + 3,), -1, dtype=out_dtype, device=DEV)
            dl = torch.full((bs,), -1, dtype=torch.int32, device=DEV)
            ops.dcp_kv_indices(r2t, torch.from_numpy(c["req_pool_indices"]).to(rpi_dt).to(DEV),
                               torch.from_numpy(c["lens"]).to(lens_dt).to(DEV), kv_indptr, kv_indices, int(c["dcp"]),
                               int(c["rank"]), kv_start=start, dcp_lens=dl)
            assert np.array_equal(kv_indptr.cpu().numpy(), c["kv_indptr"]), i
            assert np.array_equal(dl.cpu().numpy(), c["dcp_lens"]), i
            n = len(c["kv_indices"])
            assert np.array_equal(kv_indices[:n].cpu().numpy().astype(np.int64), c["kv_indices"]), i
            assert (kv_indices[n:] == -1).all()   # nothing written past the rank's share


def test_dcp_store_loc_vs_oracle(ops):
    rng = np.random.default_rng(3)
    for dcp in (2, 4):
        for rank in range(dcp):
            loc = rng.integers(1, 1 << 20, size=777)
            pos = rng.integers(0, 5000, size=777)
            for dt in (torch.int64, torch.int32):
                got = ops.dcp_store_loc(torch.from_numpy(loc).to(dt).to(DEV), torch.from_numpy(pos).to(dt).to(DEV), dcp,
                                        rank, skip_index=0)
                assert np.array_equal(got.cpu().numpy(), orc.dcp_store_loc(loc, pos, dcp, rank, 0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_dcp_merge_kernels_vs_reference_golden(ops, dtype):
    """rx_dcp_scale + (host sum) + rx_dcp_finish vs cp_lse_ag_out_rs_mha's recorded vectors (F15): scaled contribution
    of every rank, NaN rows of token-less ranks, rows empty everywhere, the head slice each rank keeps."""
    cases = _cases()
    for mi in range(int(cases["merge"]["count"])):
        c = cases[f"merge{mi}"]
        world, T, H, D = c["outs"].shape
        lses = torch.from_numpy(c["lses"]).to(DEV).contiguous()
        scaled = []
        for r in range(world):
            o32 = torch.from_numpy(c["outs"][r]).to(DEV).contiguous()
            g = torch.empty(T, H, dtype=torch.float32, device=DEV)
            ops.dcp_scale(o32, lses, r, g)
            assert np.abs(o32.cpu().numpy() - c["scaled"][r]).max() < 2e-6
            scaled.append(o32)
        summed = torch.stack(scaled).sum(0).contiguous()
        hl = H // world
        for r in range(world):
            out = torch.empty(T, hl, D, dtype=dtype, device=DEV)
            ops.dcp_finish(summed, out, r * hl)
            want = torch.from_numpy(c["final"][r]).to(dtype)          # one rounding of the reference's fp32 result
            assert torch.equal(out.cpu(), want) or (out.cpu().float() - want.float()).abs().max() <= 2e-3
            gl = g.cpu().numpy()[:, r * hl:(r + 1) * hl]
            wl = c["global_lse"][r]
            with np.errstate(invalid="ignore"):
                assert np.all((np.isneginf(gl) & np.isneginf(wl)) | (np.abs(gl - wl) < 1e-5))


def test_dcp_local_merge_and_finish_with_current_chunk(ops):
    """kv-split partials -> (fp32 output, LSE) vs the oracle's merge; then the extend path's final join of the
    cross-rank prefix part with the own-chunk partial (triton_backend.py:1560-1569) vs orc.merge_state."""
    g = torch.Generator().manual_seed(2)
    bs, H, S, D = 5, 6, 4, 64
    logits = torch.randn(bs, H, S, D, generator=g)
    lse = torch.randn(bs, H, S, generator=g) * 2
    lse[0, :, 2:] = float("-inf")
    lse[1] = float("-inf")                              # a request without local tokens
    logits[1] = float("nan")
    o32, l = ops.dcp_local_merge(logits.to(DEV), lse.to(DEV), v_scale=0.5)
    w = torch.softmax(lse.double(), dim=-1)
    w = torch.nan_to_num(w, nan=0.0)
    want = (torch.nan_to_num(logits.double(), nan=0.0) * w[..., None]).sum(2) * 0.5
    assert (o32.cpu().double() - want).abs().max() < 1e-5
    wl = torch.logsumexp(lse.double(), dim=-1)
    assert torch.all((l.cpu().double() - wl).abs().nan_to_num(0.0) < 1e-5) and torch.isneginf(l[1]).all()
    # final join
    T, Hall, hl, h0 = 7, 8, 4, 4
    pre = torch.randn(T, Hall, D, generator=g)
    pre_l = torch.randn(T, Hall, generator=g)
    pre_l[0] = float("-inf")
    cur = torch.randn(T, hl, D, generator=g).to(torch.bfloat16)
    cur_l = torch.randn(T, hl, generator=g)
    cur_l[1] = float("-inf")
    out = torch.empty(T, hl, D, dtype=torch.bfloat16, device=DEV)
    ops.dcp_finish(pre.to(DEV), out, h0, pre_l.to(DEV), cur.to(DEV), cur_l.to(DEV))
    want, _ = orc.merge_state(pre[:, h0:h0 + hl].numpy(), pre_l[:, h0:h0 + hl].numpy(),
                              cur.float().numpy(), cur_l.numpy())
    fin = np.isfinite(want)
    parity.check_out(out.float().cpu().numpy()[fin], want[fin], torch.bfloat16, "dcp_finish", ulps=1)  # one rounding of an exact blend


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["RX_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RX_ROOT"], "tests"))
from oracle import radix_oracle as orc
import parity_util as parity
from sglang_amd import ops
from sglang_amd.attention.backend import HipRadixAttnBackend
from sglang_amd.attention.dcp import DcpGroup
from sglang_amd.attention.radix_attention import RadixAttention
from sglang_amd.forward_batch import ForwardBatch
from sglang_amd.mem_cache.allocator import PagedTokenToKVPoolAllocator
from sglang_amd.mem_cache.memory_pool import MHATokenToKVPool, ReqToTokenPool

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
DEV = "cuda"
dtype = torch.bfloat16 if os.environ["RX_DTYPE"] == "bf16" else torch.float16
# the ranks of a DCP group are the TP ranks that would otherwise REPLICATE one kv head: one kv head per rank, and the
# group's gathered q heads are that head's whole GQA group
H_ALL, HKV, D, PS = 8, 1, int(os.environ["RX_D"]), 16
MLA = D == 576                                 # the latent-row pool (one "kv head" of 576, V = its first 512 columns):
DV = 512 if MLA else D                         # replicated over TP in the reference, i.e. the case DCP exists for
HL = H_ALL // world
VSIZE = 4096                                   # virtual slots (page 16, pages aligned so that slot % dcp == position % dcp)
if MLA:
    from sglang_amd.mem_cache.memory_pool import MLATokenToKVPool
    pool = MLATokenToKVPool(VSIZE // world + PS, 1, dtype, 512, 64, 1, DEV)
else:
    pool = MHATokenToKVPool(VSIZE // world + PS, 1, dtype, HKV, D, 1, DEV)    # this rank's share
r2t = ReqToTokenPool(8, 1024, DEV)
alloc = PagedTokenToKVPoolAllocator(VSIZE, PS, dtype, DEV, pool)               # same decisions on every rank
g0 = torch.Generator().manual_seed(5)
alloc.free_pages = alloc.free_pages[torch.randperm(len(alloc.free_pages), generator=g0).to(DEV)]

class MC: num_attention_heads, num_key_value_heads, context_len = H_ALL, HKV, 1024
class MR:
    device = DEV; req_to_token_pool = r2t; token_to_kv_pool = pool; token_to_kv_pool_allocator = alloc
    model_config = MC; page_size = 1; tp_size = world
    class server_args: triton_attention_num_kv_splits = 4
backend = HipRadixAttnBackend(MR, dcp=DcpGroup(world, rank))
assert backend.num_head == H_ALL and backend.local_num_head == HL and backend.decode_index_mode == "indices"
layer = RadixAttention(HL, D, D ** -0.5, HKV, 0, v_head_dim=DV)
gen = torch.Generator().manual_seed(11)        # same stream on every rank
rows = [3, 1, 5]
kc = np.zeros((VSIZE + PS, HKV, D), dtype=np.float64)      # the virtual cache, for the oracle
vc = np.zeros((VSIZE + PS, HKV, DV), dtype=np.float64)
seq = [0, 0, 0]
ok = True

def rnd(*shape):
    return torch.randn(*shape, generator=gen).to(dtype)

def check(got, want, what, absw, ulps):
    global ok
    try:
        parity.check_out(got.float().cpu().numpy(), want, dtype, what, ulps=ulps, absw=absw)
    except AssertionError as e:
        ok = False
        print(f"rank {rank} {what}: {e}", flush=True)

def extend_step(ext):
    global seq
    pre = list(seq); new = [p + e for p, e in zip(pre, ext)]
    pre_t, new_t = torch.tensor(pre, dtype=torch.int64), torch.tensor(new, dtype=torch.int64)
    last = torch.tensor([int(r2t.req_to_token[r, p - 1]) if p > 0 else -1 for r, p in zip(rows, pre)], dtype=torch.int64, device=DEV)
    loc = alloc.alloc_extend(pre_t.to(DEV), pre_t, new_t.to(DEV), new_t, last, sum(ext))
    ops.write_req_to_token(r2t.req_to_token, torch.tensor(rows, dtype=torch.int64, device=DEV), None, pre_t.to(DEV),
                           new_t.to(DEV), (new_t - pre_t).to(DEV), loc)
    T = sum(ext)
    q, k = rnd(T, H_ALL, D), rnd(T, HKV, D)
    v = k[..., :DV] if MLA else rnd(T, HKV, D)
    kc[loc.cpu().numpy()] = k.double().numpy(); vc[loc.cpu().numpy()] = v.double().numpy()
    fb = ForwardBatch.for_extend(torch.tensor(rows, dtype=torch.int64, device=DEV), new_t.to(DEV), loc, pre, ext)
    # slot % dcp == position % dcp: what makes (virtual slot // dcp) collision-free on a rank
    assert torch.equal(loc % world, fb.positions % world)
    backend.init_forward_metadata(fb)
    ql = q[:, rank * HL:(rank + 1) * HL].contiguous().to(DEV)
    kd = k.to(DEV)
    out = layer(ql.view(T, -1), kd, kd[..., :DV] if MLA else v.to(DEV), fb, backend)
    args = (r2t.req_to_token.cpu().numpy(), np.array(rows), np.array(new), np.array(pre), np.array(ext), D ** -0.5)
    want = orc.sdpa_extend_req_to_token(q.double().numpy(), kc, vc, *args, causal=True)
    absw = orc.sdpa_extend_req_to_token(q.double().numpy(), kc, np.abs(vc), *args, causal=True)
    # two 16-bit roundings where a prefix exists: the extend kernels' partials are 16-bit (as merge_state's inputs are)
    check(out.view(T, HL, DV), want[:, rank * HL:(rank + 1) * HL], f"extend {pre}+{ext}",
          absw[:, rank * HL:(rank + 1) * HL], 2.0 if sum(pre) else 1.0)
    seq = new

def decode_step():
    global seq
    new = [s + 1 for s in seq]
    new_t = torch.tensor(new, dtype=torch.int64)
    last = torch.tensor([int(r2t.req_to_token[r, s - 1]) for r, s in zip(rows, seq)], dtype=torch.int64, device=DEV)
    loc = alloc.alloc_decode(new_t.to(DEV), new_t, last)
    for r, s, l in zip(rows, seq, loc.tolist()):
        r2t.req_to_token[r, s] = l
    bs = len(rows)
    q, k = rnd(bs, H_ALL, D), rnd(bs, HKV, D)
    v = k[..., :DV] if MLA else rnd(bs, HKV, D)
    kc[loc.cpu().numpy()] = k.double().numpy(); vc[loc.cpu().numpy()] = v.double().numpy()
    fb = ForwardBatch.for_decode(torch.tensor(rows, dtype=torch.int64, device=DEV), new_t.to(DEV), loc)
    backend.init_forward_metadata(fb)
    ql = q[:, rank * HL:(rank + 1) * HL].contiguous().to(DEV)
    kd = k.to(DEV)
    out = layer(ql.view(bs, -1), kd, kd[..., :DV] if MLA else v.to(DEV), fb, backend)
    args = (r2t.req_to_token.cpu().numpy(), np.array(rows), np.array(new), D ** -0.5)
    want = orc.sdpa_decode_req_to_token(q.double().numpy(), kc, vc, *args)
    absw = orc.sdpa_decode_req_to_token(q.double().numpy(), kc, np.abs(vc), *args)
    # fp32 partials end to end (as the reference): one output rounding
    check(out.view(bs, HL, DV), want[:, rank * HL:(rank + 1) * HL], f"decode {new}", absw[:, rank * HL:(rank + 1) * HL], 1.0)
    seq = new

extend_step([37, 64, 5])          # prefill: no prefix anywhere, no collective on the data path
extend_step([20, 3, 140])         # second chunk: prefix sharded over the ranks, cross-rank LSE join + own-chunk join
for _ in range(3):
    decode_step()
# every rank stored exactly its own tokens: positions p with p % world == rank, at slot // world
kb = pool.get_key_buffer(0).float().cpu().numpy()
r2 = r2t.req_to_token.cpu().numpy()
for r, s in zip(rows, seq):
    for p in range(s):
        v_slot = int(r2[r, p])
        mine = np.abs(kb[v_slot // world] - kc[v_slot]).max() < 1e-6
        if (p % world == rank) != bool(mine):
            other = [pp for pp in range(s) if pp % world == rank and int(r2[r, pp]) // world == v_slot // world]
            if not other:
                ok = False; print(f"rank {rank}: row {r} position {p} ownership wrong", flush=True)
assert pool.check_errors() == 0
dist.barrier()
dist.destroy_process_group()
print("RANK_OK" if ok else "RANK_FAIL", flush=True)
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("dt,d", [("bf16", 128), ("fp16", 64), ("bf16", 576)], ids=["bf16-128", "fp16-64", "mla-bf16"])
def test_dcp_prefill_and_decode_across_processes(world, dt, d, tmp_path):
    script = tmp_path / "dcp_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + world + (d == 64) + 20 * (d == 576)),
               WORLD_SIZE=str(world), RX_DTYPE=dt, RX_D=str(d), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\nTIMEOUT"
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RANK_OK" in out, f"rank {r}:\n{out[-3000:]}"
