"""C3: the quick all-reduce (csrc/rx_quick_allreduce.hip; the reference's QuickAllReduce, srt/distributed/
device_communicators/quick_all_reduce.py + kernels/aot/csrc/allreduce/quick_all_reduce.cuh) across PROCESSES through
IPC-mapped regions: every level (FP / INT8 / INT6 / INT4), fp16 / bf16-as-fp16 / native bf16, world 2 / 4 / 8, eagerly, in
place, and under HIP-graph replay -- bit for bit against oracle/radix_oracle.py quick_allreduce fed with the v_rcp_f16 table
read back from THIS GPU, plus the properties the reference's own test asserts (test/manual/test_quick_allreduce.py).
The gpurun box has one GPU: the ranks share cuda:0 (handle exchange, slot / flag protocol, tile walking and arithmetic are
exercised; xGMI coherence is not), and the grid is capped (option qr_max_blocks) so that eight co-resident kernels fit one
device and every workgroup walks several tiles."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, ctypes as C
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["RX_ROOT"])
from sglang_amd import lib as L
from sglang_amd.parallel import QuickAllReduce, TPGroup
from oracle import radix_oracle as O
sys.path.insert(0, os.path.join(os.environ["RX_ROOT"], "tools"))
from dump_rcp_f16 import read_table

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
L.set_option("qr_max_blocks", 4)
table = read_table()
golden = os.path.join(os.environ["RX_ROOT"], "tests", "golden", "rcp_f16_gfx950.npy")
ok = True
if os.path.exists(golden) and not np.array_equal(np.load(golden), table):
    ok = False
    print(f"rank {rank}: the committed v_rcp_f16 table differs from this GPU's", flush=True)

def bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1)

def parts_for(seed, n, dt, kind):
    g = torch.Generator().manual_seed(seed)
    out = []
    for r in range(world):
        if kind == "ints":      # the reference's test data: integers in [1, 23)
            x = torch.randint(1, 23, (n,), generator=g).float()
        else:                   # activations with structure: a zero run, a tiny run (fp16 subnormals), a large run, signs
            x = torch.randn(n, generator=g)
            if n >= 512:
                x[64:192] = 0.0
                x[192:256] *= 3e-7
                x[256:320] *= 2e3
                x[320:384] = x[320:384].abs()
                x[384:448] = -x[384:448].abs()
        out.append(x.to(dt))
    return out   # the same list on every rank

def check(qr, parts, dt, tag, in_place=False):
    global ok
    x = parts[rank].to(dev)
    got = qr.quick_all_reduce(x, out=x if in_place else None)
    torch.cuda.synchronize()
    want = O.quick_allreduce([bits(p) for p in parts], dt == torch.bfloat16, qr.qr_quant_level.value,
                             bool(qr.use_fp16_kernels), rcp_f16_table=table)
    g = bits(got)
    if not np.array_equal(g, want):
        bad = np.flatnonzero(g != want)
        only_zero_sign = bool(((g[bad] | want[bad]) & 0x7fff).max() == 0)
        ok = False
        print(f"rank {rank} {tag}: {len(bad)} of {len(g)} elements differ (only the sign of zeros: {only_zero_sign}); first at "
              f"{bad[:4]}: got {g[bad[:4]]} want {want[bad[:4]]}", flush=True)
    if not in_place and not torch.equal(x.cpu(), parts[rank]):
        ok = False
        print(f"rank {rank} {tag}: the input was modified", flush=True)
    return got

# every (level, dtype mode) at every world size = every kernel instance of the library (36); the small sizes, the integer data
# and the variable-input loop run at world 2 only
COMBOS = [(lv, dt, c) for lv in ("FP", "INT8", "INT6", "INT4") for dt, c in ((torch.float16, 0), (torch.bfloat16, 1), (torch.bfloat16, 0))]
SIZES = {2: [8, 64 * 3 + 8, 16384 * 9 + 72], 4: [16384 * 4 + 72], 8: [16384 * 4 + 8]}[world]   # (4 workgroups: the last sizes walk 2-3 tiles each)
import time
T0 = time.perf_counter()
def mark(what):
    if rank == 0:
        print(f"[t+{time.perf_counter() - T0:6.1f}s] {what}", flush=True)
seed = 0
# ONE context per process for the whole run, as in serving (GroupCoordinator builds qr_comm once): the level and the bf16 mode
# are per-call arguments of rx_quick_allreduce, which the host object keeps as attributes
from sglang_amd.parallel import QuickReduceRegime
qr = QuickAllReduce(None, dev, regime="FP")
assert not qr.disabled
mark("context up")
for level, dt, cast in COMBOS:
    qr.qr_quant_level, qr.use_fp16_kernels = QuickReduceRegime[level], int(cast)
    mark(f"{level} {dt} cast={cast}")
    for n in SIZES:
        for kind in (("acts", "ints") if world == 2 else ("acts",)):
            seed += 1
            parts = parts_for(seed, n, dt, kind)
            got = check(qr, parts, dt, f"{level} {dt} cast={cast} n={n} {kind}", in_place=(seed % 2 == 0))
            if kind == "ints":   # test_quick_allreduce.py:150-160: atol 1.25 W, rtol 0.5 W against the exact sum
                exact = sum(p.float() for p in parts)
                err = (got.float().cpu() - exact).abs()
                if not bool((err <= 1.25 * world + 0.5 * world * exact.abs()).all()):
                    ok = False
                    print(f"rank {rank} {level} {dt}: integer sums off by {err.max().item()}", flush=True)
                if level == "FP" and err.max().item() != 0:
                    ok = False
                    print(f"rank {rank} FP {dt}: integer sums are not exact", flush=True)
    if "quick_allreduce_kernel" not in L.last_dispatch():
        ok = False
        print(f"rank {rank}: dispatch record {L.last_dispatch()!r}", flush=True)
    # the reference's variable-input loop (test_quick_allreduce.py:213-240): zeros stay zeros, ones sum to `world`, at
    # alternating sizes through the same context
    mark("oracle comparisons done")
    for i in range(4 if world == 2 else 2 if dt == torch.float16 else 0):
        n = 16384 * (3 if i % 2 else 6)
        x = (torch.zeros if i % 2 == 0 else torch.ones)(n, dtype=dt, device=dev)
        y = qr.quick_all_reduce(x)
        torch.cuda.synchronize()
        if not bool((y == (0 if i % 2 == 0 else world)).all()):
            ok = False
            print(f"rank {rank} {level} {dt}: constant input {i % 2} gave {y.unique().tolist()[:4]}", flush=True)
    # ONE captured launch replayed with changing input (test_quick_allreduce.py:243-305: the tile counters live on the
    # device, a replay must not see the previous round's flags)
    mark("constants done")
    if world == 2 or (level in ("FP", "INT4") and dt == torch.float16):   # (the reference replays at TP 4 and 8, test_quick_allreduce.py:308-316)
        n = 16384 * (9 if world == 2 else 4)
        inp = torch.empty(n, dtype=dt, device=dev)
        out = torch.empty(n, dtype=dt, device=dev)
        inp.fill_(1.0)
        qr.quick_all_reduce(inp, out=out)
        torch.cuda.synchronize()
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            qr.quick_all_reduce(inp, out=out)
        torch.cuda.synchronize()
        dist.barrier()
        for v in ((1, 2, 3) if world == 2 else (2, 3)):
            inp.fill_(float(v))
            dist.barrier()
            graph.replay()
            torch.cuda.synchronize()
            dist.barrier()
            if not bool((out.float() == float(v * world)).all()):
                ok = False
                print(f"rank {rank} {level} {dt} replay {v}: got {out.float().unique().tolist()[:4]}", flush=True)
        del graph
    mark("graph replays done")
    if dt == torch.float16 and level in ("FP", "INT6"):   # option qr_fenced: the release / acquire form of the flag handshake, same bits
        L.set_option("qr_fenced", 1)
        seed += 1
        check(qr, parts_for(seed, SIZES[-1], dt, "acts"), dt, f"{level} {dt} fenced")
        L.set_option("qr_fenced", 0)
    if qr.check_errors() != 0:
        ok = False
        print(f"rank {rank}: device error word set", flush=True)
    if world == 2 and level == "INT4" and dt == torch.float16:
        # GroupCoordinator.all_reduce's order (parallel_state.py:886-900): a message inside the level's size window goes
        # to the quick kernel through the group; one below it does not
        tp = TPGroup(None, quick_ar=qr)
        big = parts_for(999, 1 << 19, dt, "acts")          # 1 MiB: the fp16 / world 2 / INT4 minimum
        xb = big[rank].to(dev)
        L.load().rx_set_option(b"qr_max_blocks", 8)
        tp.all_reduce(xb)
        torch.cuda.synchronize()
        want = O.quick_allreduce([bits(p) for p in big], False, 3, False, rcp_f16_table=table)
        if not np.array_equal(bits(xb), want) or "quick_allreduce_kernel" not in L.last_dispatch():
            ok = False
            print(f"rank {rank}: TPGroup.all_reduce did not take the quick kernel for a 1 MiB message", flush=True)
        # ... and on the communication stream (all_reduce_async: the object's second context), interleaved with a reduce on the
        # caller's stream: two ordered streams of launches, one context each
        big2 = parts_for(998, 1 << 19, dt, "acts")
        xa, xm = big2[rank].to(dev), big[rank].to(dev)
        h = tp.all_reduce_async(xa)
        tp.all_reduce(xm)
        h.wait()
        torch.cuda.synchronize()
        want2 = O.quick_allreduce([bits(p) for p in big2], False, 3, False, rcp_f16_table=table)
        if not np.array_equal(bits(xa), want2) or not np.array_equal(bits(xm), want):
            ok = False
            print(f"rank {rank}: side-stream / main-stream quick all-reduces disagree with the oracle", flush=True)
        if qr.size_ok(dt, (1 << 20) - 16) or qr.size_ok(torch.float32, 1 << 22):
            ok = False
            print(f"rank {rank}: size gate", flush=True)
        L.set_option("qr_max_blocks", 4)
if world == 2:
    # a peer that never arrives: the wait is BOUNDED (option ar_spin_log2 polls), the kernel raises RX_DEVERR_AR_TIMEOUT in the
    # context's error word and returns -- no hang.  Own context: its tile counters are out of step afterwards.
    qt = QuickAllReduce(None, dev, regime="FP", lanes=1)
    dist.barrier()
    if rank == 0:
        L.set_option("ar_spin_log2", 12)
        y = qt.quick_all_reduce(torch.ones(16384, dtype=torch.float16, device=dev))
        torch.cuda.synchronize()
        L.set_option("ar_spin_log2", 27)
        if qt.check_errors() != 2:   # RX_DEVERR_AR_TIMEOUT
            ok = False
            print("rank 0: a reduce without its peer did not raise the timeout word", flush=True)
    dist.barrier()
    qt.close()
if world == 2:
    # the op-level surface in the reference's own call order (qr_variable_input, test_quick_allreduce.py:189-240:
    # init_custom_qr -> qr_get_handle -> all_gather_object -> qr_open_handles -> qr_all_reduce(ptr, inp, out, 3, cast_bf2half=True))
    import sglang_amd.quick_ar_ops as ops
    assert ops.IS_QUICK_AR_AVAILABLE and ops.qr_max_size() == 1 << 31
    _ptr = ops.init_custom_qr(rank, world, None)
    handle = ops.qr_get_handle(_ptr)
    assert handle.dtype == torch.uint8 and handle.numel() == 64 and not handle.is_cuda
    handles = [None] * world
    dist.all_gather_object(handles, handle)
    ops.qr_open_handles(_ptr, handles)
    dist.barrier()
    for num in range(1, 9):
        s2 = 1024 if num % 2 == 0 else 2048
        inp1 = (torch.zeros if num % 2 == 0 else torch.ones)((64, s2), dtype=torch.float16, device=dev)
        result = torch.empty_like(inp1)
        ops.qr_all_reduce(_ptr, inp1, result, 3, cast_bf2half=True)
        torch.cuda.synchronize()
        if not bool(torch.all(result == (0 if num % 2 == 0 else world))):
            ok = False
            print(f"rank {rank}: op-level variable-input round {num} wrong", flush=True)
    try:
        ops.qr_all_reduce(_ptr, torch.zeros(8, device=dev), torch.zeros(8, device=dev), 0)
        ok = False
    except RuntimeError as e:   # quick_all_reduce.cu:82-84
        assert "float16 and bfloat16" in str(e)
    dist.barrier()
    ops.qr_close_peers(_ptr)
    dist.barrier()
    ops.qr_destroy(_ptr)
    ops.qr_destroy(0)   # (null handle: a no-op, as the reference's check)
qr.close()
dist.barrier()
print("RANK_OK" if ok else "RANK_FAIL", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_quick_allreduce_across_processes(world, tmp_path):
    """World sizes of the reference's QuickAllReduce (quick_all_reduce.py:51 _SUPPORTED_WORLD_SIZES = [2, 4, 8]).  World 2
    runs every level x dtype mode x size x data kind; 4 and 8 every level x dtype mode once (all 36 kernel instances run)."""
    script = tmp_path / "qr_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + world), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ROCM_QUICK_REDUCE_QUANTIZATION", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300 + 30 * world)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\nTIMEOUT"
        outs.append(out)
    print(f"[world {world}] rank 0 timeline:\n" + "\n".join(ln for ln in outs[0].splitlines() if ln.startswith("[t+")))  # (pytest -s / -rA)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RANK_OK" in out, f"rank {r}:\n{out[-3000:]}"
