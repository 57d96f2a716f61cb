"""Time the peer-to-peer all-reduce kernels (csrc/rx_allreduce.hip) with W PROCESSES ON ONE GPU:
    python tools/allreduce_bench.py [W=2] [KiB=2048]
Per-call time of rx_allreduce (two-shot), rx_allreduce_det (one-shot) and rx_allreduce_rmsnorm (hidden 4096) on a
decode-sized message, launches back to back inside one HIP graph (16 per replay).  One GPU: the flag handshake, the kernel's
own passes and the HBM traffic are in the number, xGMI is not."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["RX_ROOT"])
from sglang_amd.parallel import CustomAllReduce
rank, world, kib = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["AR_KIB"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
ar = CustomAllReduce(None, dev, max_bytes=max(8 << 20, kib * 1024), lanes=1)
n = kib * 1024 // 2
H = 4096
x = torch.randn(n // H, H, device=dev).bfloat16()
res = torch.randn(n // H, H, device=dev).bfloat16()
w = torch.ones(H, device=dev).bfloat16()
def run(kind):
    if kind == "two_shot":
        ar.all_reduce(x)
    elif kind == "one_shot_det":
        ar.all_reduce_det(x)
    else:
        ar.fused_allreduce_rmsnorm(x, res, w, 1e-6)
CAPTURE = torch.cuda.Stream()   # ONE capture stream for the whole run: every new stream is a new hardware queue, and once the
# processes' queues outnumber the hardware's the scheduler time-slices them -- a spinning kernel then burns its whole quantum:
# 8 processes, a fresh stream per measurement: 76 us per call for the first measurement, 21 ms (two exchanges x ~10.6 ms) after
def measure(kind):
    for _ in range(3):
        run(kind)
    torch.cuda.synchronize(); dist.barrier()
    s = CAPTURE
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(16):
            run(kind)
    torch.cuda.synchronize(); dist.barrier()
    g.replay(); torch.cuda.synchronize(); dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    t = torch.tensor([e0.elapsed_time(e1) / (reps * 16) * 1e3], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    del g
    return t.item()
# every kind is measured three times with a fresh graph and the best is quoted
for kind in ("two_shot", "two_shot", "one_shot_det", "fused_rmsnorm"):
    us = min(measure(kind) for _ in range(3))
    if rank == 0:
        print("ARBENCH " + json.dumps({"world": world, "message_KiB": kib, "kind": kind, "us_per_call": round(us, 2),
              "note": "W processes share ONE GPU; launches back to back in a graph; best of three"}), flush=True)
assert ar.check_errors() == 0
ar.close()
dist.destroy_process_group()
'''

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    kib = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "w.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, RX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29812", WORLD_SIZE=str(world), AR_KIB=str(kib),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for r in range(world)]
        rc = 0
        for r, p in enumerate(procs):
            out, _ = p.communicate(timeout=600)
            rc |= p.returncode
            for ln in out.splitlines():
                if ln.startswith("ARBENCH "):
                    print(ln[8:])
            if p.returncode != 0:
                print(f"rank {r} failed:\n{out[-2000:]}", file=sys.stderr)
        sys.exit(rc)
