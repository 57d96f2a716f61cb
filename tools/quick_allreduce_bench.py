"""Time the quick all-reduce (csrc/rx_quick_allreduce.hip) with W PROCESSES ON ONE GPU (the pool has one GPU per box):
    [QR_MAX_BLOCKS=512] python tools/quick_allreduce_bench.py [W=2] [MiB=64]
What this measures: the codec arithmetic + the HBM traffic of W co-resident kernels sharing one device -- an upper bound on
the kernel's own cost per message byte.  What it does NOT measure: xGMI.  On a node the kernel is link bound (each GPU pushes
(W-1)/W of the ENCODED message twice over W-1 links); this number only shows whether the codec could keep up with the links.
Rank 0 prints one JSON line per (level, dtype mode)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["RX_ROOT"])
from sglang_amd.parallel import QuickAllReduce, QuickReduceRegime
rank, world, mib = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["QR_MIB"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
if os.environ.get("QR_MAX_BLOCKS"):   # W kernels of 1024 workgroups oversubscribe ONE GPU (5 fit per CU): cap each rank's grid
    from sglang_amd import lib as L
    L.set_option("qr_max_blocks", int(os.environ["QR_MAX_BLOCKS"]))
qr = QuickAllReduce(None, dev, regime="FP")
n = mib * (1 << 20) // 2
for dt, cast in ((torch.float16, 0), (torch.bfloat16, 1), (torch.bfloat16, 0)):
    x = torch.randn(n, device=dev).to(dt)
    y = torch.empty_like(x)
    for level in ("FP", "INT8", "INT6", "INT4"):
        qr.qr_quant_level, qr.use_fp16_kernels = QuickReduceRegime[level], cast
        for _ in range(3):
            qr.quick_all_reduce(x, out=y)
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            qr.quick_all_reduce(x, out=y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        t = torch.tensor([ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            wire = {"FP": 4096, "INT8": 2176, "INT6": 1664, "INT4": 1152}[level] / 4096
            print("QRBENCH " + json.dumps({"world": world, "message_MiB": mib, "dtype": str(dt).split(".")[-1], "bf16_as_fp16": bool(cast),
                  "level": level, "grid_cap": int(os.environ.get("QR_MAX_BLOCKS", 0)), "ms_per_call": round(t.item(), 4), "message_GB_per_s_per_rank": round(mib / 1024 * 1.073741824 / (t.item() / 1e3), 1),
                  "wire_bytes_per_message_byte": round(wire, 3), "note": "W processes share ONE GPU: codec + HBM cost, not xGMI"}), flush=True)
assert qr.check_errors() == 0
qr.close()
dist.destroy_process_group()
'''

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "w.py")
        open(script, "w").write(WORKER)
        env = dict(os.environ, RX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29811", WORLD_SIZE=str(world), QR_MIB=str(mib),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for r in range(world)]
        rc = 0
        for r, p in enumerate(procs):
            out, _ = p.communicate(timeout=600)
            rc |= p.returncode
            for ln in out.splitlines():
                if ln.startswith("QRBENCH "):
                    print(ln[8:])
            if p.returncode != 0:
                print(f"rank {r} failed:\n{out[-2000:]}", file=sys.stderr)
        sys.exit(rc)
