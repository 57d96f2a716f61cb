"""C1: the peer-to-peer two-shot all-reduce (csrc/rx_allreduce.hip) across PROCESSES through IPC-mapped
regions.  The gpurun box has one GPU, so the ranks share cuda:0 -- that exercises the handle exchange,
the flag protocol, buffer alternation and the arithmetic, not xGMI coherence (DESIGN.md says so).
gloo carries the 64-byte handles."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["RX_ROOT"])
from sglang_amd.parallel import CustomAllReduce, TPGroup
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
ar = CustomAllReduce(None, dev, max_bytes=4 << 20)
tp = TPGroup(None, custom_ar=ar)
ok = True
for it, (n, dt) in enumerate([(8, torch.bfloat16), (256 * 4096, torch.bfloat16), (1000 * 8, torch.float16),
                              (2 << 20, torch.bfloat16), (4096, torch.float16)] * 3):
    g = torch.Generator().manual_seed(100 * it)
    parts = [torch.randn(n, generator=g).to(dt) for _ in range(world)]   # same on every rank
    x = parts[rank].to(dev)
    want = sum(p.float() for p in parts).to(dt)                          # fp32 sum in rank order, one rounding
    if it % 2 == 0:
        tp.all_reduce(x)                      # in place, current stream
        got = x
    else:
        got = tp.all_reduce_async(x).wait()   # side stream + events
    torch.cuda.synchronize()
    if not torch.equal(got.cpu(), want):
        ok = False
        print(f"rank {rank} it {it} n {n}: max diff", (got.cpu().float() - want.float()).abs().max().item(), flush=True)
assert ar.check_errors() == 0
assert not ar.supports(torch.zeros(7, device=dev, dtype=torch.bfloat16))   # falls back to the group
ar.close()
dist.destroy_process_group()
print("RANK_OK" if ok else "RANK_FAIL", flush=True)
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world", [2, 4])
def test_custom_allreduce_across_processes(world, tmp_path):
    script = tmp_path / "ar_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + world),
               WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\nTIMEOUT"
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RANK_OK" in out, f"rank {r}:\n{out[-2000:]}"
