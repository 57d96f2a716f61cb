import os, sys, tempfile, subprocess
ROOT="/root/repo"
WORKER = r'''
import json, os, sys, ctypes as C
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["RX_ROOT"])
from sglang_amd.parallel import CustomAllReduce
from sglang_amd import lib as L
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
ar = CustomAllReduce(None, dev, max_bytes=8 << 20, lanes=1)
lib = L.load()
n = 256 * 1024 // 2
x = torch.randn(n, device=dev).bfloat16(); y = torch.empty_like(x)
cp = C.c_void_p
def direct(fn, inp, out):
    L.check(fn(ar._ctxs[0], cp(inp.data_ptr()), cp(out.data_ptr()), inp.numel(), L.RX_BF16, cp(torch.cuda.current_stream().cuda_stream)), "x")
kinds = {"two_shot_inplace": lambda: direct(lib.rx_allreduce, x, x), "two_shot_oop": lambda: direct(lib.rx_allreduce, x, y),
         "det_inplace": lambda: direct(lib.rx_allreduce_det, x, x), "det_oop": lambda: direct(lib.rx_allreduce_det, x, y),
         "det_wrapper": lambda: ar.all_reduce_det(x)}
for kind, run in kinds.items():
    for _ in range(3): run()
    torch.cuda.synchronize(); dist.barrier()
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(16): run()
    torch.cuda.synchronize(); dist.barrier()
    g.replay(); torch.cuda.synchronize(); dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    if rank == 0: print("PROBE", kind, round(e0.elapsed_time(e1) / 80 * 1e3, 2), flush=True)
    dist.barrier(); del g
ar.close(); dist.destroy_process_group()
'''
with tempfile.TemporaryDirectory() as d:
    script=os.path.join(d,"w.py"); open(script,"w").write(WORKER)
    env=dict(os.environ, RX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29813", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps=[subprocess.Popen([sys.executable, script], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for p in ps:
        out,_=p.communicate(timeout=300)
        print("\n".join(l for l in out.splitlines() if l.startswith("PROBE") or "Error" in l or "error" in l))
