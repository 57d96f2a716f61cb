import os, sys, json, time
sys.argv = ["bench.py", "--no-extend", "--no-radix-hit", "--no-cpu-baseline"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
args = bench.parse()
rank, world, lr = bench.build_world(args)
dev = torch.device("cuda", lr)
from sglang_amd.forward_batch import ForwardBatch
st = bench.make_decode_state(args, 1, dev)
fb = ForwardBatch.for_decode(st.req_pool_indices, st.seq_lens, st.out_cache_loc, st.seq_lens_cpu)
st.ev_stride = 1
for rep in range(6):
    st.ev_pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 5 * args.layers)]
    pairs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        bench.decode_step(st, fb, 1, pairs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    d = np.array([a.elapsed_time(b) for a, b in pairs]).reshape(5, args.layers)
    per_layer = d.mean(0)
    print(f"rep {rep}: step {dt*1e3:.2f} ms, launch mean {d.mean()*1e3:.0f} us, per-layer min {per_layer.min()*1e3:.0f} max {per_layer.max()*1e3:.0f}; slow layers (>740us): {[int(i) for i in np.where(per_layer > 0.74)[0]]}")
    time.sleep(0.5)
