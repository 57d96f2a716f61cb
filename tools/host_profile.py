#!/usr/bin/env python3
"""Dev: cProfile of the host-side enqueue path of one decode step (TP=8 shard shapes)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = ["bench.py", "--tp-sim", "8", "--ctx", "512"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from sglang_amd.forward_batch import ForwardBatch  # noqa: E402

st = bench.make_decode_state(args, 8, dev)
fb = ForwardBatch.for_decode(st.req_pool_indices, st.seq_lens, st.out_cache_loc, st.seq_lens_cpu)
for _ in range(3):
    bench.decode_step(st, fb, 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    bench.decode_step(st, fb, 1)
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr).sort_stats("tottime")
ps.print_stats(18)
