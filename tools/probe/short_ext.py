"""Dev: short extends over a prefix at the head dims of the AGPR template -- the template (packed rows) against the
kernel a short call takes otherwise (option extend_d256_min_rows).  python tools/probe/short_ext.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from sglang_amd import lib as rxlib
args = bench.parse(); dev = torch.device("cuda:0")
for d, dv, shape in ((256, 256, (1024, 32, 64)), (256, 256, (2048, 16, 64)), (256, 256, (4096, 8, 64)), (64, 64, (2048, 16, 64)),
                     (64, 64, (2048, 32, 64)), (192, 128, (2048, 16, 64)), (96, 96, (2048, 8, 64)), (256, 256, (0, 32, 256))):
    row = []
    for mr in (129, 1):
        with rxlib.option("extend_d256_min_rows", mr):
            r = bench.extend_bench(args, dev, 1, d, dv, nchunks=10, layers=2, shape=shape)
        row.append(f"min_rows {mr}: {r['kernel_only']['tflops']:.0f} (path {r['tflops']:.0f}) {r['kernel'].split('::')[1].split('<')[0]}")
    print(d, dv, shape, " | ".join(row), flush=True)
