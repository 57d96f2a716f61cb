#!/usr/bin/env python3
"""Dev sweep: the D = 128 extend kernel's two workgroup forms over (prefix, extend) shapes -- four waves x 32 rows,
unpacked (option ext32_small_wg_tiles large), eight waves unpacked, and eight waves with self-packed GQA rows (option
ext32_pack_min_len) -- to place the launcher's switches.  SHAPES="P,E,chunk;..."  python tools/extend_forms.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sglang_amd import lib as rxlib  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
shapes = os.environ.get("SHAPES", "0,2048,8;0,512,32;0,256,64;0,128,128;512,128,32;512,64,64;2048,128,32;2048,64,64;4096,128,32;4096,32,64;512,512,32;3584,512,32")
for sh in shapes.split(";"):
    P, E, chunk = (int(x) for x in sh.split(","))
    row = []
    for name, thr, pml, pmt in (("4w", 100000, 1 << 30, 4), ("8w", 0, 1 << 30, 4), ("8w-packed", 0, 1, 0), ("default", 28, 32, 4)):
        with rxlib.option("ext32_small_wg_tiles", thr), rxlib.option("ext32_pack_min_len", pml), rxlib.option("ext32_pack_min_tiles", pmt):
            r = bench.extend_bench(args, dev, int(os.environ.get("TP", "1")), shape=(P, E, chunk), layers=2, nchunks=10)
        row.append(f"{name} {r['kernel_only']['tflops']:.0f} (path {r['tflops']:.0f})")
    print(f"P={P} E={E} x{chunk}: est_tiles={(P + E // 2) // 64}  " + " | ".join(row), flush=True)
