#!/usr/bin/env python3
"""Config-3 extend chunk at every MFMA head-dim pair: TFLOP/s per (Dk, Dv).  python tools/extend_dims.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
for d, dv in ((128, 128), (64, 64), (96, 96), (192, 128), (192, 192), (256, 256)) if not os.environ.get("DIMS") else [tuple(int(x) for x in t.split("x")) for t in os.environ["DIMS"].split(",")]:
    r = bench.extend_bench(args, dev, int(os.environ.get("TP", "1")), d, dv, nchunks=3)  # TP=8: one rank's heads of a TP=8 job (Hq 4 / Hkv 1)
    print(json.dumps({"dk": d, "dv": dv, "tflops": round(r["tflops"], 1), "ms": round(r["ms_per_chunk"], 3)}))
