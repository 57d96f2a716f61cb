import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
args = bench.parse(); dev = torch.device("cuda:0")
for d, dv, shape in ((64, 64, (2048, 64, 64)), (64, 64, (512, 128, 32)), (192, 128, (2048, 64, 64)), (256, 256, (1024, 32, 64)), (96, 96, (2048, 96, 32)), (64, 64, (0, 128, 64))):
    r = bench.extend_bench(args, dev, 1, d, dv, nchunks=10, layers=2, shape=shape)
    print(d, dv, shape, round(r["kernel_only"]["tflops"], 1), r["kernel"], flush=True)
