"""Compile oracle/rx_oracle.c -> oracle/librx_oracle.so (gcc, OpenMP).  Checker only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rx_oracle.c")
LIB = os.path.join(HERE, "librx_oracle.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    # x86-64-v3 (AVX2+FMA): runs on the build container's Xeon and the GPU box's EPYC alike
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-o", LIB + ".tmp", SRC,
           "-lm"]
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
