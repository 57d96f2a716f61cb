"""Read v_rcp_f16 of all 65536 fp16 bit patterns back from the GPU (rx_rcp_f16_table) and save the table:
    python tools/dump_rcp_f16.py gpurun_out/rcp_f16_gfx950.npy
The committed copy is tests/golden/rcp_f16_gfx950.npy (the fp16 codecs of the quick all-reduce take their encode scale
from this instruction, which the ISA specifies to 1 ulp; oracle/radix_oracle.py takes the table).  Prints how many entries
differ from the correctly rounded reciprocal."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import lib as L  # noqa: E402


def read_table(device="cuda:0") -> np.ndarray:
    out = torch.zeros(65536, dtype=torch.int16, device=device)
    L.check(L.load().rx_rcp_f16_table(C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "rx_rcp_f16_table")
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint16)


if __name__ == "__main__":
    t = read_table()
    with np.errstate(all="ignore"):
        x = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float64)
        cr = (1.0 / x).astype(np.float16).view(np.uint16)
    nan = np.isnan(x)
    diff = (t != cr) & ~nan
    print(f"entries differing from the correctly rounded reciprocal: {int(diff.sum())} of {int((~nan).sum())} non-NaN inputs")
    for i in np.flatnonzero(diff)[:10]:
        print(f"  x = 0x{i:04x} ({x[i]!r}): hardware 0x{t[i]:04x}, correctly rounded 0x{cr[i]:04x}")
    if len(sys.argv) > 1:
        np.save(sys.argv[1], t)
        print("saved", sys.argv[1])
