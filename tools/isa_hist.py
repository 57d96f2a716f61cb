#!/usr/bin/env python3
"""Opcode histogram of the hot path of given basic blocks (each cut at its first conditional branch):
    python tools/isa_hist.py file.s kernel_substring LBB0_131 LBB0_135 ..."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r"\n(_Z\w*%s\w*):[^\n]*\n" % re.escape(sys.argv[2]), s)
k = s[m.end():]
k = k[: k.index(".Lfunc_end")]
blocks = re.split(r"\n(\.LBB\d+_\d+):", k)
d = {blocks[i].lstrip("."): blocks[i + 1] for i in range(1, len(blocks), 2)}
hist = collections.Counter()
for name in sys.argv[3:]:
    lines = [l.strip() for l in d[name].split("\n") if l.strip() and not l.strip().startswith(";")]
    n = 0
    for l in lines:
        hist[l.split()[0]] += 1
        n += 1
        if l.startswith("s_cbranch"):
            break
    print(name, n, "instructions on the hot path")
valu = sum(c for op, c in hist.items() if op.startswith("v_") and not op.startswith(("v_mfma", "v_accvgpr")))
print("VALU", valu, "MFMA", sum(c for op, c in hist.items() if op.startswith("v_mfma")), "SALU",
      sum(c for op, c in hist.items() if op.startswith("s_")))
for op, c in hist.most_common(70):
    print(f"{op:30s}{c}")
