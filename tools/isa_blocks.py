#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -save-temps .s file (dev tool):
    python tools/isa_blocks.py file.s kernel_substring [min_lines]"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
m = re.search(r"\n(_Z\w*%s\w*):[^\n]*\n" % re.escape(key), s)
name = m.group(1)
k = s[m.end():]
k = k[: k.index(".Lfunc_end")]
blocks = re.split(r"\n(\.LBB\d+_\d+):", "\n.LBB0_entry:" + k) if not k.startswith(".LBB") else re.split(r"\n(\.LBB\d+_\d+):", k)
print(name)
for i in range(1, len(blocks), 2):
    nm, body = blocks[i], blocks[i + 1]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    c = lambda p: sum(1 for l in lines if re.match(p, l))  # noqa: E731
    if len(lines) >= mn:
        print(f"{nm:12s} n={len(lines):4d} mfma={c('v_mfma'):3d} acc={c('v_accvgpr'):3d} scratch={c('scratch_'):2d} "
              f"valu={c(r'v_(?!mfma|accvgpr)'):3d} exp={c('v_exp'):2d} ds={c('ds_'):2d} glds={c('global_load_lds'):2d} "
              f"gl={c(r'global_(load|store)_dword'):2d} wait={c('s_waitcnt'):2d} nop={c('s_nop'):2d} br={c('s_c?branch'):2d}")
for key2 in ("vgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size", "sgpr_count"):
    mm = re.findall(r"\.%s:\s+(\d+)" % key2, s)
    print(key2, mm)
