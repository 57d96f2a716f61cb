#!/usr/bin/env python3
"""Audit of a kernel that owns accumulator registers by NAME (rx::extend_mfma64_kernel, cdna_hip_programming.md 5.7 item 4):
in the compiler's .s output, outside the ;;#ASMSTART / ;;#ASMEND brackets of the kernel's own asm statements there must be
NO instruction that names an accumulator register (a compiler spill into a[..] or a v_accvgpr_* of its own would corrupt
the kernel's O / Q^T silently), no scratch traffic, and the descriptor must allocate exactly the owned registers.

    python tools/audit_acc_ownership.py <file.s> [<owned agprs, default: every kernel's own .amdhsa_accum count is printed>]
Exit code 1 on a violation."""
import re
import sys


def audit(text):
    bad, stats = [], {}
    name, inside, in_fn = None, False, False
    for ln in text.split("\n"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, inside, in_fn = m.group(1), False, True
            stats[name] = {"asm_acc": 0, "lines": 0}
            continue
        if not in_fn:
            continue
        if ".Lfunc_end" in ln:
            in_fn = False
            continue
        if ";;#ASMSTART" in ln:
            inside = True
            continue
        if ";;#ASMEND" in ln:
            inside = False
            continue
        code = ln.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        stats[name]["lines"] += 1
        names_acc = re.search(r"\ba\[?\d+", code) is not None or "accvgpr" in code
        if inside:
            stats[name]["asm_acc"] += names_acc
        elif names_acc or code.startswith(("scratch_", "buffer_store", "buffer_load")) and "lds" not in code:
            bad.append((name, code))
    return bad, stats


def main():
    text = open(sys.argv[1]).read()
    bad, stats = audit(text)
    for n, st in stats.items():
        if st["asm_acc"]:
            print(f"{n}: {st['lines']} instructions, {st['asm_acc']} accumulator-file references, all inside the kernel's asm statements")
    for n, code in bad[:20]:
        print(f"VIOLATION in {n}: {code}")
    spills = re.findall(r"\.(?:vgpr|sgpr)_spill_count:\s*(\d+)", text) + re.findall(r"\.private_segment_fixed_size:\s*(\d+)", text)
    if any(int(x) for x in spills):
        print("VIOLATION: spill or scratch in the metadata:", spills)
        bad.append(("metadata", "spill"))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
