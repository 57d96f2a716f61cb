#!/usr/bin/env python3
"""Per-MFMA-gap instruction table of a kernel's hottest loop, from the ISA.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 <the file's flags from sglang_amd/build.py> -I include -I sglang_amd/csrc \
          -S --cuda-device-only sglang_amd/csrc/rx_extend32.hip -o /tmp/rx_extend32.s
    python tools/isa_gaps.py /tmp/rx_extend32.s 'extend_mfma32_kernel<rx::BF16, long, false, false, 8, false, true, 4>' [long:.LBB1_108 ...]

The loop is the innermost backward branch whose body holds the most MFMAs.  Inside it the HOT path is followed: at a
forward conditional branch the shorter side is taken (the long side is a slow path -- the sum check's redo, the
accumulator rescale -- and is listed with its size so the choice can be audited); `long:<label>` takes the longer side
of the branch to that label instead (the prefix-tile side of the slot-id loads).  A "gap" is everything issued after
one v_mfma up to the next; classes: VALU (non-transcendental), TRANS (v_exp / v_rcp / v_log / v_sqrt / v_rsq / v_sin /
v_cos), LDS (ds_*), VMEM (global_* / buffer_* / flat_*), SALU (s_* except waits / nops / barriers / branches), WAIT
(s_waitcnt*), NOP (s_nop: hazard padding), BAR (s_barrier), BR (branches)."""
import re
import subprocess
import sys

TRANS = ("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")
COLS = ("VALU", "TRANS", "LDS", "VMEM", "SALU", "WAIT", "NOP", "BAR", "BR")


def classify(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith(TRANS):
        return "TRANS"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op == "s_nop":
        return "NOP"
    if op == "s_barrier":
        return "BAR"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BR"
    if op.startswith("s_"):
        return "SALU"
    return "VALU"


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2]
    take_long = {a[5:] for a in sys.argv[3:] if a.startswith("long:")}
    # function labels are mangled: demangle them all once
    labels = re.findall(r"^(_Z\w+):", text, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(labels), capture_output=True, text=True).stdout.splitlines()
    name = next(m for m, d in zip(labels, dem) if want in d)
    body = text[text.index("\n" + name + ":"):]
    body = body[: body.index(".Lfunc_end")]
    lines = []
    for ln in body.split("\n")[2:]:
        ln = ln.split(";")[0].strip()
        if ln and not ln.startswith((".", "#")) or re.match(r"\.LBB\d+_\d+:", ln):
            lines.append(ln)
    pos = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    # innermost loop with the most MFMAs (or `loop:<label>`: the loop that starts at that label, up to its last backward branch)
    best = None
    forced = next((a[5:] for a in sys.argv[3:] if a.startswith("loop:")), None)
    for i, ln in enumerate(lines):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if forced:
            if m and m.group(1) == forced and pos[forced] < i:
                best = (pos[forced], i, sum(1 for x in lines[pos[forced]:i] if x.startswith("v_mfma")))
            continue
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            lo = pos[m.group(1)]
            n = sum(1 for x in lines[lo:i] if x.startswith("v_mfma"))
            inner = any(re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", x) and pos.get(x.split()[-1], 1 << 30) < j + lo and
                        pos.get(x.split()[-1], -1) >= lo for j, x in enumerate(lines[lo:i]))
            if not inner and (best is None or n > best[2]):
                best = (lo, i, n)
    lo, hi, n_mfma = best
    # hot path through [lo, hi]
    hot, skipped, i = [], [], lo
    while i <= hi:
        ln = lines[i]
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if m and i < hi and lo < pos.get(m.group(1), -1) <= hi and pos[m.group(1)] > i:
            tgt = pos[m.group(1)]
            fall = lines[i + 1:tgt]
            hot.append(ln)
            if fall and re.match(r"s_branch\s+(\.LBB\d+_\d+)", fall[-1]) and pos.get(fall[-1].split()[-1], -1) > tgt:
                join = pos[fall[-1].split()[-1]]  # if / else: the shorter side is hot
                other = lines[tgt:join]
                if (len(fall) <= len(other)) != (m.group(1) in take_long):
                    hot += fall
                    skipped.append((m.group(1), len(other)))
                else:
                    hot += other
                    skipped.append(("fallthrough of " + ln.split()[0] + " " + m.group(1), len(fall)))
                i = join
            else:  # if without else: the guarded block is the slow side
                skipped.append(("guarded by " + ln.split()[0] + " " + m.group(1), len(fall)))
                i = tgt
            continue
        hot.append(ln)
        i += 1
    gaps, cur = [], {c: 0 for c in COLS}
    pre = None
    for ln in hot:
        if ln.endswith(":"):
            continue
        c = classify(ln.split()[0])
        if c == "MFMA":
            if pre is None:
                pre = cur
            else:
                gaps.append(cur)
            cur = {c2: 0 for c2 in COLS}
        else:
            cur[c] += 1
    tail = cur
    print(f"kernel: {dem[labels.index(name)]}")
    print(f"loop: {lines[lo]} .. {lines[hi]}  ({n_mfma} MFMAs in the body, {len([x for x in hot if x.startswith('v_mfma')])} on the hot path)")
    print("slow sides left out (label, instructions):", skipped)
    print("gap  " + " ".join(f"{c:>5s}" for c in COLS) + "  total")
    tot = {c: 0 for c in COLS}
    rows = [("head", pre)] + [(str(k + 1), g) for k, g in enumerate(gaps)] + [("tail", tail)]
    for nm, g in rows:
        print(f"{nm:>4s} " + " ".join(f"{g[c]:5d}" for c in COLS) + f"  {sum(g.values()):5d}")
        for c in COLS:
            tot[c] += g[c]
    nm_ = len(gaps) + 1
    print(" sum " + " ".join(f"{tot[c]:5d}" for c in COLS) + f"  {sum(tot.values()):5d}")
    print(f"per MFMA ({nm_}): " + ", ".join(f"{c} {tot[c] / nm_:.2f}" for c in COLS) +
          f"; non-MFMA issues per MFMA {sum(tot.values()) / nm_:.2f}")
    fill = [sum(g.values()) for _, g in rows[1:-1]]
    print(f"fillers per gap: max {max(fill)}, gaps with > 5: {sum(1 for f in fill if f > 5)} of {len(fill)}; "
          f"gaps with > 1 transcendental: {sum(1 for _, g in rows[1:-1] if g['TRANS'] > 1)}")


if __name__ == "__main__":
    main()
