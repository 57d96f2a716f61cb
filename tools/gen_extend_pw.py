#!/usr/bin/env python3
"""Generates sglang_amd/csrc/rx_extend_pw_body.inc: the hand-scheduled steady-state iteration of
rx::extend_pw_kernel (csrc/rx_extend_pw.hip) -- 64 MFMA gaps per 64-token tile, every gap written out.

Why generated: one wave per SIMD issues in order, so a VALU instruction overlaps the matrix pipe only inside the
32-cycle shadow of the MFMA in front of it (MI355X_MICROARCH.md, 'one wave per SIMD ... single-issue instructions
HIDDEN per v_mfma_f32_32x32x16 gap').  The softmax of a (32 queries x 32 keys) block is therefore cut into micro-ops
(scale-subtract F, exp2 X, row-sum add A, pack C, the row-max steps) and dealt over the 16 gaps of that query block
with the stages of one element in DIFFERENT gaps (F one gap before X, A / C one gap after): no gap holds a dependent
chain, none holds more than two exp2.  Issue cost per gap (4 cycles per plain VALU, 8 per exp2): 12..28, mean 22.

Pipeline of one iteration (tile t; b0 / b1 = its two 32-key blocks; 16 MFMAs per group, qb = the wave's two 32-query
blocks alternate):
    G1  QK^T(b0, t)    | stream (b1, t-1) slots 8-15         K fragments of tile t by ds_read_b128
    G2  QK^T(b1, t)    | stream (b0, t)   slots 0-7
    G3  PV(b1, t-1)    | stream (b0, t)   slots 8-15         V^T fragments of tile t, k-steps 0,1
    G4  PV(b0, t)      | stream (b1, t)   slots 0-7          V^T fragments of tile t, k-steps 2,3 -> vfc (for G3 of t+1)
Every LDS read of an iteration comes from tile t, so the K/V ring is two tiles deep and the fragment addresses flip between the
two slots by one XOR each per tile; the 8 LDS-DMA pieces of tile t+1 and the reads of their
4 row offsets are spread over the qb = 1 gaps.

    python tools/gen_extend_pw.py        # rewrites the .inc next to the kernel
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "sglang_amd", "csrc", "rx_extend_pw_body.inc")
OUT_DRAIN = os.path.join(ROOT, "sglang_amd", "csrc", "rx_extend_pw_drain.inc")

# ---- the softmax stream of one (block, query block): 16 slots + a tail -----------------------------------------
# element e: F (t = fma(s, c2, -m)) one slot before X (p = exp2 t), A (row sum) and C (pack of a finished pair)
# one slot after.  X per slot 4..15: 1 1 2 1 1 2 1 1 2 1 1 2.
X_SLOTS = {}
_e = 0
for _s, _n in zip(range(4, 16), [1, 1, 2] * 4):
    X_SLOTS[_s] = list(range(_e, _e + _n))
    _e += _n
assert _e == 16
F_SLOTS = {s - 1: es for s, es in X_SLOTS.items()}          # slot 3 .. 14
A_SLOTS = {s + 1: es for s, es in X_SLOTS.items()}          # slot 5 .. 16 (16 = tail)
C_SLOTS = {}
for s, es in X_SLOTS.items():
    for e in es:
        if e & 1:
            C_SLOTS.setdefault(s + 1, []).append(e >> 1)


def stream_slot(blk, qb, slot):
    """C++ statements of one slot of the stream of block `blk` (0 / 1), query block qb.  Names: S = s0 / s1,
    pk = pk0 / pk1, m_prev / m_new per the chain m1(t-1) -> m0(t) -> m1(t)."""
    S = f"s{blk}[{qb}]"
    PK = f"pk{blk}[{qb}]"
    ma, mb = f"ma{blk}[{qb}]", f"mb{blk}[{qb}]"
    mprev = f"m{1 - blk}[{qb}]"      # block 0 follows block 1 of the previous tile, block 1 follows block 0
    mnew = f"m{blk}[{qb}]"
    alpha = f"alpha{blk}[{qb}]"
    psa, psb = f"psa{blk}[{qb}]", f"psb{blk}[{qb}]"
    tv = lambda e: f"tv{blk}_{qb}_{e}"  # noqa: E731  (declared by the kernel: float tvB_Q_E)
    out, anchors, launder = [], [], []
    if slot == 0:
        out += [f"{ma} = max3f({S}[0], {S}[1], {S}[2]);", f"{mb} = max3f({S}[3], {S}[4], {S}[5]);",
                f"{ma} = max3f({ma}, {S}[6], {S}[7]);", f"{mb} = max3f({mb}, {S}[8], {S}[9]);"]
        anchors += [ma, mb]
    elif slot == 1:
        launder += [ma]
        out += [f"{ma} = max3f({ma}, {S}[10], {S}[11]);", f"{mb} = max3f({mb}, {S}[12], {S}[13]);",
                f"{ma} = max3f({ma}, {S}[14], {S}[15]);", f"{ma} = max2f({ma}, {mb});"]
        anchors += [ma]
    elif slot == 2:
        launder += [ma]
        out += [f"{ma} = max2f(half_swap_max({ma}) * c2r, -1e20f);"]   # extend_attention.py:474-475 (-inf rows)
        anchors += [ma]
    elif slot == 3:
        launder += [ma]
        # thresholded running max without a compare / select (VCC hazards cost wait states): m = max(m_prev, mt - slack)
        # keeps exp2(s - m) <= 2^slack; against the select form (m = mt on a jump) P is scaled by exactly 2^slack, so
        # the roundings are the same
        out += [f"{mnew} = max2f({mprev}, {ma} - kPwSlack);",
                f"{alpha} = fast_exp2({mprev} - {mnew});"]
        anchors += [mnew, alpha]
    if slot in F_SLOTS:
        if slot != 3:
            launder += [mnew]
        for e in F_SLOTS[slot]:
            out.append(f"{tv(e)} = __builtin_fmaf({S}[{e}], c2r, -{mnew});")
            anchors.append(tv(e))
    if slot in X_SLOTS:
        for e in X_SLOTS[slot]:
            if tv(e) not in launder:
                launder.append(tv(e))
            out.append(f"{S}[{e}] = fast_exp2({tv(e)});")
            anchors.append(f"{S}[{e}]")
    if slot in A_SLOTS:
        for e in A_SLOTS[slot]:
            acc = psa if (e & 1) == 0 else psb
            out.append(f"{acc} = {S}[{e}];" if e < 2 else f"{acc} += {S}[{e}];")
            if acc not in anchors:
                anchors.append(acc)
    if slot in C_SLOTS:
        for p in C_SLOTS[slot]:
            out.append(f"{PK}[{p >> 2}][{p & 3}] = pack2<T>({S}[{2 * p}], {S}[{2 * p + 1}]);")
            anchors.append(f"{PK}[{p >> 2}][{p & 3}]")
    lines = [f'asm volatile("" : "+v"({x}));' for x in launder] + out
    if anchors:
        lines.append('asm volatile("" :: ' + ", ".join(f'"v"({x})' for x in anchors) + ");")
    return lines


def stream_tail(blk, qb):
    """slot 16: the last two row-sum adds and the last pack, then the fold of the block into l."""
    lines = stream_slot(blk, qb, 16)
    alpha = f"alpha{blk}[{qb}]"
    lines.append(f"l_run[{qb}] = l_run[{qb}] * {alpha} + (psa{blk}[{qb}] + psb{blk}[{qb}]);")
    return lines


def gen_body():
    L = []
    add = L.append

    def gap(title, pre, mfma, valu, post):
        add(f"  {{  // {title}")
        for x in pre:
            add("    " + x)
        add("    " + mfma)
        for x in valu:
            add("    " + x)
        for x in post:
            add("    " + x)
        add("    PW_FENCE();")
        add("  }")

    # DMA plan: the byte offset of row j (a ds_read_b64 from the offset table) at (group j+1, step 1, qb 1); its K piece
    # at step 3, its V piece at step 6
    def dma_post(group, i, qb):
        j = group - 1
        if qb != 1:
            return []
        if i == 1:
            return [f"PW_ROW({j});"]
        if i == 3:
            return [f"PW_DMA({j}, 0);"]
        if i == 6:
            return [f"PW_DMA({j}, 1);"]
        return []

    # ---------------- G1: QK^T(b0) | stream (b1, t-1) slots 8-15
    add("  // ======== G1: QK^T(b0, t) | stream (b1, t-1) slots 8..15")
    add("  kf[0] = PW_LDK(0, 0); kf[1] = PW_LDK(0, 1);")
    for i in range(8):
        for qb in range(2):
            pre = []
            if qb == 0:
                nxt = i + 2
                pre.append(f"kf[{nxt % 4}] = PW_LDK({nxt // 8}, {nxt % 8});")
            mf = f"PW_QK({'true' if i == 0 else 'false'}, s0[{qb}], kf[{i % 4}], {qb}, {i});"
            gap(f"G1 step {i} qb {qb}", pre, mf, stream_slot(1, qb, 8 + i), dma_post(1, i, qb))
    for qb in range(2):
        for x in stream_tail(1, qb):
            add("  " + x)
    add("  PW_FENCE();")
    # ---------------- G2: QK^T(b1) | stream (b0, t) slots 0-7
    add("  // ======== G2: QK^T(b1, t) | stream (b0, t) slots 0..7")
    for i in range(8):
        for qb in range(2):
            pre = []
            if qb == 0 and i + 2 < 8:
                nxt = 8 + i + 2
                pre.append(f"kf[{nxt % 4}] = PW_LDK(1, {nxt % 8});")
            mf = f"PW_QK({'true' if i == 0 else 'false'}, s1[{qb}], kf[{(8 + i) % 4}], {qb}, {i});"
            gap(f"G2 step {i} qb {qb}", pre, mf, stream_slot(0, qb, i), dma_post(2, i, qb))
    add("  PW_RESCALE(alpha1);   // O^T at the scale of m1(t-1), before PV(b1, t-1)")
    # ---------------- G3: PV(b1, t-1) from vfc | stream (b0, t) slots 8-15 | V(t) k-steps 0,1 -> vfa, vfb; ka toggle
    add("  // ======== G3: PV(b1, t-1) | stream (b0, t) slots 8..15")
    for g2 in range(8):
        for qb in range(2):
            pre = []
            if qb == 0 and g2 >= 6:   # the first two V^T fragments of tile t for G4 (k-step 0, db 0 / 1)
                pre.append(f"vfa[{g2 - 6}] = PW_LDV(0, {g2 - 6});")
            mf = f"PW_PVC({g2}, pk1[{qb}][{g2 // 4}], {qb}, {g2 % 4});"
            post = dma_post(3, g2, qb)
            if qb == 0 and g2 < 4:   # the K addresses move to the other ring slot (all K reads of tile t are done)
                post = post + [f"PW_TOGGLE(ka[{2 * g2}]); PW_TOGGLE(ka[{2 * g2 + 1}]);"]
            gap(f"G3 step {g2} qb {qb}", pre, mf, stream_slot(0, qb, 8 + g2), post)
    for qb in range(2):
        for x in stream_tail(0, qb):
            add("  " + x)
    add("  PW_FENCE();")
    add("  PW_RESCALE(alpha0);   // ... and of m0(t), before PV(b0, t)")
    # ---------------- G4: PV(b0, t) from vfa / vfb | stream (b1, t) slots 0-7 | V(t) k-steps 2,3 -> vfc; va toggle
    add("  // ======== G4: PV(b0, t) | stream (b1, t) slots 0..7")
    for g2 in range(8):
        for qb in range(2):
            pre = []
            if qb == 0 and g2 + 2 < 8:   # PV fragment g2 + 2 (k-step (g2 + 2) / 4, db (g2 + 2) % 4) into the ring of four
                f2 = g2 + 2
                pre.append(f"vfa[{f2 % 4}] = PW_LDV({f2 // 4}, {f2 % 4});")
            if qb == 1:                   # ... and one fragment of k-steps 2, 3 per step for G3 of the next tile
                pre.append(f"PW_LDVC({g2}, {2 + g2 // 4}, {g2 % 4});")
            src = f"vfa[{g2 % 4}]"
            mf = f"PW_PV({src}, pk0[{qb}][{g2 // 4}], {qb}, {g2 % 4});"
            post = dma_post(4, g2, qb)
            if qb == 1 and g2 == 7:       # after the last V read of tile t
                post = post + ["PW_TOGGLE(va[0]); PW_TOGGLE(va[1]); PW_TOGGLE(va[2]); PW_TOGGLE(va[3]);",
                               "PW_TOGGLE(va[4]); PW_TOGGLE(va[5]); PW_TOGGLE(va[6]); PW_TOGGLE(va[7]);"]
            gap(f"G4 step {g2} qb {qb}", pre, mf, stream_slot(1, qb, g2), post)
    return "\n".join(L) + "\n"


def gen_drain():
    """After the last tile of a run: stream (b1, t_last) slots 8-15 + tail, then PV(b1, t_last) from vfc."""
    L = []
    for i in range(8):
        for qb in range(2):
            L.append(f"  {{  // drain slot {8 + i} qb {qb}")
            for x in stream_slot(1, qb, 8 + i):
                L.append("    " + x)
            L.append("  }")
    for qb in range(2):
        for x in stream_tail(1, qb):
            L.append("  " + x)
    L.append("  PW_FENCE();")
    L.append("  PW_RESCALE(alpha1);")
    L.append("  PW_WAIT_LDS();   // the vfc fragments were read by asm statements the compiler does not count")
    for g2 in range(8):
        for qb in range(2):
            L.append(f"  PW_PVC({g2}, pk1[{qb}][{g2 // 4}], {qb}, {g2 % 4});")
    L.append("  PW_FENCE();")
    return "\n".join(L) + "\n"


HEADER = ("// GENERATED by tools/gen_extend_pw.py -- do not edit; the schedule lives in the generator.\n"
          "// Steady-state iteration of rx::extend_pw_kernel for one fully visible 64-token tile (see the generator's\n"
          "// docstring for the pipeline).  Included inside the run loop of csrc/rx_extend_pw.hip.\n")


def main():
    with open(OUT, "w") as f:
        f.write(HEADER + gen_body())
    with open(OUT_DRAIN, "w") as f:
        f.write(HEADER.replace("Steady-state iteration", "Drain of a run") + gen_drain())
    # issue-cost audit of the stream (cycles per slot)
    cost = []
    for s in range(17):
        c = 0
        c += {0: 16, 1: 16, 2: 28, 3: 24}.get(s, 0)
        c += 4 * len(F_SLOTS.get(s, [])) if s != 3 else 4
        c += 8 * len(X_SLOTS.get(s, [])) + 4 * len(A_SLOTS.get(s, [])) + 4 * len(C_SLOTS.get(s, []))
        cost.append(c)
    print("stream issue cycles per slot:", cost, "mean", sum(cost) / 16.0)


if __name__ == "__main__":
    main()
