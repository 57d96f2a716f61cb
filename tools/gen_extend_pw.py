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
# dev variants (A/B builds): PW_GEN_TAG names the output, the other knobs change the schedule
TAG = os.environ.get("PW_GEN_TAG", "")
OUT = os.path.join(ROOT, "sglang_amd", "csrc", f"rx_extend_pw_body{TAG}.inc")
OUT_DRAIN = os.path.join(ROOT, "sglang_amd", "csrc", "rx_extend_pw_drain.inc")
DMA_EARLY = int(os.environ.get("PW_GEN_DMA_EARLY", "0"))   # 1: all eight pieces in G1 (steps 0..7, qb 1)
NO_FENCE = int(os.environ.get("PW_GEN_NO_FENCE", "0"))     # 1: no sched_barrier between gaps (hipcc schedules)
NO_SOFTMAX = int(os.environ.get("PW_GEN_NO_SOFTMAX", "0"))  # 1: MFMA + LDS + DMA skeleton only (garbage results)
KA = int(os.environ.get("PW_GEN_KA", "2"))                 # K fragments read this many steps ahead (ring of 4: <= 3)
LSUM_VALU = int(os.environ.get("PW_GEN_LSUM_VALU", "1"))   # 1: row sums by 64 VALU adds per tile (else by 8 extra MFMAs against a ones fragment)
VA = int(os.environ.get("PW_GEN_VA", "2"))
RK = int(os.environ.get("PW_GEN_RK", "4"))                 # registers rings of the K / V^T fragments (fragments): > steps ahead
RV = int(os.environ.get("PW_GEN_RV", "4"))                 # V^T fragments read this many steps ahead (ring of 4: <= 3)

# ---- the softmax stream of one (block, query block): 16 slots + a tail -----------------------------------------
# slots 0, 1: the lane's maximum over its 16 raw scores; slot 2: jump test (PW_JUMP: one compare + a wave-uniform
# branch; the rare slow path exchanges the maxima, moves the row's reference maximum and sets alpha);
# element e: F (t = fma(s, c2, -m)) one slot before X (p = exp2 t), A (row sum) and C (pack of a finished pair) one
# slot after.  X per slot 3..15: 1 1 1 2 1 1 1 2 1 1 1 2 1.
X_SLOTS = {}
_e = 0
for _s, _n in zip(range(3, 16), [1, 1, 1, 2, 1, 1, 1, 2, 1, 1, 1, 2, 1]):
    X_SLOTS[_s] = list(range(_e, _e + _n))
    _e += _n
assert _e == 16
F_SLOTS = {s - 1: es for s, es in X_SLOTS.items()}          # slot 2 .. 14
A_SLOTS = {s + 1: es for s, es in X_SLOTS.items()}          # slot 4 .. 16 (16 = tail)
C_SLOTS = {}
for s, es in X_SLOTS.items():
    for e in es:
        if e & 1:
            C_SLOTS.setdefault(s + 1, []).append(e >> 1)


def stream_slot(blk, qb, slot):
    """C++ statements of one slot of the stream of block `blk` (0 / 1), query block qb.  Names: S = s0 / s1,
    pk = pk0 / pk1; m[qb] is the row's reference maximum (one per query block: only one stream is in its F stage at
    any time), alpha0 / alpha1 the rescale a jump of block 0 / 1 leaves for the next safe point."""
    S = f"s{blk}[{qb}]"
    PK = f"pk{blk}[{qb}]"
    ma, mb = f"ma{blk}[{qb}]", f"mb{blk}[{qb}]"
    mref = f"mref[{qb}]"
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
        out += [f"PW_JUMP({blk}, {qb});"]
    if slot in F_SLOTS:
        launder += [mref]
        for e in F_SLOTS[slot]:
            out.append(f"{tv(e)} = __builtin_fmaf({S}[{e}], c2r, -{mref});")
            anchors.append(tv(e))
    if slot in X_SLOTS:
        for e in X_SLOTS[slot]:
            if tv(e) not in launder:
                launder.append(tv(e))
            out.append(f"{S}[{e}] = fast_exp2({tv(e)});")
            anchors.append(f"{S}[{e}]")
    if LSUM_VALU and slot in A_SLOTS:
        for e in A_SLOTS[slot]:
            acc = psa if (e & 1) == 0 else psb
            out.append(f"{acc} = {S}[{e}];" if e < 2 else f"{acc} += {S}[{e}];")
            if acc not in anchors:
                anchors.append(acc)
    if slot in C_SLOTS:
        for p in C_SLOTS[slot]:
            out.append(f"{PK}[{p >> 2}][{p & 3}] = pack2<T>({S}[{2 * p}], {S}[{2 * p + 1}]);")
            anchors.append(f"{PK}[{p >> 2}][{p & 3}]")
    if NO_SOFTMAX:
        return []
    pre = [f'asm volatile("" : "+v"({x}));' for x in launder]
    if slot == 2:   # the jump test first (it may move mref), then the launder + F
        lines = [out[0]] + pre + out[1:]
    else:
        lines = pre + out
    if anchors:
        lines.append('asm volatile("" :: ' + ", ".join(f'"v"({x})' for x in anchors) + ");")
    return lines


def stream_tail(blk, qb):
    """slot 16: the last row-sum add and the last pack, then the fold of the block into l."""
    if NO_SOFTMAX:
        return []
    lines = stream_slot(blk, qb, 16)
    if LSUM_VALU:
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
        if not NO_FENCE:
            add("    PW_FENCE();")
        add("  }")

    # DMA plan: the byte offset of row j (a ds_read_b64 from the offset table) at (group j+1, step 1, qb 1); its K piece
    # at step 3, its V piece at step 6
    def dma_post(group, i, qb):
        j = group - 1
        if qb != 1:
            return []
        if DMA_EARLY:
            if group != 1:
                return []
            pieces = [f"PW_DMA({i // 2}, {i % 2});"]
            return ([f"PW_ROW({i // 2});"] if i % 2 == 0 else []) + pieces
        if i == 1:
            return [f"PW_ROW({j});"]
        if i == 3:
            return [f"PW_DMA({j}, 0);"]
        if i == 6:
            return [f"PW_DMA({j}, 1);"]
        return []

    # ---------------- G1: QK^T(b0) | stream (b1, t-1) slots 8-15
    add("  PW_STAMP(0);   // everything since the end of the previous tile's G4: barrier wait, table step")
    add("  // ======== G1: QK^T(b0, t) | stream (b1, t-1) slots 8..15")
    add("  " + " ".join(f"kf[{j}] = PW_LDK(0, {j});" for j in range(KA)))
    for i in range(8):
        for qb in range(2):
            pre = []
            if qb == 0:
                nxt = i + KA
                pre.append(f"kf[{nxt % RK}] = PW_LDK({nxt // 8}, {nxt % 8});")
            mf = f"PW_QK({'true' if i == 0 else 'false'}, s0[{qb}], kf[{i % RK}], {qb}, {i});"
            gap(f"G1 step {i} qb {qb}", pre, mf, stream_slot(1, qb, 8 + i), dma_post(1, i, qb))
    for qb in range(2):
        for x in stream_tail(1, qb):
            add("  " + x)
    add("  PW_FENCE();")
    # ---------------- G2: QK^T(b1) | stream (b0, t) slots 0-7
    add("  PW_STAMP(1);")
    add("  // ======== G2: QK^T(b1, t) | stream (b0, t) slots 0..7")
    for i in range(8):
        for qb in range(2):
            pre = []
            if qb == 0 and i + KA < 8:
                nxt = 8 + i + KA
                pre.append(f"kf[{nxt % RK}] = PW_LDK(1, {nxt % 8});")
            if qb == 0 and i >= 8 - VA:   # the first V^T fragments of PV(b1, t-1): tile t-1, k-step 2, db 0 ..
                pre.append(f"vfa[{(i - (8 - VA)) % RV}] = PW_LDVP(2, {i - (8 - VA)});")
            mf = f"PW_QK({'true' if i == 0 else 'false'}, s1[{qb}], kf[{(8 + i) % RK}], {qb}, {i});"
            gap(f"G2 step {i} qb {qb}", pre, mf, stream_slot(0, qb, i), dma_post(2, i, qb))
    add("  PW_STAMP(2);")
    if not NO_SOFTMAX:
        add("  PW_RESCALE(alpha1, jump1);   // a jump of (b1, t-1): O^T moves to the new reference before PV(b1, t-1)")
    # ---------------- G3: PV(b1, t-1) from vfc | stream (b0, t) slots 8-15 | V(t) k-steps 0,1 -> vfa, vfb; ka toggle
    add("  // ======== G3: PV(b1, t-1) | stream (b0, t) slots 8..15")
    for g2 in range(8):
        for qb in range(2):
            pre = []
            if qb == 0:
                f2 = g2 + VA
                if f2 < 8:    # PV(b1, t-1) fragment f2: tile t-1, k-step 2 + f2 / 4, db f2 % 4 (VA steps ahead)
                    pre.append(f"vfa[{f2 % RV}] = PW_LDVP({2 + f2 // 4}, {f2 % 4});")
                else:         # the first V^T fragments of tile t for G4 (k-step 0, db 0 ..)
                    pre.append(f"vfa[{f2 % RV}] = PW_LDV(0, {f2 - 8});")
            mf = f"PW_PV(vfa[{g2 % RV}], pk1[{qb}][{g2 // 4}], {qb}, {g2 % 4});"
            post = dma_post(3, g2, qb)
            if qb == 0 and g2 < 4:   # the K addresses move to the other ring slot (all K reads of tile t are done)
                post = post + [f"PW_TOGGLE(ka[{2 * g2}]); PW_TOGGLE(ka[{2 * g2 + 1}]);"]
            gap(f"G3 step {g2} qb {qb}", pre, mf, stream_slot(0, qb, 8 + g2), post)
        if not LSUM_VALU and g2 in (3, 7):   # row sums of the k-step just multiplied: ones x P^T (one MFMA per query block)
            for qb in range(2):
                gap(f"G3 row sums k-step {g2 // 4} qb {qb}", [], f"PW_LSUM(pk1[{qb}][{g2 // 4}], {qb});", [], [])
    for qb in range(2):
        for x in stream_tail(0, qb):
            add("  " + x)
    add("  PW_FENCE();")
    add("  PW_STAMP(3);")
    if not NO_SOFTMAX:
        add("  PW_RESCALE(alpha0, jump0);   // ... and a jump of (b0, t) before PV(b0, t)")
    # ---------------- G4: PV(b0, t) from vfa / vfb | stream (b1, t) slots 0-7 | V(t) k-steps 2,3 -> vfc; va toggle
    add("  // ======== G4: PV(b0, t) | stream (b1, t) slots 0..7")
    for g2 in range(8):
        for qb in range(2):
            pre = []
            if qb == 0 and g2 + VA < 8:   # PV fragment g2 + VA (k-step (g2 + VA) / 4, db (g2 + VA) % 4) into the ring of four
                f2 = g2 + VA
                pre.append(f"vfa[{(8 + f2) % RV}] = PW_LDV({f2 // 4}, {f2 % 4});")
            src = f"vfa[{(8 + g2) % RV}]"
            mf = f"PW_PV({src}, pk0[{qb}][{g2 // 4}], {qb}, {g2 % 4});"
            post = dma_post(4, g2, qb)
            if qb == 1 and g2 == 7:       # after the last V read of tile t
                post = post + ["PW_TOGGLE_V(0); PW_TOGGLE_V(1); PW_TOGGLE_V(2); PW_TOGGLE_V(3);",
                               "PW_TOGGLE_V(4); PW_TOGGLE_V(5); PW_TOGGLE_V(6); PW_TOGGLE_V(7);"]
            gap(f"G4 step {g2} qb {qb}", pre, mf, stream_slot(1, qb, g2), post)
        if not LSUM_VALU and g2 in (3, 7):
            for qb in range(2):
                gap(f"G4 row sums k-step {g2 // 4} qb {qb}", [], f"PW_LSUM(pk0[{qb}][{g2 // 4}], {qb});", [], [])
    add("  PW_STAMP(4);")
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
    L.append("  PW_RESCALE(alpha1, jump1);")
    for g2 in range(8):
        for qb in range(2):
            L.append(f"  PW_PV(PW_LDVP({2 + g2 // 4}, {g2 % 4}), pk1[{qb}][{g2 // 4}], {qb}, {g2 % 4});")
        if not LSUM_VALU and g2 in (3, 7):
            for qb in range(2):
                L.append(f"  PW_LSUM(pk1[{qb}][{g2 // 4}], {qb});")
    L.append("  PW_FENCE();")
    return "\n".join(L) + "\n"


HEADER = ("// GENERATED by tools/gen_extend_pw.py -- do not edit; the schedule lives in the generator.\n"
          "// Steady-state iteration of rx::extend_pw_kernel for one fully visible 64-token tile (see the generator's\n"
          "// docstring for the pipeline).  Included inside the run loop of csrc/rx_extend_pw.hip.\n")


def main():
    with open(OUT, "w") as f:
        f.write(HEADER + gen_body())
    if not TAG:
        with open(OUT_DRAIN, "w") as f:
            f.write(HEADER.replace("Steady-state iteration", "Drain of a run") + gen_drain())
    # issue-cost audit of the stream (cycles per slot: 4 per plain VALU, 8 per exp2)
    cost = []
    for sl in range(17):
        c = {0: 16, 1: 16, 2: 4}.get(sl, 0)
        c += 4 * len(F_SLOTS.get(sl, [])) + 8 * len(X_SLOTS.get(sl, [])) + 4 * len(A_SLOTS.get(sl, [])) + 4 * len(C_SLOTS.get(sl, []))
        cost.append(c)
    print("stream issue cycles per slot:", cost, "mean", sum(cost) / 16.0)


if __name__ == "__main__":
    main()
