#!/usr/bin/env python3
"""Condense gpurun_out/pmc (tools/pmc_round.sh) into profiles/<tag>_*_sq_counters.json: mean per launch of every counter
of the named kernel + the derived fractions DESIGN.md quotes.  python tools/pmc_round_summary.py [src] [dst] [tag]"""
import collections, csv, glob, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
tag = sys.argv[3] if len(sys.argv) > 3 else "r04"


def counters(prefix, sub):
    acc = collections.defaultdict(list)
    names = set()
    for f in glob.glob(os.path.join(src, prefix + "_p*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"])
    return {k: sum(v) / len(v) for k, v in sorted(acc.items())}, sorted(names), {k: len(v) for k, v in acc.items()}


def derived(c):
    d = {}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        d["kernel_cycles_per_launch(GRBM_GUI_ACTIVE/8)"] = cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024 SIMDs)"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
    if c.get("SQ_INSTS_MFMA"):
        d["valu_per_mfma"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_INSTS_MFMA"]
        d["salu_per_mfma"] = c.get("SQ_INSTS_SALU", 0) / c["SQ_INSTS_MFMA"]
        d["lds_insts_per_mfma"] = c.get("SQ_INSTS_LDS", 0) / c["SQ_INSTS_MFMA"]
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in c:
                d[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_WAVE_CYCLES"):  # issue classes: cycles a wave had an instruction of the class in flight / wave cycles
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
                  "SQ_ACTIVE_INST_ANY"):
            if k in c:
                d[k.lower() + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU_TRANS_F32" in c:
        d["trans_per_mfma"] = c["SQ_INSTS_VALU_TRANS_F32"] / c["SQ_INSTS_MFMA"]
        d["vmem_rd_per_mfma"] = c.get("SQ_INSTS_VMEM_RD", 0) / c["SQ_INSTS_MFMA"]
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
        d["valu_coexec_frac_of_mfma_busy"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    return d


JOBS = [
    (tag + "_extend32_sq_counters.json", "ext32", "extend_mfma32_kernel<rx::BF16, long, false, false, 8, false, true, 4>", "python3 bench.py --extend-only (config-3 chunk, D = 128, the dispatched eight-wave packed PLAIN instance only: the forms leg runs other instances)"),
    (tag + "_extend_d256_sq_counters.json", "dims", "extend_d256_kernel<rx::BF16, 256", "DIMS=256x256,64x64,192x128 python3 tools/extend_dims.py (config-3 chunk at D = 256)"),
    (tag + "_extend_d64_sq_counters.json", "dims", "extend_d256_kernel<rx::BF16, 64", "the same run, D = 64 (config 0's head dim; on the D = 256 kernel's template)"),
    (tag + "_mla_decode_fp8_sq_counters.json", "mla8", "decode_mla8", "PS=64 FP8=1 python3 tools/mla_bench.py (config-5 shard shape, fp8 rows)"),
    (tag + "_mla_decode_bf16_sq_counters.json", "mla16", "decode_mla_kernel", "PS=64 python3 tools/mla_bench.py (16-bit rows)"),
]
for out, prefix, sub, what in JOBS:
    c, names, n = counters(prefix, sub)
    if not c:
        print("no data for", out)
        continue
    json.dump({"source": "bash tools/pmc_round.sh: two rocprofv3 --pmc passes of `" + what + "`, means per launch", "kernels": names,
               "launches_per_counter": n, "counters": c, "derived": derived(c)}, open(os.path.join(dst, out), "w"), indent=1)
    print(out, json.dumps(derived(c)))
