#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/...) into the small summaries committed under
profiles/:  python profiles/summarize.py gpurun_out/prof r01
  <tag>_kernel_stats.csv   -- `rocprofv3 --kernel-trace --stats` per-kernel table (names shortened)
  <tag>_pmc_summary.json   -- per-kernel FETCH_SIZE / WRITE_SIZE means from the two --pmc passes and
                              the HBM traffic per launch with the gfx950 correction
                              (MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of a wide coalesced
                              read stream; counters are in KiB): traffic = (2*FETCH + WRITE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"<.*", "", name) if name.startswith("void at::") else name
    return name[:110]


def main(src, tag):
    out_dir = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        rows = list(csv.DictReader(open(ks[0])))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                            r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    pmc = {}
    for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, sub, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc.setdefault(k, {})[counter] = {"mean_KiB": sum(v) / len(v), "launches": len(v)}
    for k, d in pmc.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE"]["mean_KiB"] + d["WRITE_SIZE"]["mean_KiB"]) * 1024
    if pmc:
        with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                                 "--no-cpu-baseline --no-extend --steps 2 --warmup 1",
                       "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024  [gfx950: FETCH_SIZE counts 64 B per "
                                     "128-B request of a wide coalesced stream]",
                       "kernels": pmc}, f, indent=1)


def mla(src, tag):
    """FETCH_SIZE of the MLA decode kernels (tools/mla_bench.py under --pmc, 16-bit and fp8 latent rows) against the
    algorithmic bytes of the shape (bs 64 x ctx 8192 x 576 elements)."""
    out_dir = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for sub, name, row_bytes in (("mla16_pmc", "bf16_rows", 1152), ("mla8_pmc", "fp8_rows", 576)):
        files = glob.glob(os.path.join(src, sub, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == "FETCH_SIZE" and "decode_mla" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        alg = 64 * 8192 * row_bytes
        for k, v in acc.items():
            kib = sum(v) / len(v)
            res[name] = {"kernel": k, "launches": len(v), "FETCH_SIZE_mean_KiB": kib,
                         "hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)": 2 * kib * 1024,
                         "algorithmic_kv_bytes": alg, "read_over_algorithmic": 2 * kib * 1024 / alg}
    if res:
        with open(os.path.join(out_dir, f"{tag}_mla_pmc_summary.json"), "w") as f:
            json.dump({"source": "PS=64 rocprofv3 --pmc FETCH_SIZE -- python3 tools/mla_bench.py (FP8=1 for fp8 rows; PS = the page_size, 64 as in "
                                 "the bench leg since r04 -- r02 / r03 ran page_size 1); same "
                                 "gfx950 correction as the decode summary", "kernels": res}, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
    mla(sys.argv[1], sys.argv[2])
