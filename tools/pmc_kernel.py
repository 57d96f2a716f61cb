#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs: mean per launch of every counter for kernels
whose name contains a substring.  python tools/pmc_kernel.py <dir> <substr>"""
import collections, csv, glob, json, os, sys

def main(src, sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
    out["_launches"] = {k: len(v) for k, v in acc.items()}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
