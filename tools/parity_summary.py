#!/usr/bin/env python3
"""gpurun_out/parity_errors.jsonl (written by tests/parity_util.py during `pytest -m gpu`) -> one JSON summary:
per test function and dtype, the number of checks, the largest observed max-abs error and the bound it was held to.
    python tools/parity_summary.py [in.jsonl] [out.json]"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl")
dst = sys.argv[2] if len(sys.argv) > 2 else None
groups = collections.OrderedDict()
for line in open(src):
    r = json.loads(line)
    t = r["test"]
    fn = t.split("::")[-1].split("[")[0]
    params = t[t.index("[") + 1: -1] if "[" in t else ""
    dt = r.get("dtype") or ("fp16" if re.search(r"float16|fp16", params) else ("bf16" if re.search(r"bfloat16|bf16", params) else ""))
    key = (os.path.basename(t.split("::")[0]), fn, dt)
    g = groups.setdefault(key, {"checks": 0, "max_err": 0.0, "tol": 0.0, "max_err_over_bound": None, "worst": None,
                                "checks_with_strict_ratio": 0, "pass_at_floor_1e-3_1ulp_no_absw": 0,
                                "max_err_over_strict_bound": None})
    g["checks"] += 1
    g["tol"] = max(g["tol"], r["tol"])
    if "ratio" in r:  # element-wise bound (parity_util.check_out): worst |err| / bound over all elements
        g["max_err_over_bound"] = max(g["max_err_over_bound"] or 0.0, r["ratio"])
    if "ratio_floor_1e-3_no_absw" in r:  # the fp16 bar max(1e-3, ulps * ulp) without the |V| term, reported only
        g["checks_with_strict_ratio"] += 1
        g["pass_at_floor_1e-3_1ulp_no_absw"] += int(r["ratio_floor_1e-3_no_absw"] <= 1.0)
        g["max_err_over_strict_bound"] = max(g["max_err_over_strict_bound"] or 0.0, r["ratio_floor_1e-3_no_absw"])
    if r["err"] >= g["max_err"]:
        g["max_err"], g["worst"] = r["err"], (params + (" | " + r["tag"] if r.get("tag") else ""))[:160]
out = [{"file": k[0], "test": k[1], "dtype": k[2], **v} for k, v in groups.items()]
bf = [o for o in out if o["dtype"] == "bf16" and o["checks_with_strict_ratio"]]
doc = {"source": os.path.relpath(src, ROOT), "north_star_bar_fp16": 1e-3,
       "bf16_checks_at_1ulp_without_the_4e-3_floor": {
           "checks": sum(o["checks_with_strict_ratio"] for o in bf),
           "pass": sum(o["pass_at_floor_1e-3_1ulp_no_absw"] for o in bf),
           "bound": "max(1e-3, ulps * ulp_bf16(|want|)), no |V| term"},
       "fp16_checks_over_1e-3": [o for o in out if o["dtype"] == "fp16" and o["max_err"] > 1e-3], "groups": out}
text = json.dumps(doc, indent=1)
if dst:
    open(dst, "w").write(text + "\n")
for o in out:
    rb = "" if o["max_err_over_bound"] is None else f' err/bound={o["max_err_over_bound"]:.2f}'
    print(f'{o["file"]:30s} {o["test"]:52s} {o["dtype"]:5s} n={o["checks"]:4d} max_err={o["max_err"]:.3e} tol={o["tol"]:.1e}{rb}')
