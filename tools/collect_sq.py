#!/usr/bin/env python3
"""Summarises gpurun_out/prof_sq_<tag>/ (tools/profile_sq.sh) into profiles/<tag>_sq_counters.json:
per-kernel means of the SQ counters over the full-depth launches, plus the ratios quoted in DESIGN.md."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    base = os.path.join(ROOT, "gpurun_out", "prof_sq_" + tag)
    vals = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per dispatch
    for path in glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                k = "k_jacobi_strip" if "k_jacobi_strip" in k else ("k_deriv_cv" if "k_deriv_cv" in k else None)
                if k:
                    vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {"tag": tag, "source": "rocprofv3 --pmc (two passes, SQ block), tools/profile_sq.sh", "kernels": {}}
    for k, cs in vals.items():
        d = {}
        for name, v in cs.items():
            v = sorted(v)
            keep = [x for x in v if x >= 0.5 * v[-1]] if k == "k_jacobi_strip" else v  # drop shallower tail launches
            d[name] = sum(keep) / len(keep)
        r = {}
        wc = d.get("SQ_WAVE_CYCLES")
        if wc:
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if name in d:
                    r[name + "/SQ_WAVE_CYCLES"] = round(d[name] / wc, 4)
        if d.get("SQ_LDS_IDX_ACTIVE"):
            r["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
        if d.get("SQ_WAVES"):
            for name in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if name in d:
                    r[name + " per wave"] = round(d[name] / d["SQ_WAVES"], 1)
        out["kernels"][k] = {"mean_per_dispatch": {n: round(x, 1) for n, x in sorted(d.items())}, "ratios": r}
    path = os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
