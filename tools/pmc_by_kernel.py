#!/usr/bin/env python3
"""Per-kernel medians of FETCH_SIZE / WRITE_SIZE from the two rocprofv3 PMC passes of tools/pmc_by_kernel.sh.
FETCH_SIZE is doubled (gfx950: the counter reports half of the bytes of a wide coalesced read stream, MI355X_MICROARCH.md);
both counters are in KiB.  Prints JSON."""
import collections
import csv
import glob
import json
import os
import re
import statistics
import sys


def rows(d, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter:
                    out[short(r.get("Kernel_Name", ""))].append(float(r["Counter_Value"]))
    return out


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    d, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    f, w = rows(os.path.join(d, "fetch"), "FETCH_SIZE"), rows(os.path.join(d, "write"), "WRITE_SIZE")
    px = W * H
    out = {"width": W, "height": H, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x 2 (gfx950); medians per kernel",
           "compulsory_write_bytes_per_jacobi_launch": 8 * px, "compulsory_read_bytes_per_jacobi_launch": 12 * px, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fk, wk = f.get(k, []), w.get(k, [])
        e = {"launches": max(len(fk), len(wk))}
        if fk:
            e["hbm_read_bytes"] = 2.0 * statistics.median(fk) * 1024.0
        if wk:
            e["hbm_write_bytes"] = statistics.median(wk) * 1024.0
            e["write_over_compulsory"] = e["hbm_write_bytes"] / (8.0 * px)
        if fk and wk:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
