#!/usr/bin/env python3
"""gpurun_out/sweep_counters_<tag>/ (tools/sweep_counters.sh) -> profiles/<tag>_sweep_top5_counters.csv: the five fastest
strip-kernel shapes of the sweep with their time and the HBM bytes per full-depth Jacobi launch from the PMC passes
(FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is; both counters in KiB)."""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def median_counter(d, counter):
    vals = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") == counter and "k_jacobi_strip<" in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    vals.sort()
    keep = [v for v in vals if v >= 0.5 * vals[-1]] if vals else []   # full-depth launches only (drop tails)
    return keep[len(keep) // 2] if keep else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02_4k"
    base = os.path.join(ROOT, "gpurun_out", "sweep_counters_" + tag)
    out = os.path.join(ROOT, "profiles", "%s_sweep_top5_counters.csv" % tag)
    src_csv = os.path.join(ROOT, "gpurun_out", "sweep_%s.csv" % tag)
    if os.path.exists(src_csv):
        shutil.copy(src_csv, os.path.join(ROOT, "profiles", "%s_sweep.csv" % tag))
    with open(out, "w") as f:
        f.write("name,T,rows_per_lane,threads,median_ms,tiles,tile_w,tile_h,fetch_kib_raw,write_kib,hbm_bytes_per_launch,algorithmic_bytes_per_launch\n")
        for line in open(os.path.join(base, "top5.txt")):
            name, T, R, NT, ms, tiles, tw, th = line.split()
            fk = median_counter(os.path.join(base, name, "fetch"), "FETCH_SIZE")
            wk = median_counter(os.path.join(base, name, "write"), "WRITE_SIZE")
            hbm = (2.0 * fk + wk) * 1024.0 if fk is not None and wk is not None else float("nan")
            px = {"r02_4k": 3840 * 2160}.get(tag, 3840 * 2160)
            f.write("%s,%s,%s,%s,%s,%s,%s,%s,%s,%s,%.0f,%.0f\n" % (name, T, R, NT, ms, tiles, tw, th, fk, wk, hbm, 28.0 * px * int(T)))
    print(open(out).read())


if __name__ == "__main__":
    main()
