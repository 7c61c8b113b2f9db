#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into committed summaries under profiles/:
   profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary
   profiles/<tag>_traffic.json          HBM bytes per Jacobi launch from the PMC passes
   (bench.py reads every profiles/*_traffic.json and takes the one whose kernel, launch shape and frame size match)
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half of the bytes of a wide
coalesced read stream, so it is doubled; WRITE_SIZE is taken as is.  Both counters are in KiB."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    r = glob.glob(pattern, recursive=True)
    return max(r, key=os.path.getmtime) if r else None  # newest run wins


def counter_per_dispatch(path, counter, kernel_substr, grid=None, wg=None):
    """Counter values of the dispatches of one kernel; grid / wg: only launches of that grid and workgroup size (a bench run
    also launches the same kernel on other frame sizes: the side figures)."""
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter or kernel_substr not in row.get("Kernel_Name", ""):
                continue
            if grid is not None and int(float(row.get("Grid_Size", 0))) != grid:
                continue
            if wg is not None and int(float(row.get("Workgroup_Size", 0))) != wg:
                continue
            vals.append(float(row["Counter_Value"]))
    return vals


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = find(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag))
    bench = json.load(open(os.path.join(src, "bench_trace.json")))
    shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(ROOT, "profiles", "%s_bench_under_rocprof.json" % tag))
    kern = bench["roofline"]["kernel"] + "<"  # "k_jacobi_strip<": not the first-launch variant k_jacobi_strip_deriv<
    out = {"tag": tag, "kernel": kern[:-1], "width": bench["config"]["width"], "height": bench["config"]["height"],
           "pairs": bench["config"]["pairs_per_gpu"], "fuse_steps": bench["config"]["fuse_steps"],
           "rows_per_lane_or_groups": bench["config"]["rows_per_lane_or_groups"], "threads": bench["config"]["threads"],
           "tiles_per_launch": bench["config"]["tiles_per_launch"], "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)"}
    fetch = find(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"))
    write = find(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"))
    if fetch and write:
        grid = out["tiles_per_launch"] * out["threads"]
        fv = counter_per_dispatch(fetch, "FETCH_SIZE", kern, grid, out["threads"])
        wv = counter_per_dispatch(write, "WRITE_SIZE", kern, grid, out["threads"])
        if fv and wv:
            # drop tail launches (fewer sweeps): keep the most common magnitude via the median
            fv.sort(); wv.sort()
            f_kib, w_kib = fv[len(fv) // 2], wv[len(wv) // 2]
            out.update({"fetch_size_kib_raw": f_kib, "write_size_kib": w_kib, "dispatches": len(fv),
                        "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0})
    with open(os.path.join(ROOT, "profiles", "%s_traffic.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    if stats:
        print(open(stats).read()[:1500])


if __name__ == "__main__":
    main()
