"""Builds libhsflow.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
opticalflowhs_amd/libhsflow.so travels to the GPU box with the source snapshot.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhsflow.so")
SOURCES = ["hsflow.hip", "pair_pipeline.cpp", "multi_gpu.cpp"]
DEPS = ["hsflow.hip", "pair_pipeline.cpp", "multi_gpu.cpp", "hs_context.hip.h", "hs_plan_launch.hip.h", "hs_runtime.hip.h", "hs_solve.hip.h", "hs_kernels.hip.h", "hs_kernels_strip.hip.h", "hs_kernels_pre.hip.h", "hs_kernels_classic.hip.h", "hs_kernels_classic_strip.hip.h", os.path.join("..", "..", "include", "hsflow.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine",
         "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS + ["hs_kernels_classic.hip.h", "hs_kernels_pre.hip.h"]:
        p = os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return os.path.getmtime(os.path.abspath(__file__)) > t


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    log = r.stdout + r.stderr
    with open(os.path.join(HERE, "build_resource_usage.log"), "w") as f:
        f.write(log)
    if r.returncode != 0:
        sys.stderr.write(log[-8000:])
        raise RuntimeError("hipcc failed (%d)" % r.returncode)
    if verbose:
        print(summary(log))
    return LIB


def summary(log=None):
    if log is None:
        with open(os.path.join(HERE, "build_resource_usage.log")) as f:
            log = f.read()
    out = []
    name = None
    vals = {}
    for line in log.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            if name:
                out.append((name, vals))
            name, vals = m.group(1), {}
            continue
        m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name:
            vals[m.group(1)] = int(m.group(2))
    if name:
        out.append((name, vals))
    lines = []
    for n, v in out:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
        d = re.sub(r"\(.*", "", d)
        lines.append("%-60s vgpr=%3d spill=%d occ=%d scratch=%d" % (
            d[:60], v.get("VGPRs", -1), v.get("VGPRs Spill", -1), v.get("Occupancy [waves/SIMD]", -1),
            v.get("ScratchSize [bytes/lane]", -1)))
    return "\n".join(lines)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
