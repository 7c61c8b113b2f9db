#!/usr/bin/env python3
"""Per-sweep durations inside one strip-kernel launch (HSFLOW_DEBUG_STAMPS + HSFLOW_DEBUG_STAMPS_SWEEPS): median over the
workgroups of the cycles each sweep of the last launch took.  Diagnosis only: needs a library built with
-DHS_SWEEP_STAMPS=1 (tools/diag_build.sh 0; HSFLOW_LIB_PATH=tools/bin/libhsflow_diag0.so)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = tempfile.mktemp(prefix="hs_stamps_")
os.environ["HSFLOW_DEBUG_STAMPS"] = path
os.environ["HSFLOW_DEBUG_STAMPS_SWEEPS"] = "1"
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth

W, H = 1920, 1080
ctx = hs.HSFlow(W, H, 1, own_stream=True)
A, B = synth.translating_pair(W, H, seed=1)
ctx.set_frames(A, B)
for cfg in (sys.argv[1] if len(sys.argv) > 1 else "20:5:1024").split(";"):
    T, R, nt = [int(x) for x in cfg.split(":")]
    kw = dict(lam=1.0, max_iter=4 * T, term_type=hs.TERM_ITER, kernel=hs.KERNEL_STRIP, fuse_steps=T, strip_rows=R, threads=nt)
    for _ in range(3):
        ctx.solve(**kw)
    if os.path.exists(path):
        os.remove(path)
    ctx.solve(**kw)
    sw = np.array([[int(x) for x in l.split()[2:]] for l in open(path) if l.startswith("S ")], dtype=np.float64)
    d = np.diff(np.concatenate([np.zeros((sw.shape[0], 1)), sw], axis=1), axis=1)
    print("T=%d R=%d threads=%d: cycles per sweep (median over %d workgroups); sweep 0 includes the coefficient set-up" % (T, R, nt, sw.shape[0]))
    print("  " + " ".join("%d" % x for x in np.median(d, axis=0)))
    print("  sum %.0f, mean of sweeps 1.. %.0f" % (np.median(d, axis=0).sum(), np.median(d, axis=0)[1:].mean()))
ctx.close()
