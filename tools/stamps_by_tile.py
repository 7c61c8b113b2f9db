#!/usr/bin/env python3
"""Per-tile view of the strip kernel's phase stamps (HSFLOW_DEBUG_STAMPS): which tiles have the long load phases /
totals?  Prints the mean load / sweeps / total cycles by tile column and by tile row for the default 1080p plan."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = tempfile.mktemp(prefix="hs_stamps_")
os.environ["HSFLOW_DEBUG_STAMPS"] = path
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth

W, H = 1920, 1080
ctx = hs.HSFlow(W, H, 1, own_stream=True)
A, B = synth.translating_pair(W, H, seed=1)
ctx.set_frames(A, B)
for _ in range(3):
    ctx.solve(lam=1.0, max_iter=100, term_type=hs.TERM_ITER)
if os.path.exists(path):
    os.remove(path)
info = ctx.solve(lam=1.0, max_iter=100, term_type=hs.TERM_ITER)
rows = np.array([[int(x) for x in l.split()] for l in open(path) if not l.startswith("#")], dtype=np.float64)
tx = -(-W // info["tile_w"])
load, sweeps, store, total, xcc, tile = rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4], rows[:, 6], rows[:, 7].astype(int)
bx, by = tile % tx, tile // tx
print("tiles %d (%d x %d), T %d" % (info["tiles"], tx, info["tiles"] // tx, info["fuse_steps"]))
print("by tile column: " + "  ".join("bx%d load %.0f sweeps %.0f total %.0f" % (c, load[bx == c].mean(), sweeps[bx == c].mean(), total[bx == c].mean()) for c in range(tx)))
for r in sorted(set(by)):
    m = by == r
    print("tile row %2d: load %6.0f  sweeps %6.0f  store %5.0f  total %6.0f   (max total %6.0f)" % (r, load[m].mean(), sweeps[m].mean(), store[m].mean(), total[m].mean(), total[m].max()))
print("by XCC: " + "  ".join("x%d n%d load %.0f total %.0f" % (x, (xcc.astype(int) & 15 == x).sum(), load[(xcc.astype(int) & 15) == x].mean(), total[(xcc.astype(int) & 15) == x].mean()) for x in range(8)))
slow = np.argsort(-total)[:12]
print("slowest: " + "  ".join("tile(%d,%d) load %.0f total %.0f" % (bx[i], by[i], load[i], total[i]) for i in slow))
ctx.close()
os.remove(path)
