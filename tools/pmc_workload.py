#!/usr/bin/env python3
"""What tools/pmc_by_kernel.sh runs under rocprofv3 --pmc: eager (no hipGraph) solves of one resident pair with ITER (the
<.., 0, ..> kernels) and asynchronous ITER|EPS (the witness kernels <.., 2, ..>), so that the counter rows carry the
kernel names.   usage: pmc_workload.py W H ITERS [REPS]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import opticalflowhs_amd as hs  # noqa: E402
from opticalflowhs_amd import synth  # noqa: E402

W, H, it = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
A, B = synth.translating_pair(W, H, seed=1)
with hs.HSFlow(W, H, 1, device=0, own_stream=True) as ctx:
    ctx.set_frames(A, B)
    for tt in (hs.TERM_ITER, hs.TERM_ITER | hs.TERM_EPS):
        p = ctx.make_params(lam=1.0, max_iter=it, term_type=tt, epsilon=float(np.float32(1e-6)))
        for _ in range(reps):
            ctx.solve_async(p)
            ctx.synchronize()   # (settles each ITER|EPS solve: no takeover, every solve runs its witness launches)
    i = ctx.info()
    print("plan: T=%d R=%d threads=%d tiles=%d" % (i["fuse_steps"], i["groups_per_thread"], i["threads"], i["tiles"]))
