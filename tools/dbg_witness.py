import sys, numpy as np
sys.path.insert(0, '/root/repo')
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
W, H, it = 700, 300, 57
A, B = synth.translating_pair(W, H, seed=21, dx=1.25, dy=-0.75)
eps6 = float(np.float32(1e-6))
with hs.HSFlow(W, H, own_stream=True) as ctx:
    ctx.set_frames(A, B)
    for T in (1, 2, 3, 4, 5, 6, 8, 19):
        for R in (0, 1, 2, 3, 4, 5, 6):
            for n in (T, 2 * T, 57):
                try:
                    r = ctx.solve(lam=0.7, max_iter=n, epsilon=eps6, term_type=3, kernel=hs.KERNEL_STRIP, fuse_steps=T, strip_rows=R)
                except hs.HsflowError as e:
                    continue
                if r["eps_rerun"]:
                    print("T", T, "R", R, "n", n, "rerun", r["eps_rerun"], "threads", r["threads"], "rows", r["groups_per_thread"], "tiles", r["tiles"], "tile", r["tile_w"], r["tile_h"])
print("done")
