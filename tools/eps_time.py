import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
W,H=1920,1080
A,B=synth.translating_pair(W,H,seed=1)
ctx=hs.HSFlow(W,H,own_stream=True); ctx.set_frames(A,B)
for tt,name in ((1,'ITER'),(3,'ITER|EPS')):
    for prof in (True,False):
        p=ctx.make_params(lam=1.0,max_iter=100,term_type=tt,profile=prof)
        for _ in range(5): info=ctx.solve(p)
        t0=time.perf_counter()
        for _ in range(50): info=ctx.solve(p)
        dt=(time.perf_counter()-t0)/50*1e3
        print(name,'profile',prof,'wall %.3f ms'%dt, {k:info[k] for k in ('fuse_steps','groups_per_thread','threads','jacobi_launches','deriv_ms','jacobi_ms','solve_ms','iterations_done','last_eps')})
