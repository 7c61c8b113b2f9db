import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import opticalflowhs_amd as hs
from opticalflowhs_amd import synth
W,H=1920,1080
ts=torch.cuda.Stream(); torch.cuda.set_stream(ts)
ctx=hs.HSFlow(W,H,1,stream=ts.cuda_stream)
A,B=synth.translating_pair(W,H,seed=1); ctx.set_frames(A,B)
p=ctx.make_params(lam=1.0,max_iter=100,term_type=3,epsilon=float(np.float32(1e-6)),use_graph=True)
for _ in range(5): ctx.solve_async(p)
torch.cuda.synchronize()
out=[]
t00=time.perf_counter()
for b in range(150):
    t0=time.perf_counter()
    for _ in range(20): ctx.solve_async(p)
    torch.cuda.synchronize()
    out.append((time.perf_counter()-t00, (time.perf_counter()-t0)/20*1e3))
print(' '.join('%.0fms:%.4f'%(a*1e3,b) for a,b in out[:60]))
print('tail median', np.median([b for a,b in out[100:]]))
