import sys, time, numpy as np
sys.path.insert(0,'/root/repo/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth
ctx=P.Context(0); r=0.005
t,_=synth.make_tile(1000000,r); t=t.astype(np.float32)
for i in range(3):
    t0=time.perf_counter(); nb=ctx.knn(t,45,2*r); print("knn %.1f ms"%((time.perf_counter()-t0)*1e3))
