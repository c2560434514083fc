"""Device-memory leak check: repeated pair create / run / destroy and stand-alone stage calls; the free memory must not drift."""
import sys, os, numpy as np
sys.path.insert(0,'' + os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/piecewise-icp_amd'); sys.path.insert(0,'' + os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/tests')
import pwicp_amd as P, _data, torch
from pwicp_amd import synth
ctx=P.Context(0)
tgt,src,_=_data.pair(100000)
l1,n1=synth.grid_labels(tgt,10*_data.R); l2,n2=synth.grid_labels(src,10*_data.R)
def free(): 
    f,t=torch.cuda.mem_get_info(0); return f/2**20
f0=None
for i in range(120):
    p=P.Pair(ctx,tgt,l1,n1,src,l2,n2,_data.params()); r=p.run(); p.close()
    if i%3==0:
        a=ctx.preprocess(tgt,_data.R,14,5.0); nb=ctx.knn(src[:20000],45)
    if i==10: f0=free()
print("free MiB after 10: %.1f, after 120: %.1f"%(f0, free()))
