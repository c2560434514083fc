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
# the front end keeps its work buffers with the context (grow-only): repeated calls must not drift, closing the context frees them
f1 = None
for i in range(24):
    n = 60000 + 20000 * (i % 4)
    ctx.frontend_segment(tgt[:n], 10 * _data.R, 45, _data.R)
    if i == 7: f1 = free()
print("front end: free MiB after 8 calls: %.1f, after 24: %.1f" % (f1, free()))
ctx.close()
print("after closing the context: %.1f MiB free" % free())
