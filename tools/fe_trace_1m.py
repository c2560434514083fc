import os, sys, time
sys.path.insert(0,'/root/repo/piecewise-icp_amd'); sys.path.insert(0,'/root/repo/tests')
import pwicp_amd as P, _data
ctx=P.Context(0)
tgt, src, _ = _data.pair(1000000)
ctx.frontend_segment(tgt, 10*_data.R, 45, _data.R)
os.environ["PWICP_TRACE"]="1"
t=time.time(); ctx.frontend_segment(tgt, 10*_data.R, 45, _data.R); print("total %.1f ms" % (1e3*(time.time()-t)))
