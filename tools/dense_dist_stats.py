import sys, numpy as np
sys.path.insert(0,'/root/repo/piecewise-icp_amd'); sys.path.insert(0,'/root/repo/tests')
import pwicp_amd as P, _data
ctx=P.Context(0)
for epoch in (1,2,3):
    tgt, src, _ = _data.pair(1000000, epoch=epoch)
    idx, d2 = ctx.determineCorrespondences(tgt, src)
    d = np.sqrt(d2)/_data.R
    st = d < 10
    ds = d[st]
    print("epoch", epoch, "n", len(ds), "mean %.2f r  median %.2f  p75 %.2f  p90 %.2f  p99 %.2f" % (ds.mean(), np.median(ds), np.percentile(ds,75), np.percentile(ds,90), np.percentile(ds,99)),
          "frac>1.5r %.3f >2r %.3f >2.5r %.3f >3r %.3f >4r %.3f" % tuple((ds>t).mean() for t in (1.5,2,2.5,3,4)))
