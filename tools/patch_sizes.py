import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P, _data
ctx = P.Context(0)
tgt, src, _ = _data.pair(1000000)
l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
Ps = ctx.selectPatches(src, l2, n2)
off = np.asarray(Ps["off"]); sz = np.diff(off)
print("patches", len(sz), "mean %.1f" % sz.mean(), "percentiles 50/90/99/max:", np.percentile(sz, [50, 90, 99]), sz.max())
# per wave of 8 consecutive patches: max size
m = len(sz) // 8 * 8
w = sz[:m].reshape(-1, 8).max(1)
print("max per wave (8 patches): mean %.1f, percentiles 50/90/99/max:" % w.mean(), np.percentile(w, [50, 90, 99]), w.max())
b = sz[:len(sz) // 32 * 32].reshape(-1, 32).max(1)
print("max per block (32 patches): mean %.1f max %d" % (b.mean(), b.max()))
