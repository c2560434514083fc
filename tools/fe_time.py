"""Wall time of the device front end of one cloud, host buffer in -> labels out (median / min of REPS runs after a warm-up):
python tools/fe_time.py [real <epoch> | <points>]   (env: REPS; run on the GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P, _data
ctx = P.Context(0)
a = sys.argv[1:]
if a and a[0] == "real":
    from pwicp_amd.pcd import read_pcd
    cl = ctx.preprocess(read_pcd(os.path.join(ROOT, "tests/golden/inputs/Epoch_%03d.pcd" % int(a[1]))), 0.005, 14, 5.0)
    cl = (cl - cl.mean(axis=0)).astype(np.float32); sv, sp = 0.05, 0.005
else:
    cl = _data.pair(int(a[0]) if a else 1000000, epoch=1)[1]; sv, sp = 10 * _data.R, _data.R
reps = int(os.environ.get("REPS", "9"))
lab0 = None
ts = []
for i in range(reps + 2):
    t = time.perf_counter(); lab, nsv = ctx.frontend_segment(cl, sv, 45, sp); dt = time.perf_counter() - t
    if i >= 2: ts.append(1e3 * dt)
    if lab0 is None: lab0 = lab.copy()
    assert np.array_equal(lab, lab0)
ts.sort()
print("%d points, %d supervoxels: median %.2f ms, min %.2f ms over %d runs" % (len(cl), nsv, ts[len(ts) // 2], ts[0], reps))
