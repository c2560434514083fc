"""Times the setup stages either side of the loop on a synthetic 1 M-point epoch: preprocessing (host vs GPU) and the
supervoxel front end (host vs GPU k-NN).  Run on the GPU box: python tools/time_setup_stages.py [n_points]"""
import os, sys, time
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
r = 0.005
ctx = P.Context(0)
t, _ = synth.make_tile(n, r)
t = t.astype(np.float32)
for name, fn in (("preprocess gpu", lambda: ctx.preprocess(t, r, 14, 5.0)), ("preprocess gpu (2nd)", lambda: ctx.preprocess(t, r, 14, 5.0)),
                 ("preprocess host", lambda: P.preprocess(t, r, 14, 5.0))):
    t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
    print("%-22s %8.3f s  -> %d points" % (name, dt, len(out)))
a = ctx.preprocess(t, r, 14, 5.0); b = P.preprocess(t, r, 14, 5.0)
print("identical:", a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)))
for name, fn in (("front end gpu-knn", lambda: ctx.frontend_segment(a, 10 * r, 45, r)), ("front end host", lambda: P.frontend_segment(a, 10 * r))):
    t0 = time.perf_counter(); lab, nsv = fn(); dt = time.perf_counter() - t0
    print("%-22s %8.3f s  -> %d supervoxels" % (name, dt, nsv))
