"""k-NN graph on the device: timing against the cloud size, and the fall-back kernel (k > 48: lists in global memory) against
scipy (run on the GPU box)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth
from scipy.spatial import cKDTree
ctx = P.Context(0); r = 0.005
for n in (5000, 20000, 140000, 1000000):
    t, _ = synth.make_tile(n, r); t = (t - t.mean(0)).astype(np.float32)
    ctx.knn(t, 45, 2 * r)
    t0 = time.perf_counter(); nb = ctx.knn(t, 45, 2 * r); dt = time.perf_counter() - t0
    print("n=%7d  k=45: %.2f ms (upload, grid, kernel, download)" % (n, dt * 1e3), flush=True)
t, _ = synth.make_tile(30000, r); t = (t - t.mean(0)).astype(np.float32)
for k in (45, 60):
    nb = ctx.knn(t, k, 2 * r)
    d, ii = cKDTree(t.astype(np.float64)).query(t.astype(np.float64), k=k)
    dn = np.linalg.norm(t[nb].astype(np.float64) - t[:, None, :].astype(np.float64), axis=2)
    print("k=%d: distances equal to scipy's: %s, index mismatches (ties) %.2e" % (k, np.allclose(dn, d, rtol=0, atol=1e-12), (nb != ii).mean()))
