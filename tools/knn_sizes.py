"""Timing of the k-NN-45 graph (pwicp_knn: upload, grid, kernel, download) against the cloud size (run on the GPU box)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth
ctx = P.Context(0); r = 0.005
for n in (5000, 20000, 70000, 140000, 300000, 1000000):
    t, _ = synth.make_tile(n, r); t = (t - t.mean(0)).astype(np.float32)
    ctx.knn(t, 45, 2 * r)
    t0 = time.perf_counter(); nb = ctx.knn(t, 45, 2 * r); dt = time.perf_counter() - t0
    print("n=%7d  %.2f ms" % (n, dt * 1e3), flush=True)
