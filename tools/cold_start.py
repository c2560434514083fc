#!/usr/bin/env python3
"""Where the first call of a process spends its time: runtime start, context, first kernels (code object load), the first front
end of a stream (work space, pinned staging) against the second.  Run on the GPU box."""
import os, sys, time
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
t00 = time.perf_counter()
import pwicp_amd as P
from pwicp_amd import synth
def lap(what, t0):
    t1 = time.perf_counter(); print("%-58s %8.1f ms" % (what, 1e3 * (t1 - t0)), flush=True); return t1
t = lap("import pwicp_amd (dlopen of libpwicp.so)", t00)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 140000
r = 0.005
cloud, _ = synth.make_tile(n, r); cloud = (cloud - cloud.mean(0)).astype(np.float32)
t = time.perf_counter()
nd = P.device_count() if hasattr(P, "device_count") else None
t = lap("device count (HIP runtime start)", t)
ctx = P.Context(0); t = lap("first context", t)
ctx2 = P.Context(0); t = lap("second context", t)
p = ctx.preprocess(cloud, r, 14, 5.0); t = lap("first preprocess (VoxelGrid + SOR: code object, buffers)", t)
p = ctx.preprocess(cloud, r, 14, 5.0); t = lap("second preprocess", t)
l, ns = ctx.frontend_segment(cloud, 10 * r, 45, r); t = lap("first front end on context 1", t)
l, ns = ctx.frontend_segment(cloud, 10 * r, 45, r); t = lap("second front end on context 1", t)
l, ns = ctx2.frontend_segment(cloud, 10 * r, 45, r); t = lap("first front end on context 2 (work space, pinned staging)", t)
l, ns = ctx2.frontend_segment(cloud, 10 * r, 45, r); t = lap("second front end on context 2", t)
lap("total", t00)
