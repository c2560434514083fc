"""The registration loop on one of the reference's own pairs (Epoch_<e> -> Epoch_001, fixtures): per-run wall time and the result's
iteration counts; run under rocprofv3 --kernel-trace for the timeline (tools/trace_last_step.py).  python tools/real_pair_loop.py [e] [runs]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
e = int(sys.argv[1]) if len(sys.argv) > 1 else 2
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = os.path.join(ROOT, "tests", "golden", "inputs")
ctx = P.Context(0)
p1 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 5.0)
p2 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_%03d.pcd" % e)), 0.005, 14, 5.0)
cen = p1[:, :3].mean(0); p1[:, :3] -= cen; p2[:, :3] -= cen
l1, n1 = ctx.frontend_segment(p1, 0.05, 45, 0.005)
l2, n2 = ctx.frontend_segment(p2, 0.05, 45, 0.005)
pair = P.Pair(ctx, p1, l1, n1, p2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
pair.set_profiling(0)
ts = []
for _ in range(runs):
    pair.reset()
    t0 = time.perf_counter(); r = pair.run(); ts.append(time.perf_counter() - t0)
no = r.n_outer
print("epoch %d: %d / %d points, patches %s, outer %d, inner %s, stable %s" % (e, len(p1), len(p2), pair.num_patches(), no, list(r.n_inner[:no]), list(r.n_stable[:no])))
print("dense: kbar %.1f, queries %d, launches %d, d75 %s, DT %s" % (r.dense_kbar, r.n_corr_dense, r.n_dense_nn_launches, [round(float(x), 5) for x in r.d75[:no]], [round(float(x), 5) for x in r.DTseries[:no + 1]]))
print("maxBB %s  (DTmin 0.004)" % [round(float(x), 5) for x in r.maxBB[:no]])
print("loop wall: median %.3f ms, min %.3f ms (t_loop_ms %.3f)" % (1e3 * sorted(ts)[len(ts) // 2], 1e3 * min(ts), r.t_loop_ms))
