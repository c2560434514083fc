"""The dense search of one of the reference's own pairs: its time inside the loop (HIP events around the launch, cold caches: the
launch runs once per Stage-1 iteration, ~100 us of other kernels before it) against 20 back-to-back replays of the same launch
(warm L2).  python tools/real_dense_replay.py [epoch]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
e = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = os.path.join(ROOT, "tests", "golden", "inputs")
ctx = P.Context(0)
p1 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 5.0)
p2 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_%03d.pcd" % e)), 0.005, 14, 5.0)
cen = p1[:, :3].mean(0); p1[:, :3] -= cen; p2[:, :3] -= cen
l1, n1 = ctx.frontend_segment(p1, 0.05, 45, 0.005)
l2, n2 = ctx.frontend_segment(p2, 0.05, 45, 0.005)
pair = P.Pair(ctx, p1, l1, n1, p2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
pair.set_profiling(1 | 4)
for _ in range(5):
    pair.reset(); r = pair.run()
print("epoch %d: in the loop %d dense launches, %.1f us each on average (events around search + far launch)" % (e, r.n_dense_nn_launches, 1e3 * r.t_dense_nn_ms / max(r.n_dense_nn_launches, 1)))
ms, nq, kb, edge = pair.bench_dense_nn(20)
print("replayed back to back (first launch of the run, %d queries, kbar %.1f): %.1f us per launch" % (nq, kb, 1e3 * ms))
