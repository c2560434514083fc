"""Runs the same resident pair repeatedly and reports how many distinct results (T, VCM, DT series, counts) come out.
A correct build gives exactly one.  python tools/determinism_check.py [n_points] [repeats] [labels]"""
import os, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
labels = sys.argv[3] if len(sys.argv) > 3 else "grid"
r = 0.005
ctx = P.Context(0)
t, _ = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1)
c = t.mean(0); t = (t - c).astype(np.float32); s = (s - c).astype(np.float32)
if labels == "grid":
    l1, n1 = synth.grid_labels(t, 10 * r); l2, n2 = synth.grid_labels(s, 10 * r)
else:
    l1, n1 = ctx.frontend_segment(t, 10 * r, 45, r); l2, n2 = ctx.frontend_segment(s, 10 * r, 45, r)
prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
pair = P.Pair(ctx, t, l1, n1, s, l2, n2, prm)
seen = {}
for i in range(reps):
    pair.reset()
    res = pair.run()
    key = (bytes(np.array(res.T16, np.float32)), bytes(np.array(res.VCM, np.float64)), res.n_outer, int(res.n_inner_total),
           bytes(np.array(res.DTseries[:res.n_outer + 1], np.float32)), bytes(np.array(res.maxBB[:res.n_outer], np.float32)),
           tuple(res.n_stable[:res.n_outer]), bytes(np.array(res.d75[:res.n_outer], np.float64)))
    seen.setdefault(key, []).append(i)
print("distinct results: %d over %d runs" % (len(seen), reps))
if len(seen) > 1:
    keys = list(seen)
    names = ["T", "VCM", "n_outer", "n_inner", "DTseries", "maxBB", "n_stable", "d75"]
    for j, nm in enumerate(names):
        if len({k[j] for k in keys}) > 1:
            print("  differs:", nm)
    print("  groups:", [len(v) for v in seen.values()])
