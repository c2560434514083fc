"""Repeats the registration loop on the golden Epoch_001/002 pair (one resident pwicp_pair) and counts distinct results."""
import os, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
g = os.path.join(R_, "tests", "golden", "inputs")
c1 = read_pcd(os.path.join(g, "Epoch_001.pcd")); c2 = read_pcd(os.path.join(g, "Epoch_002.pcd"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = P.Context(0)
p1 = ctx.preprocess(c1, 0.005, 14, 2.7); p2 = ctx.preprocess(c2, 0.005, 14, 2.7)
cen = p1[:, :3].mean(0)
p1[:, :3] -= cen; p2[:, :3] -= cen
l1, n1 = ctx.frontend_segment(p1, 0.05, 45, 0.005); l2, n2 = ctx.frontend_segment(p2, 0.05, 45, 0.005)
pair = P.Pair(ctx, p1, l1, n1, p2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
seen = {}
for i in range(reps):
    pair.reset()
    res = pair.run()
    key = (tuple(res.n_inner[:res.n_outer]), tuple(res.n_stable[:res.n_outer]), bytes(np.array(res.Tk[0], np.float32)),
           bytes(np.array(res.maxBB[:res.n_outer], np.float32)), bytes(np.array(res.d75[:res.n_outer], np.float64)),
           bytes(np.array(res.T16, np.float32)))
    seen.setdefault(key, []).append(i)
print("distinct results: %d over %d runs, group sizes %s" % (len(seen), reps, [len(v) for v in seen.values()]))
ks = list(seen)
for a in ks[1:3]:
    names = ["n_inner", "n_stable", "Tk[0]", "maxBB", "d75", "T"]
    print("  vs first:", [(nm, ks[0][j], a[j]) if j < 2 else nm for j, nm in enumerate(names) if a[j] != ks[0][j]])
