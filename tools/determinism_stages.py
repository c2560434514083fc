"""Hashes the output of every setup stage and of the loop on the golden Epoch_001/002 scans, several times over, to find
a stage whose result changes from run to run."""
import hashlib, os, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
g = os.path.join(R_, "tests", "golden", "inputs")
c1 = read_pcd(os.path.join(g, "Epoch_001.pcd")); c2 = read_pcd(os.path.join(g, "Epoch_002.pcd"))
def h(a): return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:8]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rows = []
for i in range(reps):
    ctx = P.Context(0)
    p1 = ctx.preprocess(c1, 0.005, 14, 2.7); p2 = ctx.preprocess(c2, 0.005, 14, 2.7)
    cen = p1[:, :3].mean(0)
    q1 = p1.copy(); q1[:, :3] -= cen; q2 = p2.copy(); q2[:, :3] -= cen
    l1, n1 = ctx.frontend_segment(q1, 0.05, 45, 0.005); l2, n2 = ctx.frontend_segment(q2, 0.05, 45, 0.005)
    S1 = ctx.selectPatches(q1, l1, n1); S2 = ctx.selectPatches(q2, l2, n2)
    prm = P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004)
    pair = P.Pair(ctx, q1, l1, n1, q2, l2, n2, prm)
    res = pair.run()
    rows.append((h(p1), h(p2), h(l1), h(l2), h(S1["pat"]), h(S2["pat"]), h(S2["ct"]), h(np.array(res.T16, np.float32)),
                 res.n_outer, int(res.n_inner_total), h(np.array(res.DTseries[:res.n_outer + 1], np.float32)),
                 h(np.array(res.maxBB[:res.n_outer], np.float32)), tuple(res.n_stable[:res.n_outer])))
    pair.close()
names = ["prep1", "prep2", "lab1", "lab2", "pat1", "pat2", "ct2", "T", "n_outer", "n_inner", "DT", "maxBB", "n_stable"]
for j, nm in enumerate(names):
    vals = [r[j] for r in rows]
    print("%-9s %s %s" % (nm, "same   " if len(set(vals)) == 1 else "DIFFERS", vals if len(set(vals)) > 1 else vals[0]))
