"""The registration loops of the reference's 19 pairs (Epoch_002 .. Epoch_020 -> Epoch_001, fixtures), K at a time through
pwicp_pairs_run_concurrent (K contexts of one GPU, one device-side target): wall time of all 19 against one after the other, results
compared bit by bit.  python tools/real_pairs_concurrent.py [K] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = os.path.join(ROOT, "tests", "golden", "inputs")
ctxs = [P.Context(0) for _ in range(K)]
c0 = ctxs[0]
p1 = c0.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 5.0)
cen = p1[:, :3].mean(0); p1[:, :3] -= cen
l1, n1 = c0.frontend_segment(p1, 0.05, 45, 0.005)
prm = P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004)
T = P.Target(c0, p1, l1, n1, prm.Res1, prm.SVRes1)
pairs = []
for k, e in enumerate(range(2, 21)):
    p2 = c0.preprocess(read_pcd(os.path.join(g, "Epoch_%03d.pcd" % e)), 0.005, 14, 5.0)
    p2[:, :3] -= cen
    l2, n2 = c0.frontend_segment(p2, 0.05, 45, 0.005)
    pairs.append(P.Pair(ctxs[k % K], None, None, 0, p2, l2, n2, prm, target=T))
def key(r):
    no = r.n_outer
    return (r.status, no, list(r.T16), list(r.VCM), list(r.DTseries[:no + 1]), list(r.n_inner[:no]))
alone = []
for pr in pairs:
    pr.reset(); alone.append(key(pr.run()))
def serial():
    t0 = time.perf_counter()
    for pr in pairs:
        pr.reset(); pr.run()
    return time.perf_counter() - t0
def concurrent():
    t0 = time.perf_counter()
    out = P.run_pairs_concurrent(pairs)          # (pair k lives on context k mod K: K chains of pairs, no barrier between them)
    return time.perf_counter() - t0, out
serial(); concurrent()
ts = sorted(serial() for _ in range(reps)); tc = []
same = True
for _ in range(reps):
    t, out = concurrent(); tc.append(t)
    same = same and [key(r) for r in out] == alone
tc.sort()
print("19 pairs, one after the other: median %.2f ms;  %d at a time (pwicp_pairs_run_concurrent): median %.2f ms;  results identical: %s"
      % (1e3 * ts[len(ts) // 2], K, 1e3 * tc[len(tc) // 2], same))
