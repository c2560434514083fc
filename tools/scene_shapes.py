import sys, os, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth
n = 1000000; r = 0.005
ctx = P.Context(0)
for shape in ("flat", "steep_z", "face_yz", "diagonal"):
    t, L = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1); c = t.mean(0)
    t = (t - c).astype(np.float32); s = (s - c).astype(np.float32)
    out = []
    for a in (t, s):
        a = a.copy()
        if shape == "steep_z":
            a[:, 2] += (1.5 * np.sin(6.0 * a[:, 0])).astype(np.float32)
        elif shape == "face_yz":
            a = np.ascontiguousarray(a[:, [2, 0, 1]])
        elif shape == "diagonal":
            c_, s_ = np.float32(np.cos(np.pi / 4)), np.float32(np.sin(np.pi / 4))
            x, z = a[:, 0].copy(), a[:, 2].copy()
            a[:, 0] = c_ * x + s_ * z; a[:, 2] = -s_ * x + c_ * z
        out.append(a.astype(np.float32))
    t, s = out
    l1, n1 = ctx.frontend_segment(t, 10 * r, 45, r); l2, n2 = ctx.frontend_segment(s, 10 * r, 45, r)
    prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
    pair = P.Pair(ctx, t, l1, n1, s, l2, n2, prm); pair.set_profiling(1 | 4)
    res = pair.run(check=False)
    ms, nq, kb, edge = pair.bench_dense_nn(20)
    best = 1e9
    for _ in range(5):
        pair.reset(); rr = pair.run(check=False); best = min(best, rr.t_loop_ms)
    print("%-9s status %d outer %d inner %d  loop %.3f ms  dense %.1f us (nq %d, kbar %.1f)  patches %d/%d" % (shape, res.status, res.n_outer, res.n_inner_total, best, ms * 1e3, nq, kb, n1, n2), flush=True)
    pair.close()
