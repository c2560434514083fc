"""Device front end against the serial passes on large clouds with different shapes (run on the GPU box)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth
ctx = P.Context(0); r = 0.005
rng = np.random.default_rng(3)
for case in range(6):
    n = int(rng.choice([700000, 1500000, 2500000]))
    t, _ = synth.make_tile(n, r, offset=(float(case), 0.0, 0.0)); t = (t - t.mean(0)).astype(np.float32)
    kind = case % 3
    if kind == 1:
        t[:, 2] += rng.normal(0, 1.5 * r, len(t)).astype(np.float32)
    elif kind == 2:
        t = t[rng.permutation(len(t))]
    sv = float(rng.choice([6, 10, 16])) * r
    out = {}
    for mode in ("host", "device"):
        os.environ["PWICP_FRONTEND"] = mode
        t0 = time.perf_counter(); out[mode] = ctx.frontend_segment(t, sv, 45, r); out[mode + "_t"] = time.perf_counter() - t0
    same = out["host"][1] == out["device"][1] and np.array_equal(out["host"][0], out["device"][0])
    print("n=%7d kind=%d sv=%.3f nsv=%d  host %.2f s device %.3f s  identical=%s" % (len(t), kind, sv, out["device"][1], out["host_t"], out["device_t"], same), flush=True)
