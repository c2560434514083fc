"""Device front end against the host passes: identical labels, stage timings (run on the GPU box)."""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth

ctx = P.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [100000, 1000000]
r = 0.005
for n in sizes:
    t, _ = synth.make_tile(n, r)
    t = (t - t.mean(0)).astype(np.float32)
    out = {}
    for mode in ("host", "device"):
        os.environ["PWICP_FRONTEND"] = mode
        t0 = time.time()
        lab, nsv = ctx.frontend_segment(t, float(os.environ.get("FE_SV", "10")) * r, 45, r)
        out[mode] = (lab, nsv, time.time() - t0)
    same = out["host"][1] == out["device"][1] and np.array_equal(out["host"][0], out["device"][0])
    print("n=%d nsv=%d/%d host %.3f s device %.3f s identical=%s" % (n, out["host"][1], out["device"][1], out["host"][2], out["device"][2], same),
          flush=True)
