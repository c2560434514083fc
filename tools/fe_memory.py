"""Device memory the front end's work buffers keep with a context (grow-only; run on the GPU box)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/piecewise-icp_amd")
import pwicp_amd as P, torch
from pwicp_amd import synth
r = 0.005
def free(): return torch.cuda.mem_get_info(0)[0] / 2**20
for n in (140000, 1000000, 5000000):
    ctx = P.Context(0)
    t, _ = synth.make_tile(n, r); t = (t - t.mean(0)).astype(np.float32)
    f0 = free()
    ctx.frontend_segment(t, 10 * r, 45, r)
    print("n=%7d: %.0f MiB kept by the context (%.0f bytes per point)" % (n, f0 - free(), (f0 - free()) * 2**20 / n), flush=True)
    ctx.close()
