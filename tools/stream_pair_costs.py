"""What a streamed epoch costs around the loop (5 M points): host->device copy of the scan from pageable / pinned memory,
pwicp_pair_create_with_target, the loop, pwicp_pair_destroy."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pwicp_amd as P
from pwicp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
r = 0.005
ctx = P.Context(0)
tgt, _ = synth.make_tile(n, r)
c = tgt.mean(axis=0); tgt = (tgt - c).astype(np.float32)
src, _ = synth.make_source(n, r, epoch=1); src = (src - c).astype(np.float32)
l1, n1 = ctx.frontend_segment(tgt, 10 * r, 45, r)
l2, n2 = ctx.frontend_segment(src, 10 * r, 45, r)
prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
T = P.Target(ctx, tgt, l1, n1, prm.Res1, prm.SVRes1)
s4 = P.f4(src); l2 = np.ascontiguousarray(l2, np.int32)
d = torch.empty(s4.shape, dtype=torch.float32, device="cuda")
for name, h in (("pageable", torch.from_numpy(s4)), ("pinned", torch.from_numpy(s4).pin_memory())):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("copy of %d MB from %s memory: %.2f ms (%.1f GB/s)" % (s4.nbytes >> 20, name, 1e3 * t, s4.nbytes / t / 1e9))
for rep in range(3):
    ta = time.perf_counter(); pair = P.Pair(ctx, None, None, 0, s4, l2, n2, prm, target=T)
    tb = time.perf_counter(); res = pair.run()
    tc = time.perf_counter(); pair.close(); td = time.perf_counter()
    print("create %.2f ms | loop %.2f ms | destroy %.2f ms" % (1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (td - tc)))
os.environ["PWICP_TRACE"] = "1"
pair = P.Pair(ctx, None, None, 0, s4, l2, n2, prm, target=T)
