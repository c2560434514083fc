"""How much of bench.py's ms_per_step is the Python / ctypes layer: wall per step against the loop time measured inside
pwicp_pair_run (run on the GPU box)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth
ctx = P.Context(0); r = 0.005; n = 1000000
t, _ = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1); c = t.mean(0)
t = (t - c).astype(np.float32); s = (s - c).astype(np.float32)
l1, n1 = ctx.frontend_segment(t, 10 * r, 45, r); l2, n2 = ctx.frontend_segment(s, 10 * r, 45, r)
pair = P.Pair(ctx, t, l1, n1, s, l2, n2, P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r))
for _ in range(3):
    pair.reset(); pair.run()
K = 50
t0 = time.perf_counter(); inner = 0.0
for _ in range(K):
    pair.reset(); res = pair.run(); inner += res.t_loop_ms
wall = (time.perf_counter() - t0) / K * 1e3
print("wall per step %.4f ms, inside pwicp_pair_run %.4f ms, layer %.1f us" % (wall, inner / K, (wall - inner / K) * 1e3))
t0 = time.perf_counter()
for _ in range(K): pair.reset()
print("reset alone: %.1f us" % ((time.perf_counter() - t0) / K * 1e6))
t0 = time.perf_counter()
for _ in range(K): P.Result()
print("Result() alone: %.1f us" % ((time.perf_counter() - t0) / K * 1e6))
res = P.Result()
import ctypes as C
t0 = time.perf_counter(); inner = 0.0
for _ in range(K):
    pair.reset(); pair._L.pwicp_pair_run(pair._h, C.byref(res)); inner += res.t_loop_ms
wall = (time.perf_counter() - t0) / K * 1e3
print("with a reused Result: wall %.4f ms, inside %.4f ms, layer %.1f us" % (wall, inner / K, (wall - inner / K) * 1e3))
