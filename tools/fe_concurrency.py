"""Do front ends of different clouds overlap on one GPU?  N clouds of 1 M points segmented one after the other on one context,
then side by side on N contexts (= streams) from N host threads."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
clouds = [_data.pair(n, epoch=e + 1)[1] for e in range(N)]
ctxs = [P.Context(0) for _ in range(N)]
for c, cl in zip(ctxs, clouds):
    c.frontend_segment(cl, 10 * _data.R, 45, _data.R)          # warm-up: workspaces of every context
t0 = time.perf_counter()
for cl in clouds:
    ctxs[0].frontend_segment(cl, 10 * _data.R, 45, _data.R)
t_seq = time.perf_counter() - t0
def work(i):
    ctxs[i].frontend_segment(clouds[i], 10 * _data.R, 45, _data.R)
for k in (2, N):
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    print("%d clouds side by side: %.1f ms (%.1f ms per cloud)" % (k, 1e3 * (time.perf_counter() - t0), 1e3 * (time.perf_counter() - t0) / k))
print("%d clouds one after the other: %.1f ms (%.1f ms per cloud)" % (N, 1e3 * t_seq, 1e3 * t_seq / N))
