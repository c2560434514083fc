"""What a wave of the dense 1-NN search executes against what its lanes need (round 6): candidate passes per wave
as the nested per-row loops run them (sum over rows of the max over lanes) against one flat per-lane loop (max over lanes of the
lane's own sum).  Build: tools/build_variant.sh slots "-DPW_DENSE_SLOTS" grid; run on the GPU box with
PWICP_LIB=piecewise-icp_amd/variants/libpwicp_slots.so python tools/dense_slots.py   (DV_POINTS=4000000 for the 4 M-point pair)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd import synth
n = int(os.environ.get("DV_POINTS", "1000000")); r = 0.005
ctx = P.Context(0)
t, _ = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1); c = t.mean(0)
t = (t - c).astype(np.float32); s = (s - c).astype(np.float32)
l1, n1 = synth.grid_labels(t, 10 * r); l2, n2 = synth.grid_labels(s, 10 * r)
pair = P.Pair(ctx, t, l1, n1, s, l2, n2, P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r))
pair.set_profiling(1 | 4)
res = pair.run()
L = P.load_library()
L.pwicp_debug_dense_slots(None, 0, 1)
ms, nq, kb, edge = pair.bench_dense_nn(1)          # warm-up + 1 replay of the run's first Stage-1 search = 2 identical launches
buf = (C.c_ulonglong * 128)()
L.pwicp_debug_dense_slots(buf, 128, 1)
S = np.array(buf, dtype=np.uint64).astype(np.float64) / 2.0
waves, lanes = S[0], S[1]
print("points %d  queries %d  Kbar %.1f  cell edge %.4f (%.1f r)" % (n, nq, kb, edge, edge / r))
print("waves with a stable query %d (with a resolved one %d), stable lanes %d, resolved on the small cells %d, handed to the far path %d"
      % (S[15], waves, S[12], lanes, S[11]))
print("phase A (own row segment): candidates per lane %.2f, passes per lane %.2f, passes the wave runs %.2f" % (S[9] / S[12], S[3] / S[12], S[2] / S[15]))
print("phase B (the ball):        candidates per lane %.2f, passes per lane %.2f" % (S[10] / lanes, S[5] / lanes))
print("   passes the wave runs today, sum over (slab, row, piece) of max over lanes:  %.2f" % (S[4] / waves))
print("   passes of ONE flat per-lane loop, max over lanes of the lane's sum:         %.2f" % (S[6] / waves))
print("   ... of one flat loop over phase A + B together:                             %.2f  (today A + B: %.2f)" % (S[7] / waves, (S[2] + S[4]) / waves))
print("   ratio nested / flat, phase B: %.2f x;  whole search: %.2f x" % (S[4] / S[6], (S[2] + S[4]) / (S[2] + S[6])))
print("   row set-ups (4 rows each) the wave runs: %.2f" % (S[8] / waves))
print("lanes whose ball stays inside rows cy-1 .. cy+1 x cells cx-1 .. cx+1 of a level of columns: %.1f %% of the resolved lanes; waves made of such lanes only: %.1f %%"
      % (100 * S[13] / lanes, 100 * S[14] / waves))
def hist(name, a, scale=1.0, unit=""):
    tot = a.sum()
    print(name + ": " + "  ".join("%g%s:%.1f%%" % (k * scale, unit, 100 * v / tot) for k, v in enumerate(a) if v > 0))
hist("ball radius (cells)", S[16:32], 0.25)
hist("non-empty ranges per lane", S[32:48])
hist("rows in the ball's box", S[48:64])
hist("lane's passes in phase B", S[64:80])
hist("wave's executed passes in phase B (bins of 2)", S[80:96], 2)
hist("wave's flat passes in phase B", S[96:112])
