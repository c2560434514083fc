"""Where the queries of the window search (k_nn_dense_win) end, on the synthetic 1 M-point pair (build: tools/build_variant.sh winstats
"-DPW_WIN_STATS" grid; run with PWICP_LIB=.../variants/libpwicp_winstats.so):  python tools/win_stats.py"""
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
L = P.load_library()
out = (C.c_ulonglong * 16)()
L.pwicp_debug_win_stats(out, 1)
res = pair.run()
L.pwicp_debug_win_stats(out, 1)
names = ["blocks with a query", "windows that fitted", "  rows (sum)", "  cells (sum)", "  points (sum)", "lanes resolved on the window",
         "lanes without a candidate", "lanes whose ball left the window", "lanes of blocks searching global memory", "lanes outside the grid",
         "windows refused: rows", "windows refused: cells", "windows refused: points"]
print("dense launches %d, dense queries %d" % (res.n_dense_nn_launches, res.n_corr_dense))
for k, nme in enumerate(names): print("  %-42s %10d" % (nme, out[k]))
if out[1]: print("  per window: %.1f rows, %.1f cells, %.0f points" % (out[2] / out[1], out[3] / out[1], out[4] / out[1]))
