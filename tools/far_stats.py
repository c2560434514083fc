"""Where the far queries of the dense search end (build: tools/build_variant.sh farstats "-DPW_FAR_STATS" grid; run with
PWICP_LIB=.../variants/libpwicp_farstats.so):  python tools/far_stats.py [epoch]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd.pcd import read_pcd
e = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = os.path.join(ROOT, "tests", "golden", "inputs")
ctx = P.Context(0)
p1 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 5.0)
p2 = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_%03d.pcd" % e)), 0.005, 14, 5.0)
cen = p1[:, :3].mean(0); p1[:, :3] -= cen; p2[:, :3] -= cen
l1, n1 = ctx.frontend_segment(p1, 0.05, 45, 0.005)
l2, n2 = ctx.frontend_segment(p2, 0.05, 45, 0.005)
pair = P.Pair(ctx, p1, l1, n1, p2, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
L = P.load_library()
out = (C.c_ulonglong * 16)()
L.pwicp_debug_far_stats(out, 1)
r = pair.run()
L.pwicp_debug_far_stats(out, 1)
names = ["queries", "fine: candidate's ball", "fine: small ball", "fine: wide ball", "coarse: candidate's ball", "coarse: small ball",
         "coarse: wide ball", "general search"]
print("epoch %d, dense launches %d, dense queries %d" % (e, r.n_dense_nn_launches, r.n_corr_dense))
for k, nme in enumerate(names): print("  %-26s %8d" % (nme, out[k]))
print("  block search: %d blocks / shells in all, largest radius %d coarse cells" % (out[10], out[11]))
