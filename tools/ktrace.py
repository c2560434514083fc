"""Phase stamps of the fused classification launch (build with `make -C piecewise-icp_amd EXTRA=-DPWICP_KTRACE`): runs the
bench pair a few times and prints, for the LAST k_classify_icp0 launch of a run, the wall-clock offsets of its phases."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data
L = P.load_library()
L.pwicp_debug_ktrace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ctx = P.Context(0)
if len(sys.argv) > 1 and sys.argv[1] == "real":          # python tools/ktrace.py real [epoch]: one of the reference's own pairs (fixtures)
    from pwicp_amd.pcd import read_pcd
    g = os.path.join(ROOT, "tests", "golden", "inputs")
    e = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    tgt = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 5.0)
    src = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_%03d.pcd" % e)), 0.005, 14, 5.0)
    cen = tgt[:, :3].mean(0); tgt[:, :3] -= cen; src[:, :3] -= cen
    l1, n1 = ctx.frontend_segment(tgt, 0.05, 45, 0.005)
    l2, n2 = ctx.frontend_segment(src, 0.05, 45, 0.005)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    tgt, src, _ = _data.pair(n)
    l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
    l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
names = ["first block starts", "classified (last block to get there)", "base known (last)", "partials stored (last)", "last block identified",
         "totals done, tail starts", "tail: partials summed", "tail: 6x6 inverted", "tail: T formed", "tail done", "mail sent"]
for rep in range(4):
    pair.reset()
    L.pwicp_debug_ktrace(None, 1)
    pair.run(check=False)
    buf = (C.c_ulonglong * 32)()
    L.pwicp_debug_ktrace(buf, 0)
    t0 = buf[0]
    print("run %d: " % rep + " | ".join("%s +%.2f us" % (names[i], (buf[i] - t0) / 100.0) for i in range(1, 11)))
    if buf[11]:
        print("   classify, between `classified` and `partials stored`: rows formed +%.2f | rows in LDS (barrier) +%.2f | segment sums +%.2f | barrier +%.2f us" %
              tuple((buf[i] - t0) / 100.0 for i in (14, 11, 12, 13)))
    # the last REAL k_icp_iter launch of the run (launches after convergence do not stamp)
    inames = {17: "NN done (last block)", 18: "partials stored (last)", 19: "last block identified", 22: "tail: partials summed",
              23: "tail: 6x6 inverted", 24: "tail: T formed", 25: "tail done"}
    if buf[16]:
        print("   k_icp_iter: " + " | ".join("%s +%.2f us" % (inames[i], (buf[i] - buf[16]) / 100.0) for i in sorted(inames)))
    if buf[26]:
        vn = {27: "partials summed", 28: "solved (wave 0)", 29: "residual loop done (last wave to write)", 30: "VCM formed", 31: "mail sent"}
        print("   vcm tail: " + " | ".join("%s +%.2f us" % (vn[i], (buf[i] - buf[26]) / 100.0) for i in sorted(vn)))
