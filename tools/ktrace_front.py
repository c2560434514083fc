"""Per-role wall-clock stamps of k_xf_front (build with `make -C piecewise-icp_amd EXTRA=-DPWICP_KTRACE` after touching
csrc/patch.hip): first start / last end of the normal, query and cloud blocks of the LAST k_xf_front launch of a run."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data
L = P.load_library()
L.pwicp_debug_ftrace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
L.pwicp_debug_vtrace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = P.Context(0)
tgt, src, _ = _data.pair(n)
l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
import numpy as np
pair.run(check=False)
for rep in range(3):
    pair.reset()
    L.pwicp_debug_ftrace(None, 1)
    L.pwicp_debug_vtrace(None, 1)
    pair.run(check=False)
    buf = (C.c_ulonglong * (3 * 8192))()
    L.pwicp_debug_ftrace(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    out = []
    for r, nm in enumerate(("normals", "queries", "cloud")):
        b = a[a[:, 2] == r]
        if len(b) == 0:
            continue
        dur = (b[:, 1] - b[:, 0]) / 100.0
        out.append("%s: %d blocks, first start %.2f, last start %.2f, last end %.2f us, block time mean %.2f max %.2f" % (
            nm, len(b), (b[:, 0].min() - t0) / 100.0, (b[:, 0].max() - t0) / 100.0, (b[:, 1].max() - t0) / 100.0, dur.mean(), dur.max()))
    print("run %d (last k_xf_front launch):\n   " % rep + "\n   ".join(out))
    vb = (C.c_ulonglong * (3 * 4096))()
    L.pwicp_debug_vtrace(vb, 0)
    a = np.frombuffer(vb, dtype=np.uint64).reshape(-1, 3).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    out = []
    for r, nm in enumerate(("vcm", "ctbp+patch points", "cloud")):
        b = a[a[:, 2] == r]
        if len(b) == 0:
            continue
        b = b[b[:, 1] > 0]
        dur = (b[:, 1] - b[:, 0]) / 100.0
        out.append("%s: %d blocks, first start %.2f, last start %.2f, last end %.2f us, block time mean %.2f max %.2f" % (
            nm, len(b), (b[:, 0].min() - t0) / 100.0, (b[:, 0].max() - t0) / 100.0, (b[:, 1].max() - t0) / 100.0, dur.mean(), dur.max()))
    print("   k_xf_vcm:\n   " + "\n   ".join(out))
