"""Where the front launches' centroid / boundary-point queries end (nn_device.h: nn_query_group), and how long they take:
build with `make -C piecewise-icp_amd EXTRA="-DPWICP_KTRACE -DPWICP_QSTAT"` after touching csrc/patch.hip.  Counts are over one
run of the 1 M-point pair (k_front + the run's k_xf_front launches).  (No times: 10^5 same-address atomics distort them.)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data
L = P.load_library()
L.pwicp_debug_qstat.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = P.Context(0)
tgt, src, _ = _data.pair(n)
l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
pair.run(check=False)
pair.reset()
L.pwicp_debug_qstat(None, 1)
r = pair.run(check=False)
q = (C.c_ulonglong * 32)()
L.pwicp_debug_qstat(q, 0)
q = list(q)
tot = q[0] + q[1] + q[2]
print("outer iterations", r.n_outer, "queries", tot)
for s in range(3):
    if q[s]: print("  stage %d exit: %8d (%.1f %%)" % (s + 1, q[s], 100.0 * q[s] / tot))
if q[1]: print("  stage 2: mean rows %.1f, mean points scanned %.1f" % (q[8] / q[1], q[9] / q[1]))
