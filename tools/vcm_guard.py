import ctypes as C, os, sys, struct
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P, _data
L = P.load_library()
L.pwicp_debug_ktrace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ctx = P.Context(0)
for n in (60000, 200000, 1000000):
    tgt, src, _ = _data.pair(n)
    l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
    l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
    L.pwicp_debug_ktrace(None, 1)
    pair.run(check=False)
    buf = (C.c_ulonglong * 32)()
    L.pwicp_debug_ktrace(buf, 0)
    print(n, "explicit" if buf[20] else "identity", "v'v / L'L = %.6g" % struct.unpack("d", struct.pack("Q", buf[21]))[0])
