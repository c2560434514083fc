"""Host-side time stamps of one pwicp_pair_run on the bench pair (PWICP_HOST_TRACE) - and, under rocprofv3 --kernel-trace, the
kernel timeline of a run with the profiling flags given as argv[1] (0: no events at all; 1: dense events, the default)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P, _data
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = P.Context(0)
tgt, src, _ = _data.pair(1000000)
l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
pair.set_profiling(flags)
for i in range(5):
    pair.reset(); pair.run(check=False)
os.environ["PWICP_HOST_TRACE"] = "1"
pair.reset(); r = pair.run(check=False)
print("profiling flags %d: loop %.4f ms, dense events %.4f ms" % (flags, r.t_loop_ms, r.t_dense_nn_ms))
