import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P, _data
ctx = P.Context(0)
cl = _data.pair(1000000, epoch=1)[1]
ctx.frontend_segment(cl, 10 * _data.R, 45, _data.R)
os.environ["PWICP_TRACE"] = "1"; os.environ["PWICP_TRACE_SWEEPS"] = "1"
ctx.frontend_segment(cl, 10 * _data.R, 45, _data.R)
