"""End-to-end wall time of the exported pair entry point on a synthetic N-point pair written to PCD files:
python tools/time_pair_end_to_end.py [n_points]   (run with PWICP_TRACE=1 for the stage split)"""
import os, sys, tempfile, time
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth
from pwicp_amd.pcd import write_pcd_binary
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
r = 0.005
d = tempfile.mkdtemp()
t, _ = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1)
write_pcd_binary(os.path.join(d, "t.pcd"), t.astype(np.float32)); write_pcd_binary(os.path.join(d, "s.pcd"), s.astype(np.float32))
cfg = os.path.join(d, "cfg.txt")
open(cfg, "w").write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                     "float PCres1 (m): %g\nfloat PCres2 (m): %g\nfloat SVsize1 (m): %g\nfloat SVsize2 (m): %g\n"
                     "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): %g\nfloat DTmin (m): %g\nbool isVisual (yes-1, no-0): 0"
                     % (os.path.join(d, "t.pcd"), os.path.join(d, "s.pcd"), r, r, 10 * r, 10 * r, 10 * r, 0.8 * r))
for k in range(4):
    t0 = time.perf_counter()
    ok = P.PiecewiseICP_pair_call(cfg, os.path.join(d, "out%d_" % k))
    print("PiecewiseICP_pair_call #%d: %s, %.2f s wall" % (k, ok, time.perf_counter() - t0), file=sys.stderr)
