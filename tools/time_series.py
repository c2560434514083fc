"""Wall time of the exported 4D entry point on a synthetic series (reference epoch + E source epochs of N points, PCD files):
python tools/time_series.py [n_points] [n_source_epochs]     (PWICP_SERIES_WINDOW=1 = one pair at a time)"""
import os, sys, tempfile, time
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
import pwicp_amd as P
from pwicp_amd import synth
from pwicp_amd.pcd import write_pcd_binary
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = 0.005
d = tempfile.mkdtemp()
inp = os.path.join(d, "scans"); os.mkdir(inp)
t, _ = synth.make_tile(n, r)
write_pcd_binary(os.path.join(inp, "Epoch_001.pcd"), t.astype(np.float32))
for e in range(1, E + 1):
    s, _ = synth.make_source(n, r, epoch=e)
    write_pcd_binary(os.path.join(inp, "Epoch_%03d.pcd" % (e + 1)), s.astype(np.float32))
cfg = os.path.join(d, "cfg.txt")
open(cfg, "w").write("string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\n"
                     "float PCres1 (m): %g\nfloat PCres2 (m): %g\nfloat SVsize1 (m): %g\nfloat SVsize2 (m): %g\n"
                     "bool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): %g\nfloat DTmin (m): %g\nbool isVisual (yes-1, no-0): 0"
                     % (inp, os.path.join(d, "out_"), r, r, 10 * r, 10 * r, 10 * r, 0.8 * r))
os.chdir(d)
t0 = time.perf_counter()
ok = P.PiecewiseICP_4D_call(cfg, 0, E + 1, 0, 0.75)
dt = time.perf_counter() - t0
print("PiecewiseICP_4D_call: %s, %d pairs of %d points in %.2f s wall (%.2f s per pair), window=%s"
      % (ok, E, n, dt, dt / E, os.environ.get("PWICP_SERIES_WINDOW", "auto")), file=sys.stderr)
