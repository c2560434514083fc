"""Experiment of DESIGN 4.5: the eigen step of the PCA normals with the device library's pow / acos / cos ($PWICP_NORMALS=device)
against libm's - how many normals differ, and whether a label moves (run on the GPU box)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + "/piecewise-icp_amd"); sys.path.insert(0, ROOT + "/tests")
import pwicp_amd as P
from pwicp_amd import synth
from pwicp_amd.pcd import read_pcd
ctx = P.Context(0); r = 0.005
def both(cloud, sv, name):
    os.environ["PWICP_FRONTEND"] = "host"
    lh, nh = ctx.frontend_segment(cloud, sv, 45, r)
    os.environ["PWICP_FRONTEND"] = "device"; os.environ["PWICP_NORMALS"] = "device"
    ld, nd = ctx.frontend_segment(cloud, sv, 45, r)
    os.environ.pop("PWICP_NORMALS")
    print("%-14s n=%7d  labels differ: %d  (nsv %d / %d)" % (name, len(cloud), int((lh != ld).sum()) if len(lh) == len(ld) else -1, nh, nd), flush=True)
for e in range(1, 21):
    raw = read_pcd(ROOT + "/tests/golden/inputs/Epoch_%03d.pcd" % e)
    c = ctx.preprocess(raw, 0.005, 14, 5.0); c = (c - c.mean(0)).astype(np.float32)
    both(c, 0.05, "Epoch_%03d" % e)
for n in (300000, 1000000):
    t, _ = synth.make_tile(n, r); t = (t - t.mean(0)).astype(np.float32)
    both(t, 10 * r, "synthetic")
