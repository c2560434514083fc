"""Launches k_transform_all - a kernel whose HBM byte count is known by construction (every point of cloud2, of the centroid /
boundary array and of the patch array read once and written once, 16 B each) - on the bench pair, through the single-iteration
entry point (pwicp_pair_step keeps the un-merged launches).  Profiled with --pmc FETCH_SIZE / WRITE_SIZE by
tools/collect_profiles.sh, it calibrates the counters for tools/summarize_pmc.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P, _data
ctx = P.Context(0)
tgt, src, _ = _data.pair(1000000)
l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params())
for rep in range(3):
    pair.reset()
    st = P.Step()
    st.currDT = 10 * _data.R
    while not st.toStage3:
        assert pair.step(st) == 0 and st.status == 0
print("calibration launches done")
