"""Randomised comparison of the device front end with the serial host passes (run on the GPU box): clouds of random size,
roughness, anisotropy and supervoxel size, random sweep schedules.  Usage: fe_fuzz.py [cases] [seed]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + "/piecewise-icp_amd")
import pwicp_amd as P
from pwicp_amd import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = P.Context(0)
r = 0.005
bad = 0
for c in range(cases):
    n = int(rng.choice([3000, 12000, 50000, 120000, 300000]))
    cloud, _ = synth.make_tile(n, r, offset=(float(rng.uniform(0, 3)), float(rng.uniform(0, 3)), 0.0))
    cloud = (cloud - cloud.mean(0)).astype(np.float32)
    kind = int(rng.integers(0, 4))
    if kind == 1:      # rough
        cloud[:, 2] += rng.normal(0, float(rng.uniform(0.5, 3)) * r, len(cloud)).astype(np.float32)
    elif kind == 2:    # steep
        cloud[:, 2] += (float(rng.uniform(0.2, 1.5)) * np.sin(float(rng.uniform(2, 9)) * cloud[:, 0])).astype(np.float32)
    elif kind == 3:    # shuffled point order (no scan-line coherence)
        cloud = cloud[rng.permutation(len(cloud))]
    sv = float(rng.choice([3, 5, 10, 10, 20, 45])) * r
    knobs = {}
    if rng.random() < 0.5:
        knobs["PWICP_FUSION_CHUNK"] = str(int(rng.choice([1, 2, 5, 16])))
    if rng.random() < 0.3:
        knobs["PWICP_FUSION_WAKE_DIV"] = str(int(rng.choice([1, 4, 1000])))
    if rng.random() < 0.2:
        knobs["PWICP_FUSION_QUEUE"] = str(int(rng.choice([40, 100, 300])))
    if rng.random() < 0.4:
        knobs["PWICP_FUSION_BATCH"] = str(int(rng.choice([1, 2, 16])))
    if rng.random() < 0.5:
        knobs["PWICP_FUSION_TILE"] = str(int(rng.choice([0, 3, 16, 100])))
    if rng.random() < 0.4:
        knobs["PWICP_FUSION_COLOURS"] = str(int(rng.choice([1, 2, 4])))
    if rng.random() < 0.3:
        knobs["PWICP_FUSION_CHUNK_DIV"] = str(int(rng.choice([16, 256, 8192])))
    if rng.random() < 0.3:
        knobs["PWICP_FE_PIECES"] = str(int(rng.choice([1, 2, 3, 8])))
    if rng.random() < 0.15:
        knobs["PWICP_FE_AHEAD"] = "0"
    out = {}
    for mode in ("host", "device"):
        os.environ["PWICP_FRONTEND"] = mode
        os.environ.update(knobs if mode == "device" else {})
        out[mode] = ctx.frontend_segment(cloud, sv, 45, r)
        for k in knobs:
            os.environ.pop(k, None)
    same = out["host"][1] == out["device"][1] and np.array_equal(out["host"][0], out["device"][0])
    bad += not same
    print("case %2d n=%6d kind=%d sv=%.3f knobs=%s nsv=%d %s" % (c, len(cloud), kind, sv, knobs, out["device"][1], "ok" if same else "DIFFERENT"), flush=True)
print("different: %d of %d" % (bad, cases))
