"""Randomised check of the dense 1-NN search (k_nn_dense_disc + far paths, pwicp_pair_dense_distances) against the general search
(pwicp_nn_search, itself held to the oracle's exhaustive search by tests/test_gpu_parity.py): random sizes, point densities (cells of
the searched level with 1 ... 40 points), roughness, tilt (levels of columns and of cells), offsets between the clouds from a tenth of
a spacing to beyond the target's coverage, duplicates.  Every float d2 must be bit-equal.  python tools/dense_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd import synth
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = P.Context(0)
bad = 0
for c in range(cases):
    r = 0.005
    n = int(rng.choice([3000, 20000, 80000, 250000, 600000]))
    dens = float(rng.choice([0.35, 0.6, 1.0, 1.0, 1.7, 3.0]))          # spacing of the clouds relative to the configured resolution
    tgt, _ = synth.make_tile(n, r * dens)
    src, _ = synth.make_source(n, r * dens, epoch=int(rng.integers(1, 9)))
    cen = tgt.mean(0)
    tgt = (tgt - cen).astype(np.float32); src = (src - cen).astype(np.float32)
    kind = int(rng.integers(0, 5))
    if kind == 1:                                                       # rough
        for a in (tgt, src): a[:, 2] += rng.normal(0, float(rng.uniform(0.3, 3)) * r, len(a)).astype(np.float32)
    elif kind == 2:                                                     # steep: cells instead of columns
        for a in (tgt, src): a[:, 2] += (float(rng.uniform(0.5, 2.0)) * np.sin(float(rng.uniform(3, 9)) * a[:, 0])).astype(np.float32)
    elif kind == 3:                                                     # the tile turned into another plane
        perm = [[2, 0, 1], [1, 2, 0]][int(rng.integers(0, 2))]
        tgt = np.ascontiguousarray(tgt[:, perm]); src = np.ascontiguousarray(src[:, perm])
    elif kind == 4:                                                     # duplicates in the target
        tgt[rng.integers(0, len(tgt), len(tgt) // 20)] = tgt[rng.integers(0, len(tgt), len(tgt) // 20)]
    off = float(rng.choice([0.1, 0.5, 1.5, 4.0, 12.0, 60.0])) * r
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    src = (src + (off * d).astype(np.float32)).astype(np.float32)
    if rng.random() < 0.3:                                              # a part of the source beyond the target's coverage
        src[: len(src) // 4, 0] += np.float32(float(rng.uniform(0.05, 0.6)))
    sv = 10 * r
    l1, n1 = synth.grid_labels(tgt, sv); l2, n2 = synth.grid_labels(src, sv)
    try:
        pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, P.Params(r, r, sv, sv, 1, 10 * r, 0.8 * r))
    except P.PwicpError as e:
        print("case %d: pair not created (%s)" % (c, e)); continue
    q = pair.source_patch_points()
    if len(q) == 0:
        print("case %d: no source patches" % c); pair.close(); continue
    ok = True
    for fg in (0, 1):
        d2 = pair.dense_distances(far_group=fg)
        _, ref = ctx.determineCorrespondences(tgt, q[:, :3])
        same = d2.tobytes() == np.asarray(ref, np.float32).tobytes()
        ok = ok and same
        if not same:
            w = np.nonzero(d2 != ref)[0]
            print("case %d far_group %d: %d of %d differ, first %d: %r vs %r" % (c, fg, len(w), len(d2), w[0], d2[w[0]], ref[w[0]]))
    bad += 0 if ok else 1
    print("case %2d: n %6d spacing %.2f Res kind %d offset %5.1f Res queries %7d  %s" % (c, n, dens, kind, off / r, len(q), "ok" if ok else "MISMATCH"))
    pair.close()
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
