"""Bit-level fingerprints of the loop's results on a set of synthetic pairs (grid labels and supervoxel labels, several sizes,
epochs and offsets): T16, VCM, every per-iteration series of pwicp_result.  Run before and after a change that claims to leave
the arithmetic alone (a different summation order or elimination order shows up here) and compare the output lines.
  python tools/result_hash.py [out.json]"""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data

ctx = P.Context(0)
out = {}
cases = [(60000, 1, (0, 0, 0)), (60000, 3, (0, 0, 0)), (200000, 2, (100.0, -50.0, 10.0)), (1000000, 1, (0, 0, 0)), (1000000, 4, (0, 0, 0))]
for (n, ep, off) in cases:
    tgt, src, _ = _data.pair(n, epoch=ep, offset=off)
    l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
    l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
    for manual in (True, False):
        pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params(manual))
        r = pair.run(check=False)
        h = hashlib.sha256()
        h.update(np.array(r.T16, np.float32).tobytes()); h.update(np.array(r.VCM, np.float64).tobytes())
        no = r.n_outer
        for name in ("n_inner", "n_stable", "n_stable_pts", "LoDmin", "maxBB", "d75", "DT"):
            if hasattr(r, name):
                h.update(np.array(list(getattr(r, name))[:no]).tobytes())
        h.update(np.array([np.array(r.Tk[k], np.float32) for k in range(no)]).tobytes())
        key = "n%d_e%d_%s" % (n, ep, "manual" if manual else "auto")
        out[key] = dict(sha=h.hexdigest()[:16], status=r.status, outer=no, inner=list(r.n_inner[:no]))
        print(key, out[key])
        # the pair-by-pair VCM / ICP entry points on the first case
        pair.close()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
