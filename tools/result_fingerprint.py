"""Fingerprint of four registrations (T, VCM, every per-iteration series, the moved source cloud): A/B of builds that must not change a bit\n(PWICP_LIB=<variant> python tools/result_fingerprint.py)."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data
ctx = P.Context(0)
h = hashlib.sha256()
for (n, ep, manual) in ((60000, 1, True), (60000, 3, False), (200000, 2, True), (1000000, 1, True)):
    tgt, src, _ = _data.pair(n, epoch=ep)
    l1, n1 = ctx.frontend_segment(tgt, 10 * _data.R, 45, _data.R)
    l2, n2 = ctx.frontend_segment(src, 10 * _data.R, 45, _data.R)
    pair = P.Pair(ctx, tgt, l1, n1, src, l2, n2, _data.params(manual))
    r = pair.run()
    no = r.n_outer
    for a in (np.array(r.T16, np.float32), np.array(r.VCM, np.float64), np.array(r.n_inner[:no]), np.array(r.n_stable[:no]),
              np.array(r.DTseries[:no + 1], np.float32), np.array(r.maxBB[:no], np.float32), np.array(r.d75[:no], np.float64), pair.download_source()):
        h.update(np.ascontiguousarray(a).tobytes())
    pair.close()
print("FINGERPRINT", h.hexdigest())
