"""Do registrations of independent pairs overlap on one GPU?  A registration is a chain of ~14 short launches that leaves most of
the chip idle; K pairs (K source epochs against one target: the Direct2Ref mode of the reference's series, R.cpp:89-150) on K
contexts (= streams) from K host threads, against the same K pairs one after the other on one context."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pwicp_amd as P
import _data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ctxs = [P.Context(0) for _ in range(K)]
tgt = _data.pair(n, epoch=1)[0]
l1, n1 = ctxs[0].frontend_segment(tgt, 10 * _data.R, 45, _data.R)
pairs, seq = [], []
for e in range(K):
    src = _data.pair(n, epoch=e + 1)[1]
    l2, n2 = ctxs[0].frontend_segment(src, 10 * _data.R, 45, _data.R)
    pairs.append(P.Pair(ctxs[e], tgt, l1, n1, src, l2, n2, _data.params()))
    seq.append(P.Pair(ctxs[0], tgt, l1, n1, src, l2, n2, _data.params()))
ref = []
for p in seq:
    r = p.run(check=False); ref.append(np.array(r.T16, np.float32).tobytes())
def loop(p, out, i):
    for _ in range(REP):
        p.reset(); r = p.run(check=False)
    out[i] = np.array(r.T16, np.float32).tobytes()
out = [None] * K
t0 = time.perf_counter()
for i, p in enumerate(seq): loop(p, out, i)
t_seq = (time.perf_counter() - t0) / (REP * K)
assert out == ref
print("%d pairs one after the other on one context: %.3f ms per registration" % (K, 1e3 * t_seq))
for k in sorted({2, K}):
    out = [None] * K
    th = [threading.Thread(target=loop, args=(pairs[i], out, i)) for i in range(k)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) / (REP * k)
    assert out[:k] == ref[:k], "concurrent results differ"
    print("%d pairs side by side on %d contexts: %.3f ms per registration (x%.2f), results bit-identical" % (k, k, 1e3 * dt, t_seq / dt))
for p in pairs + seq: p.close()
for c in ctxs: c.close()
