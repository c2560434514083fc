"""When the blocks of the dense search's launch start and end (build: tools/build_variant.sh bt "-DPW_DENSE_BLOCKTRACE" grid; run with
PWICP_LIB=.../variants/libpwicp_bt.so): the launch's ramp, the blocks' lifetimes and its drain.  python tools/dense_blocktrace.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "piecewise-icp_amd"))
import pwicp_amd as P
from pwicp_amd import synth
n = int(os.environ.get("DV_POINTS", "1000000")); r = 0.005
ctx = P.Context(0)
if len(sys.argv) > 2 and sys.argv[1] == "real":          # python tools/dense_blocktrace.py real <epoch>: one of the reference's own pairs
    from pwicp_amd.pcd import read_pcd
    g = os.path.join(ROOT, "tests", "golden", "inputs")
    t = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_001.pcd")), 0.005, 14, 5.0)
    s = ctx.preprocess(read_pcd(os.path.join(g, "Epoch_%03d.pcd" % int(sys.argv[2]))), 0.005, 14, 5.0)
    c = t[:, :3].mean(0); t[:, :3] -= c; s[:, :3] -= c
    l1, n1 = ctx.frontend_segment(t, 0.05, 45, 0.005); l2, n2 = ctx.frontend_segment(s, 0.05, 45, 0.005)
    pair = P.Pair(ctx, t, l1, n1, s, l2, n2, P.Params(0.005, 0.005, 0.05, 0.05, 1, 0.05, 0.004))
else:
    t, _ = synth.make_tile(n, r); s, _ = synth.make_source(n, r, epoch=1); c = t.mean(0)
    t = (t - c).astype(np.float32); s = (s - c).astype(np.float32)
    l1, n1 = synth.grid_labels(t, 10 * r); l2, n2 = synth.grid_labels(s, 10 * r)
    pair = P.Pair(ctx, t, l1, n1, s, l2, n2, P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r))
for _ in range(3):
    pair.reset(); res = pair.run()
L = P.load_library()
buf = (C.c_ulonglong * (8 * 8192))()
L.pwicp_debug_dense_blocktrace(buf, 8 * 8192)
a = np.array(buf, dtype=np.uint64).astype(np.int64).reshape(-1, 8)
a = a[(a[:, 0] > 0) & (a[:, 1] >= a[:, 0])]
a = a[a[:, 0] > a[:, 0].max() - 100000]              # the LAST launch only (entries of earlier, larger launches stay in the buffer)
t0 = a[:, 0].min()
st = (a[:, 0] - t0).astype(np.float64) / 100.0; en = (a[:, 1] - t0).astype(np.float64) / 100.0     # us
life = en - st
print("blocks %d, launch span %.1f us (first start -> last end)" % (len(a), en.max()))
print("starts: 50 %% by %.1f us, 90 %% by %.1f, 99 %% by %.1f, last %.1f" % tuple(np.percentile(st, [50, 90, 99, 100])))
print("ends:   50 %% by %.1f us, 90 %% by %.1f, 99 %% by %.1f, last %.1f" % tuple(np.percentile(en, [50, 90, 99, 100])))
print("lifetime of a block: mean %.1f us, p10 %.1f, p50 %.1f, p90 %.1f, p99 %.1f, max %.1f" % ((life.mean(),) + tuple(np.percentile(life, [10, 50, 90, 99, 100]))))
for lo in range(0, min(int(en.max()) + 1, 400), 2):
    live = int(((st <= lo) & (en > lo)).sum())
    print("  t = %4.0f us: %5d blocks resident" % (lo, live))
# phases of a block as its first thread sees them (stamps 3 .. 7: query there, own row scanned, ball scanned, block's unresolved counted, far queries done)
ph = a[:, 3:8].astype(np.float64)
ok = (ph >= a[:, 0:1]).all(axis=1) & (ph <= a[:, 1:2]).all(axis=1)
if ok.any():
    rel = (ph[ok] - a[ok, 0:1]) / 100.0
    names = ["query + own row words there", "own row scanned (phase A)", "ball scanned (phase B)", "barrier: unresolved counted", "far queries done"]
    print("phases of a block, thread 0 (mean us after the block's start; %d blocks whose thread 0 had a stable query): " % int(ok.sum()) +
          " | ".join("%s +%.2f" % (nm, rel[:, k].mean()) for k, nm in enumerate(names)) + " | end +%.2f" % life[ok].mean())
late = np.argsort(-en)[:10]
print("the ten blocks that end last: " + ", ".join("#%d (xcd %d) %.1f -> %.1f" % (i, int(a[i, 2]), st[i], en[i]) for i in late))
