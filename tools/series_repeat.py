#!/usr/bin/env python3
"""End-to-end series (8 x 1 M-point PCD files -> transforms, one GPU) repeated in one process through FRESH Series objects (the
shared target too is read, preprocessed and segmented again every time; only the parked contexts are reused): the first (cold)
wall and the warm walls.  usage: series_repeat.py [repeats]   (run on the GPU box; environment switches apply)"""
import os, sys, tempfile, time, shutil
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_ + '/piecewise-icp_amd')
# (GPU_MAX_HW_QUEUES: left to the library, which asks for eight when the variable is not set)
import pwicp_amd as P
from pwicp_amd import synth
from pwicp_amd.pcd import write_pcd_binary
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n, E, r = int(os.environ.get("SERIES_N", 1000000)), int(os.environ.get("SERIES_E", 8)), 0.005
d = tempfile.mkdtemp(dir="/dev/shm")
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
pre = os.environ.get("SERIES_REPEAT_PRE", "")       # what the process did before: "ctx" - four contexts created and closed;
if pre:                                             # "pairs" - a registration on each of them, side by side (bench.py's pairs_side_by_side)
    import threading
    ctxs = [P.Context(0) for _ in range(4)]
    if pre == "pairs":
        m = 200000
        tg, _ = synth.make_tile(m, r); sr, _ = synth.make_source(m, r, epoch=1)
        c0 = tg.mean(axis=0); tg = (tg - c0).astype(np.float32); sr = (sr - c0).astype(np.float32)
        l1, n1 = synth.grid_labels(tg, 10 * r); l2, n2 = synth.grid_labels(sr, 10 * r)
        prm = P.Params(r, r, 10 * r, 10 * r, 1, 10 * r, 0.8 * r)
        prs = [P.Pair(c, tg, l1, n1, sr, l2, n2, prm) for c in ctxs]
        def loop(i):
            for _ in range(20):
                prs[i].reset(); prs[i].run()
        th = [threading.Thread(target=loop, args=(i,)) for i in range(4)]
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        for pr in prs: pr.close()
    if pre != "ctx_keep":
        for c in ctxs: c.close()
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)
ts, ok = [], True
for rep in range(reps):
    series = P.Series(cfg, 0, E + 1, 0, 0.75, 0)
    t0 = time.perf_counter()
    recs = series.run_pairs(list(range(E)))
    ts.append(time.perf_counter() - t0)
    ok = ok and bool((recs["status"] == 0).all())
    series.close()
os.dup2(saved, 1)
w = sorted(ts[1:])
sys.stderr.write("RESULT cold %.3f  warm min %.3f median %.3f max %.3f  ok %s  %s" % (ts[0], w[0], w[len(w) // 2], w[-1], ok, [round(x, 3) for x in ts]) + "\n")
shutil.rmtree(d)
