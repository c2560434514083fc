#!/bin/bash
# per-kernel totals of the device front end of ONE 1 M-point cloud (the bench's cloud), averaged over REPS runs under rocprofv3 --kernel-trace
# usage: fe_kstats.sh [ENV=VAL ...]   (REAL_EPOCH=n: scan n of the reference's series instead, tests/golden/inputs)
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT; REPS=${REPS:-4}
cat > /tmp/fe_run.py <<PY
import os, sys
sys.path.insert(0, "$R/piecewise-icp_amd"); sys.path.insert(0, "$R/tests")
import pwicp_amd as P, _data, time
ctx = P.Context(0)
import numpy as np
E = int(os.environ.get("REAL_EPOCH", "0"))
if E:
    from pwicp_amd.pcd import read_pcd
    cl = ctx.preprocess(read_pcd("$R/tests/golden/inputs/Epoch_%03d.pcd" % E), 0.005, 14, 5.0)
    cl = (cl - cl.mean(axis=0)).astype(np.float32); sv, sp = 0.05, 0.005
else:
    cl = _data.pair(1000000, epoch=1)[1]; sv, sp = 10 * _data.R, _data.R
print("cloud of %d points" % len(cl))
for i in range($REPS + 1):
    t = time.time(); ctx.frontend_segment(cl, sv, 45, sp); print("run %d %.1f ms" % (i, 1e3 * (time.time() - t)))
PY
rm -rf $R/gpurun_out/fek
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fek -o t -- python /tmp/fe_run.py 2>&1 | grep "^run\|^cloud")
F=$(find $R/gpurun_out/fek -name "*kernel_stats.csv" | head -1)
python - "$F" $REPS <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1]))); reps = int(sys.argv[2]) + 1
tot = 0
for r in rows[:40]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"]); nm = m.group(1) if m else r["Name"][:60]
    ms = float(r["TotalDurationNs"]) / 1e6 / reps; tot += ms
    print("%-56s launches %7.1f  %8.3f ms per cloud  avg %8.1f us" % (nm[:56], int(r["Calls"]) / reps, ms, float(r["AverageNs"]) / 1e3))
print("sum of the listed kernels %.2f ms per cloud" % tot)
PY
python - "$(find $R/gpurun_out/fek -name "*kernel_trace.csv" | head -1)" <<'PY'
# the last cloud's timeline: span, busy, idle, and where the stream idles (gap after kernel A before kernel B)
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"(k_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)", r["Kernel_Name"])).group(1)) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2] == "k_knn_lds"]
E = ev[starts[-1]:]
t0 = E[0][0]; t1 = max(e[1] for e in E)
busy = 0; last_end = t0; prev = "start"; gaps = collections.Counter(); gapn = collections.Counter()
for a, b, n in E:
    if a > last_end: gaps[prev + " -> " + n] += a - last_end; gapn[prev + " -> " + n] += 1
    if b > last_end: busy += b - max(a, last_end); last_end = b; prev = n
print("last cloud: first kernel -> last kernel %.2f ms, busy %.2f ms, idle %.2f ms, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(E)))
for k, v in gaps.most_common(8): print("  idle %-52s %7.3f ms in %4d gaps" % (k, v / 1e6, gapn[k]))
PY
