#!/bin/bash
# per-kernel durations of the dense-search replay (tools/dense_variants.py worker) under rocprofv3; usage: dense_prof.sh TAG [ENV=VAL ...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
python - > $R/gpurun_out/dprof_worker.py <<'PY'
import re
src = open("/root/repo/tools/dense_variants.py").read()
m = re.search(r"WORKER = r'''(.*?)''' % ROOT", src, re.S)
print(m.group(1).replace("%r", repr("/root/repo")))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dprof_$TAG -o t -- python $R/gpurun_out/dprof_worker.py > $R/gpurun_out/dprof_$TAG.log 2>&1
cd $R
F=$(find gpurun_out/dprof_$TAG -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_nn_dense", "k_select", "k_icp", "k_front", "k_transform", "k_classify", "k_compact", "k_vcm")):
        import re
        mm = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", n)
        short = (mm.group(1) + (mm.group(2) or "")) if mm else n[:40]
        print("%-42s calls %5s avg %9.1f us  min %8.1f max %8.1f" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep "^DV" gpurun_out/dprof_$TAG.log | cut -c1-200
