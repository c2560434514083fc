#!/bin/bash
# kernel statistics of the device front end (1 M points); usage: fe_prof.sh [points]
R=$GRAFT_REPO_ROOT
N=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
PWICP_FRONTEND=device rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/feprof -o t -- python $R/tools/fe_check.py $N > $R/gpurun_out/feprof.log 2>&1
cd $R
F=$(find gpurun_out/feprof -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-60s calls %6s total %10.3f ms avg %9.1f us  %5s%%" % ((__import__("re").search(r"(k_\w+(<[^>]*>)?)", r["Name"]) or [r["Name"][:60]]*2)[1], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
          float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
