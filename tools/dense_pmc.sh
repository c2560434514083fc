#!/bin/bash
# SQ counters of the dense-search kernels (dense_variants worker); usage: dense_pmc.sh TAG [ENV=VAL ...]
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
PMC=${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY}      # (PMC="...": another group)
rocprofv3 --pmc $PMC --output-format csv -d $R/gpurun_out/dpmc_$TAG -o t -- python $R/gpurun_out/dprof_worker.py > $R/gpurun_out/dpmc_$TAG.log 2>&1
cd $R
F=$(find gpurun_out/dpmc_$TAG -name "*counter_collection.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_nn_dense" not in n: continue
    short = n.split("(")[0].split("::")[-1][:28]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())), "n=%d" % len(next(iter(d.values()))))
PY
