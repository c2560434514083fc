#!/bin/bash
# Counters of the front end's fusion kernels (k_fus_run above all) on one 1 M-point cloud: one rocprofv3 --pmc pass per group,
# summed over all launches of a kernel.  usage: fus_pmc.sh TAG [ENV=VAL ...]   -> gpurun_out/fpmc_TAG.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/fpmc_$TAG.txt
: > $OUT
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
         "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $G --output-format csv -d $R/gpurun_out/fpmc_$TAG/$N -o t -- python $R/bench.py --workload frontend --steps 1 > $R/gpurun_out/fpmc_$TAG.$N.log 2>&1
  F=$(find $R/gpurun_out/fpmc_$TAG/$N -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "group [$G]: no output ($(tail -1 $R/gpurun_out/fpmc_$TAG.$N.log | cut -c1-160))" >> $OUT; continue; fi
  python - "$F" >> $OUT <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"k_(fus|ref|knn|fe)_\w+(<[^>]*>)?", n)
    if not m: continue
    short = m.group(0)
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(short, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    print(k, " ".join("%s=%.5g" % (c, v) for c, v in sorted(d.items())), "launches=%d" % max(cnt[(k, c)] for c in d))
PY
  rm -rf $R/gpurun_out/fpmc_$TAG/$N
done
cat $OUT
