#!/bin/bash
# Vector-memory-pipe counters (TA / TD / TCP) of the dense-search kernels on the dense_variants worker (first Stage-1 launch of
# the synthetic 1 M-point pair + 20 replays).  One rocprofv3 --pmc pass per group, nothing else traced.
# (TA_*_WAVEFRONTS, TA_*_STALLED_BY_*, TD_*: rocprofv3 of this image aborts on them (signal 6 after a 5-minute hang) - left out.)
# usage: dense_ta_pmc.sh TAG [ENV=VAL ...]   -> gpurun_out/dta_TAG.txt
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
rocprofv3 --list-avail > $R/gpurun_out/avail_counters.txt 2>&1
OUT=$R/gpurun_out/dta_$TAG.txt
: > $OUT
for G in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $G --output-format csv -d $R/gpurun_out/dta_$TAG/$N -o t -- python $R/gpurun_out/dprof_worker.py > $R/gpurun_out/dta_$TAG.$N.log 2>&1
  F=$(find $R/gpurun_out/dta_$TAG/$N -name "*counter_collection.csv" | head -1)
  if [ -z "$F" ]; then echo "group [$G]: no output ($(tail -1 $R/gpurun_out/dta_$TAG.$N.log | cut -c1-160))" >> $OUT; continue; fi
  python - "$F" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_nn_dense" not in n: continue
    import re
    short = re.search(r"k_nn_dense\w*", n).group(0)
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())), "n=%d" % len(next(iter(d.values()))))
PY
done
cat $OUT
grep -o "TA_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" $R/gpurun_out/avail_counters.txt | sort -u | tr '\n' ' ' | head -c 6000
