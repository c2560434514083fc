# the loop on two of the reference's own pairs (Epoch_002 / Epoch_012 -> Epoch_001): wall per run, then the kernel timeline of the
# last run of each under rocprofv3 --kernel-trace -> gpurun_out/real_trace/timeline.txt   (usage: real_pair_trace.sh [ENV=VAL ...])
for kv in "$@"; do export "$kv"; done
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/real_trace; mkdir -p $OUT
: > $OUT/timeline.txt
for E in ${REAL_EPOCHS:-2 12}; do
  python tools/real_pair_loop.py $E 30 | tee -a $OUT/timeline.txt
  rm -rf $OUT/e$E
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/e$E -o t -- python $GRAFT_REPO_ROOT/tools/real_pair_loop.py $E 6 > $OUT/log$E.txt 2>&1)
  echo "-- kernel timeline of the last run, Epoch_$(printf %03d $E) -> Epoch_001 (rocprofv3 --kernel-trace)" >> $OUT/timeline.txt
  python tools/trace_last_step.py $OUT/e$E >> $OUT/timeline.txt
  echo >> $OUT/timeline.txt
done
cat $OUT/timeline.txt
