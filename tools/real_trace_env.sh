#!/bin/bash
# kernel timeline of the last run of tools/real_pair_loop.py E under switches: real_trace_env.sh E "ENV=VAL ..." [more settings]
E=$1; shift
for v in "$@"; do
  [ "$v" == "-" ] && v="PWICP_NOP=1"
  OUT=$GRAFT_REPO_ROOT/gpurun_out/rt_tmp; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/tools/real_pair_loop.py $E 6 > $OUT/log.txt 2>&1)
  echo "== Epoch $E  $v"; tail -1 $OUT/log.txt
  python $GRAFT_REPO_ROOT/tools/trace_last_step.py $OUT | grep -E "dense|span"
done
