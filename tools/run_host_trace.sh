#!/bin/bash
# GPU box: kernel timeline + host-side stamps of one pwicp_pair_run with and without the dense-NN events
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for F in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ht_$F -o t -- python $R/tools/host_trace.py $F 2>&1 | grep -E "profiling flags|host\]" | tail -25
  python $R/tools/trace_last_step.py $R/gpurun_out/ht_$F
done
