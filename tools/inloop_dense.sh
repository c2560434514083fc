#!/bin/bash
# the dense 1-NN launch INSIDE the loop (rocprofv3 kernel trace of bench.py's steps) under library build variants: inloop_dense.sh name [name ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$R/piecewise-icp_amd/variants/libpwicp_$v.so
  rm -rf $R/gpurun_out/inloop_$v
  PWICP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/inloop_$v -o t -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 --large-points 0 --roofline-steps 1 > /dev/null 2>&1
  F=$(find $R/gpurun_out/inloop_$v -name "*kernel_stats.csv" | head -1)
  printf "%-10s " $v; grep "k_nn_dense_disc" $F | awk -F, '{printf "k_nn_dense_disc calls %s avg %.2f us min %.2f max %.2f\n", $(NF-6), $(NF-4)/1000, $(NF-2)/1000, $(NF-1)/1000}'
done
