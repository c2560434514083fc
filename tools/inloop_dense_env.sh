#!/bin/bash
# the dense 1-NN launch INSIDE the loop under environment settings: inloop_dense_env.sh "ENV=VAL ..." ["ENV=VAL ..." ...]   ("-": defaults)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1)); D=$R/gpurun_out/inloop_env_$i; rm -rf $D
  E=""; [ "$v" != "-" ] && E="$v"
  env $E timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 --large-points 0 --roofline-steps 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f' % d['ms_per_step'], end='  ')"
  F=$(find $D -name "*kernel_stats.csv" | head -1)
  printf "%-40s " "$v"; grep "k_nn_dense_disc" $F | awk -F, '{printf "k_nn_dense_disc calls %s avg %.2f us min %.2f\n", $(NF-6), $(NF-4)/1000, $(NF-2)/1000}'
done
