#!/bin/bash
# A/B of library build variants (tools/build_variant.sh): usage ab_lib.sh "ENV=VAL ..." name [name ...]   ("base" = the regular build)
ENVS=$1; shift
for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$GRAFT_REPO_ROOT/piecewise-icp_amd/variants/libpwicp_$v.so
  echo "== $v $ENVS"
  env $ENVS PWICP_LIB=$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 --large-points 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step',d['ms_per_step'],'dense_us',r['avg_launch_us'],'kbar',r['kbar'], 'outer', d['config']['outer_iterations'], 'corr', d['config']['correspondences_per_step'])"
done
