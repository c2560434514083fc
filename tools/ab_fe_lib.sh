#!/bin/bash
# front end of one 1 M-point cloud under library build variants / switches: ab_fe_lib.sh "ENV=VAL ..." name [name ...]
ENVS=$1; shift
for v in "$@"; do
  L=""; [ "$v" != "base" ] && L=$GRAFT_REPO_ROOT/piecewise-icp_amd/variants/libpwicp_$v.so
  echo "== $v $ENVS"
  env $ENVS PWICP_LIB=$L PWICP_TRACE=1 python bench.py --workload frontend --steps 3 --no-cpu-baseline 2> /tmp/fe_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], 'identical', d['labels_identical_to_serial_passes'])"
  grep -E "round [0-9]+:|dev\] fusion" /tmp/fe_$v.txt | tail -11 | sed -e 's/\[pwicp front end\/dev\]//' | cut -c1-110
done
