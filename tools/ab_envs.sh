#!/bin/bash
# A/B of run-time switches: each argument is one "ENV=VAL ENV=VAL" setting ("-" = defaults)
for v in "$@"; do
  [ "$v" == "-" ] && v="PWICP_NOP=1"
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step',d['ms_per_step'],'dense_us',r['avg_launch_us'],'kbar',r['kbar'], 'outer', d['config']['outer_iterations'], 'corr', d['config']['correspondences_per_step'])"
done
