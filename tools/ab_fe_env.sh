#!/bin/bash
# front end of one 1 M-point cloud under run-time switches: each argument one "ENV=VAL ..." setting ("-": defaults); prints the line's
# ms per cloud, label identity, the first rounds' statistics and the fusion time of the timed clouds
for D in "$@"; do
  [ "$D" == "-" ] && D="PWICP_NOP=1"
  echo "== $D"
  env $D PWICP_TRACE=1 python bench.py --workload frontend --steps 4 --no-cpu-baseline 2> /tmp/fe.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], 'identical', d['labels_identical_to_serial_passes'])"
  grep -E "round [0-9]+:" /tmp/fe.txt | tail -10 | sed -e 's/\[pwicp front end\/dev\]//' | cut -c1-110 | head -4
  grep -E "dev\] fusion" /tmp/fe.txt | tail -3
done
