#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for ct in 0.5 1 2 4; do
  echo "CT factor $ct"; PWICP_CT_CELL_FACTOR=$ct python bench.py --no-cpu-baseline --steps 20 2>&1 | grep "{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_outer_iteration'], d['ms_per_inner_iteration'], d['roofline']['avg_launch_us'], d['roofline']['kbar'], d['value'])"
done
