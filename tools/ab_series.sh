# A/B of an environment switch on the end-to-end series line (8 x 1 M-point PCD files -> transforms), same box:
#   bash tools/ab_series.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --series-epochs 8 2>/dev/null | tail -1 | python -c "
import sys, json
s = json.loads(sys.stdin.read())['series_end_to_end']
print('$VAR=$v', s['value'], 'pairs/s  wall', s['wall_s'], s['rank0_stage_wall_ms'])"
done
