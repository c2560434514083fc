# A/B of environment settings on the front-end line (1 M-point cloud, labels checked against the serial passes), same box:
#   bash tools/ab_frontend.sh "VAR1=a VAR2=b" "VAR1=c" ...
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  env $setting python bench.py --workload frontend --steps 4 2>/dev/null | tail -1 | python -c "
import sys, json
s = json.loads(sys.stdin.read())
print('%-50s %.1f ms per cloud, labels identical to the serial passes: %s' % ('$setting', s['ms_per_step'], s.get('labels_identical_to_serial_passes')))"
done
