# A/B of an environment switch on the loop of two of the reference's own pairs (tools/real_pair_loop.py): bash tools/ab_real.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v python tools/real_pair_loop.py 2 20 | tail -2
  env $VAR=$v python tools/real_pair_loop.py 12 20 | tail -2
done
