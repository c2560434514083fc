# A/B of an environment switch on the bench pair, same box: bash tools/ab_env.sh VAR v1 v2 ...   (kernel averages + bench line per value)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  echo "== $VAR=$v"
  python bench.py --no-cpu-baseline --series-epochs 0 --pairs-in-flight 0 2>/dev/null | tail -1 | cut -c1-200
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_${VAR}_$v -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 > /dev/null 2>&1)
  python tools/kavg.py gpurun_out/ab_${VAR}_$v
done
