cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/qt_5m; mkdir -p $OUT
python bench.py --points 5000000 --steps 10 --warmup 2 --no-cpu-baseline --series-epochs 0 --pairs-in-flight 0 2>/dev/null | tail -1 | cut -c1-260
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --points 5000000 --steps 6 --warmup 2 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 > $OUT/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_last_step.py $OUT/trace
