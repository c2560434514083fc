cd $GRAFT_REPO_ROOT
python tools/real_pair_loop.py 2 20 | tail -2
python tools/real_pair_loop.py 12 20 | tail -2
OUT=$GRAFT_REPO_ROOT/gpurun_out/real_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $GRAFT_REPO_ROOT/tools/real_pair_loop.py 2 6 > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_last_step.py $OUT
