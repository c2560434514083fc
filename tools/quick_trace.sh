#!/bin/bash
# GPU box: loop parity tests, a plain bench line, and the kernel timeline of the last step (tools/trace_last_step.py).
# Output: gpurun_out/qt_$TAG/
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/qt_$TAG
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > $OUT/parity.log
python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 > $OUT/bench_trace.log 2>&1
cd $R
python tools/trace_last_step.py $OUT/trace > $OUT/timeline.txt 2>&1
cat $OUT/parity.log; tail -1 $OUT/bench.log | cut -c1-220; cat $OUT/timeline.txt
