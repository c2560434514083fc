#!/bin/bash
# usage: gpurun_prof.sh <tag>  — pytest gpu (quick subset), bench, rocprof kernel stats
TAG=${1:-x}
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/pytest_gpu_$TAG.log
python bench.py --no-cpu-baseline --labels grid > gpurun_out/bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --labels grid > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/pytest_gpu_$TAG.log; grep -h "{" gpurun_out/bench_$TAG.log | tail -1
