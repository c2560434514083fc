#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench -o $tag -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_bench_$tag.log 2>&1
done
ls $R/gpurun_out/pmc_bench
