#!/bin/bash
# Run on the GPU box (gpurun): kernel trace + stats of bench.py, then one rocprofv3 --pmc pass per counter group
# (counters never combined with sys/runtime traces).  Output under gpurun_out/prof_$TAG; summarise with
# tools/summarize_pmc.py and copy the summaries into profiles/.
TAG=${1:-final}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 --large-points 0 --roofline-steps 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/bench_trace.log 2>&1
python $R/tools/trace_last_step.py $OUT/trace > $OUT/last_step_timeline.txt 2>&1
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  N=$(echo $G | cut -d' ' -f1)
  timeout 240 rocprofv3 --pmc $G --output-format csv -d $OUT/pmc -o $N -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 --large-points 0 --roofline-steps 1 > $OUT/pmc_$N.log 2>&1
done
# the same two traffic counters on the LARGE pair of the second roofline record (bench.py roofline_large; 4 M points per cloud)
LARGE=${LARGE_POINTS:-4000000}
for N in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $N --output-format csv -d $OUT/pmc_large -o $N -- python $R/bench.py --points $LARGE --steps 3 --warmup 1 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 --large-points 0 --roofline-steps 1 > $OUT/pmc_large_$N.log 2>&1
done
echo $LARGE > $OUT/large_points.txt
# calibration of FETCH_SIZE / WRITE_SIZE on a kernel with a known byte count (k_transform_all through pwicp_pair_step)
for N in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $N --output-format csv -d $OUT/pmc -o cal_$N -- python $R/tools/pmc_calibration.py > $OUT/pmc_cal_$N.log 2>&1
done
# FETCH_SIZE on 12-byte divergent gathers with a byte count known by construction (tools/gather_calibration.py, k_gather_calibration)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o gcal_FETCH_SIZE -- python $R/tools/gather_calibration.py > $OUT/pmc_gcal_FETCH_SIZE.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc -o gcal_RDREQ -- python $R/tools/gather_calibration.py > $OUT/pmc_gcal_RDREQ.log 2>&1
# the segmentation front end (row f1): kernel statistics of one warm-up + 3 timed clouds of 1 M points
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_frontend -o fe -- python $R/bench.py --workload frontend --steps 3 > $OUT/bench_frontend_trace.log 2>&1
cd $R
timeout 600 python bench.py --no-cpu-baseline --series-epochs 0 --pairs-in-flight 0 --large-points 0 > $OUT/bench_plain.log 2>/dev/null     # (its in-run duration of the dense launch goes into the summary)
timeout 600 python bench.py --workload frontend --steps 5 > $OUT/bench_frontend.log 2>&1
timeout 600 python bench.py --workload series --epochs 4 --points 5000000 > $OUT/bench_series.log 2>&1
# the front end of one cloud: kernels per cloud, where the stream idles (tools/fe_kstats.sh), wall times (tools/fe_time.py)
(bash tools/fe_kstats.sh; echo; python tools/fe_time.py 1000000; python tools/fe_time.py real 2; python tools/fe_time.py real 12) > $OUT/frontend_timeline.txt 2>&1
# the loop on two of the reference's own pairs, with the kernel timeline of their last run
bash tools/real_pair_trace.sh > $OUT/real_pair_timeline.txt 2>&1
# the 2-rank / CPU-share rehearsals of the shared-target series
taskset -c 0-3 python bench.py --gpus 2 --single-device --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-inner-timing --series-epochs 2 --pairs-in-flight 0 2>/dev/null | tail -1 > $OUT/rehearsal_2ranks_4cpus.json
taskset -c 0-1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --pairs-in-flight 0 --large-points 0 2>/dev/null | tail -1 > $OUT/taskset_2cpus_bench.json
# N = 2 rehearsal on this box's one GPU (bench.py starts the ranks itself)
timeout 600 python bench.py --gpus 2 --single-device --backend gloo --steps 5 --warmup 1 --no-cpu-baseline --no-inner-timing --series-epochs 4 --pairs-in-flight 0 > $OUT/bench_gpus2.log 2>/dev/null
python tools/summarize_pmc.py $OUT > $OUT/summary.log 2>&1
cp $OUT/traffic.json profiles/traffic_latest.json      # (on the box's copy: the line below then carries the traffic of THESE sources)
timeout 600 python bench.py > $OUT/bench_final.log 2>/dev/null
tail -5 $OUT/summary.log
