#!/bin/bash
# copies the summaries of gpurun_out/prof_<TAG> (tools/collect_profiles.sh TAG, merged back by gpurun) into profiles/ as <TAG>_*
TAG=${1:-r04}
S=gpurun_out/prof_$TAG; D=profiles
cp $S/trace/bench_kernel_stats.csv $D/${TAG}_bench_kernel_stats.csv
cp $S/trace_frontend/fe_kernel_stats.csv $D/${TAG}_frontend_kernel_stats.csv
cp $S/pmc_summary.csv $D/${TAG}_pmc_summary.csv
cp $S/traffic.json $D/${TAG}_traffic.json
cp $S/traffic.json $D/traffic_latest.json
cp $S/last_step_timeline.txt $D/${TAG}_last_step_timeline.txt
cp $S/real_pair_timeline.txt $D/${TAG}_real_pair_timeline.txt
[ -s $S/frontend_timeline.txt ] && grep -v "^run [0-9]" $S/frontend_timeline.txt > $D/${TAG}_frontend_timeline.txt
for f in final:bench_line frontend:bench_frontend_line series:bench_series_line gpus2:bench_line_gpus2_single_device; do
  grep '^{' $S/bench_${f%%:*}.log | tail -1 > $D/${TAG}_${f##*:}.json
done
for f in rehearsal_2ranks_4cpus taskset_2cpus_bench; do [ -s $S/$f.json ] && cp $S/$f.json $D/${TAG}_$f.json; done
ls -la $D | grep $TAG
