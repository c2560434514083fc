cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_far; rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --output-format csv -d $OUT -o g$i -- python $R/tools/real_pair_loop.py 2 6 > $OUT/g$i.log 2>&1
  echo "group $i rc $?"
done
cd $R
python - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('gpurun_out/pmc_far/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for k in ('k_nn_dense_far', 'k_nn_dense_disc'):
            if k + '<' in n or k + '(' in n:
                a = acc[(k, r['Counter_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(acc.items()):
    print("%-18s %-32s n %3d mean %14.1f" % (k, c, n, v / n))
P
tail -3 $R/gpurun_out/pmc_far/g2.log
