cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_front; mkdir -p $OUT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --output-format csv -d $OUT -o g$i -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-inner-timing --series-epochs 0 --pairs-in-flight 0 > $OUT/g$i.log 2>&1
  echo "group $i rc $?"
done
cd $R
python - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('gpurun_out/pmc_front/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for k in ('k_xf_front', 'k_front', 'k_patch_normals', 'k_classify_icp0', 'k_nn_dense_disc'):
            if k + '<' in n or k + '(' in n:
                a = acc[(k, r['Counter_Name'])]; a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(acc.items()):
    print("%-18s %-24s n %3d mean %14.1f" % (k, c, n, v / n))
P
