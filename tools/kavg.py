"""Average duration per kernel name of a rocprofv3 kernel-trace CSV (loop kernels only): python tools/kavg.py <trace dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for k in ("k_xf_front", "k_front", "k_classify_icp0", "k_nn_dense_disc", "k_nn_dense_far", "k_icp_iter", "k_xf_vcm"):
        if k + "(" in n or k + "<" in n:
            d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
tot = 0
for k, v in d.items():
    v2 = sorted(v)
    print("%-18s n %3d  mean %6.2f  median %6.2f us" % (k, len(v), sum(v) / len(v), v2[len(v) // 2]))
