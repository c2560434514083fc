"""Summarises the output of tools/collect_profiles.sh: per-kernel mean of every PMC counter (pmc_summary.csv) and the
HBM bytes per launch of the dense 1-NN kernel (traffic.json).  FETCH_SIZE / WRITE_SIZE are reported in units that
must be calibrated (MI355X_MICROARCH.md, HBM section): the calibration kernel is k_transform_all, whose byte count
per launch is known by construction (every point of cloud2, of the centroid/boundary array and of the patch array is
read once and written once, 16 bytes each)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]
def short(n):
    m = re.search(r"(k_\w+|__amd_\w+|rocprim|hipcub)", n)
    return m.group(1) if m else n[:40]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
rows = sorted((k[0], k[1], v[1], v[0] / v[1]) for k, v in acc.items())
with open(os.path.join(out, "pmc_summary.csv"), "w") as o:
    o.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for r in rows:
        o.write("%s,%s,%d,%.4f\n" % r)
mean = {(k, c): m for k, c, _, m in rows}
# workload sizes from the bench line
line = [l for l in open(os.path.join(out, "bench_plain.log")) if l.startswith("{")][-1]
b = json.loads(line)
n2 = b["config"]["points_per_cloud"]
# (the bench line of THESE sources is written afterwards, by the run that reads this file: profiles/r0N_bench_line.json is the
# record of the in-run roofline, not a block in here that predates it)
res = {"bench_plain": {k: b[k] for k in ("value", "ms_per_step")}}
dense = "k_nn_dense_disc" if ("k_nn_dense_disc", "FETCH_SIZE") in mean else "k_nn_dense_direct"
res["kernel"] = dense
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
res["kernel_source_sha256"] = _bench.kernel_source_hash()
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
    if (dense, c) in mean:
        res.setdefault("dense_sq_counters", {})[c] = mean[(dense, c)]
if (dense, "FETCH_SIZE") in mean:
    f_dense, w_dense = mean[(dense, "FETCH_SIZE")], mean.get((dense, "WRITE_SIZE"), 0.0)
    f_cal, w_cal = mean.get(("k_transform_all", "FETCH_SIZE")), mean.get(("k_transform_all", "WRITE_SIZE"))
    res["raw_counter_means"] = {"dense_FETCH_SIZE": f_dense, "dense_WRITE_SIZE": w_dense,
                                "transform_FETCH_SIZE": f_cal, "transform_WRITE_SIZE": w_cal}
    # k_transform_all reads exactly what it writes (16 B per point in, 16 B per point out): its WRITE_SIZE (KiB, matches the
    # byte count computed from the array sizes) calibrates FETCH_SIZE -> the x2 of the guide for 16-B/lane streaming reads
    factor = (w_cal / f_cal) if f_cal else 2.0
    fetch_b = f_dense * 1024.0 * factor
    write_b = w_dense * 1024.0
    dur_us = b["roofline"]["avg_launch_us"]
    res["k_nn_dense_bytes_per_launch"] = int(round(fetch_b + write_b))
    res["fetch_bytes_corrected"] = int(round(fetch_b))
    res["write_bytes"] = int(round(write_b))
    res["fetch_correction_factor"] = round(factor, 4)
    res["hbm_gbs_at_in_run_duration"] = round((fetch_b + write_b) / (dur_us * 1e-6) / 1e9, 1)
    res["calibration"] = ("k_transform_all: FETCH_SIZE %.1f vs WRITE_SIZE %.1f KiB per launch for equal read and written byte counts "
                          "-> FETCH_SIZE x %.3f (MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-B requests at 64 B for 16-B/lane "
                          "loads); WRITE_SIZE taken as is" % (f_cal, w_cal, factor))
    res["source"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 5 --warmup 1 (tools/collect_profiles.sh)"
# the large pair of bench.py's second roofline record: the same two counters, collected on `--points <large>` runs (pmc_large/)
accL = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "pmc_large", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        accL[k][0] += float(r["Counter_Value"]); accL[k][1] += 1
if (dense, "FETCH_SIZE") in accL and "fetch_correction_factor" in res:
    fL = accL[(dense, "FETCH_SIZE")][0] / accL[(dense, "FETCH_SIZE")][1]
    wL = accL[(dense, "WRITE_SIZE")][0] / accL[(dense, "WRITE_SIZE")][1] if (dense, "WRITE_SIZE") in accL else 0.0
    res["large_points"] = int(open(os.path.join(out, "large_points.txt")).read().split()[0])
    res["k_nn_dense_bytes_per_launch_large"] = int(round(fL * 1024.0 * res["fetch_correction_factor"] + wL * 1024.0))
    res["raw_counter_means"].update({"dense_FETCH_SIZE_large": fL, "dense_WRITE_SIZE_large": wL,
                                     "dense_dispatches_large": accL[(dense, "FETCH_SIZE")][1]})
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
