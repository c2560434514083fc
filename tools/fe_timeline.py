"""Where the wall time of ONE device front end goes: kernel time per kernel and the gaps between consecutive launches, grouped by
(previous -> next), from a rocprofv3 kernel trace of `bench.py --workload frontend`:  python tools/fe_timeline.py <trace dir>"""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
starts = [i for i, (s, e, n) in enumerate(rows) if 'k_knn_lds' in n] + [len(rows)]
segs = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
segs = [s for s in segs if sum('k_fus_run' in n for _, _, n in s) > 50]
seg = max(segs, key=lambda s: sum(e - b for b, e, n in s)) if len(sys.argv) < 3 else segs[int(sys.argv[2])]
span = (seg[-1][1] - seg[0][0]) / 1e6
ksum = sum(e - s for s, e, n in seg) / 1e6
print("front ends in the trace: %d; the last one: %d launches, span %.1f ms, kernel time %.1f ms, gaps %.1f ms" % (len(segs), len(seg), span, ksum, span - ksum))
gap_by = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
    g = (s1 - e0) / 1e3
    if g > 0:
        v = gap_by[short(n0) + " -> " + short(n1)]
        v[0] += 1; v[1] += g
        hist[min(int(g // 10) * 10, 200)] += 1
print("gap histogram (us, count):", sorted(hist.items()))
for k, v in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:20]:
    print("%-72s n %4d  total %7.2f ms  mean %6.1f us" % (k[:72], v[0], v[1] / 1e3, v[1] / v[0]))
kd = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg:
    v = kd[short(n)]; v[0] += 1; v[1] += (e - s) / 1e3
print()
for k, v in sorted(kd.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%-40s n %4d total %7.2f ms mean %7.1f us" % (k[:40], v[0], v[1] / 1e3, v[1] / v[0]))
