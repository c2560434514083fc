#!/usr/bin/env python3
"""GPU busy time of the end-to-end series of bench.py from a rocprofv3 kernel trace (CSV): the union of the kernel intervals in
bins of 10 ms - how much of the series' wall time the device is idle, and which kernels own the busy time.
usage: series_busy.py <dir with *kernel_trace.csv>"""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"k_\w+", n)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else n[:30]))
ev.sort()
t0 = ev[0][0]
BIN = 10_000_000
bins = collections.defaultdict(lambda: [0, 0, collections.Counter()])   # union ns, sum ns, per kernel
cur_end = 0
for s, e, n in ev:
    b = (s - t0) // BIN
    bins[b][1] += e - s
    bins[b][2][n] += e - s
    us = max(s, cur_end)
    if e > us:
        a = us                                  # (a union interval is split over the bins it crosses)
        while a < e:
            bb = (a - t0) // BIN
            z = min(e, t0 + (bb + 1) * BIN)
            bins[bb][0] += z - a
            a = z
        cur_end = e
for b in sorted(bins):
    u, sm, c = bins[b]
    top = ", ".join("%s %.1f" % (k, v / 1e6) for k, v in c.most_common(3))
    print("%6d ms  busy %5.1f %%  overlap x%.2f   %s" % (b * 10, 100.0 * u / BIN, sm / max(u, 1), top))
