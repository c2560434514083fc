"""Prints the kernel sequence (duration, gap to the previous kernel) of the last pwicp_pair_run in a rocprofv3
kernel trace: python tools/trace_last_step.py <dir with *kernel_trace.csv>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(k_\w+|__amd_\w+|rocprim|hipcub)", n)
    return m.group(1) if m else n[:30]
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, s in enumerate(seq) if s[0] == "k_scal_init" or (s[0] == "k_front" and (i == 0 or seq[i - 1][0] != "k_xf_front"))]
i0 = idx[-1]
prev_end = seq[i0][1]
busy = 0
for s in seq[i0:]:
    print("%-28s dur %7.1f us  gap %6.1f us" % (s[0], (s[2] - s[1]) / 1e3, (s[1] - prev_end) / 1e3))
    prev_end = max(prev_end, s[2]); busy += s[2] - s[1]
print("span %.1f us, kernel time %.1f us" % ((seq[-1][2] - seq[i0][1]) / 1e3, busy / 1e3))
