#!/usr/bin/env python3
"""Per-kernel durations of the last decode in a rocprofv3 kernel trace (the kernels from the last load_lambda_kernel
to the last unpack_hard_kernel): python tools/trace_gaps.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
s = [i for i, k in enumerate(ks) if "load_lambda" in k[0]][-1]
e = [i for i, k in enumerate(ks) if "unpack_hard" in k[0]][-1]
for i in range(s, min(s + 10, e + 1)):
    n, a, b = ks[i]
    print("%-72s %8.2f us  gap before %6.2f us" % (n[:72], (b - a) / 1e3, (a - ks[i - 1][2]) / 1e3 if i > s else 0.0))
d = collections.defaultdict(list)
for i in range(s, e + 1):
    d[ks[i][0][:72]].append((ks[i][2] - ks[i][1]) / 1e3)
print("by kernel:")
for k, v in d.items():
    print("  %-72s n=%3d  mean %7.2f  max %7.2f  min %6.2f us" % (k, len(v), sum(v) / len(v), max(v), min(v)))
busy = sum(b - a for _, a, b in ks[s:e + 1])
print("kernels %d, busy %.1f us, span %.1f us (the profiler widens the gaps)" % (e - s + 1, busy / 1e3, (ks[e][2] - ks[s][1]) / 1e3))
