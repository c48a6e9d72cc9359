#!/usr/bin/env python3
"""Start, end and queue of the kernels around the middle of a rocprofv3 kernel trace (which kernels overlap):
    python tools/trace_timeline.py <kernel_trace.csv> [kernels to print]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
mid = len(rows) // 2
base = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + count]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ldpc::", "")[:28]
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-30s start %9.3f  end %9.3f  duration %7.3f ms  queue %s" % (name, (a - base) / 1e6, (b - base) / 1e6, (b - a) / 1e6, r["Queue_Id"]))
