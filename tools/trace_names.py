#!/usr/bin/env python
"""Diagnostic: durations of the kernels whose name contains argv[2] in a rocprofv3 kernel trace (last argv[3] seconds)."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:40], r["Queue_Id"]) for r in csv.DictReader(open(f))]
t_end = max(k[1] for k in K); W0 = t_end - int(float(sys.argv[3]) * 1e9)
for s, e, n, q in sorted(K):
    if s >= W0 and sys.argv[2] in n and e - s > 2e6:
        print("%8.0f -> %8.0f  %7.1f ms  q=%s %s" % ((s - W0) / 1e6, (e - W0) / 1e6, (e - s) / 1e6, q, n))
