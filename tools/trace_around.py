#!/usr/bin/env python
"""Diagnostic: the kernels around the largest idle gaps of a rocprofv3 --kernel-trace CSV (argv[1] dir, argv[2] window seconds, argv[3] how many gaps)."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]) for r in csv.DictReader(open(f)))
t_end = max(k[1] for k in K); W0 = t_end - int(float(sys.argv[2]) * 1e9)
K = [k for k in K if k[0] >= W0]
gaps = []; cur_end = K[0][1]
for i, (s, e, n) in enumerate(K[1:], 1):
    if s > cur_end: gaps.append((s - cur_end, i))
    cur_end = max(cur_end, e)
for g, i in sorted(gaps, reverse=True)[:int(sys.argv[3])]:
    print("---- %.2f ms idle" % (g / 1e6))
    for j in range(max(0, i - 4), min(len(K), i + 4)):
        print("   %s %9.2f -> %9.2f  %s" % (">>" if j == i else "  ", (K[j][0] - W0) / 1e6, (K[j][1] - W0) / 1e6, K[j][2]))
