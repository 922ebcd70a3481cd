#!/usr/bin/env python
"""Diagnostic: the kernels in flight around every launch of one kernel in a rocprofv3 --kernel-trace CSV (argv[1] dir, argv[2] substring of the kernel's name, argv[3] window seconds)."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]) for r in csv.DictReader(open(f)))
t_end = max(k[1] for k in K); W0 = t_end - int(float(sys.argv[3]) * 1e9)
K = [k for k in K if k[0] >= W0]
for i, (s, e, n) in enumerate(K):
    if sys.argv[2] not in n: continue
    print("---- %s: %.2f ms" % (n, (e - s) / 1e6))
    for j in range(max(0, i - 5), min(len(K), i + 6)):
        print("   %s %9.2f -> %9.2f  %s" % (">>" if j == i else "  ", (K[j][0] - W0) / 1e6, (K[j][1] - W0) / 1e6, K[j][2]))
