#!/usr/bin/env python
"""Diagnostic: the device-to-device copies (__amd_rocclr_copyBuffer dispatches) of the last step of a rocprofv3 --kernel-trace rocpd database that take longest,
each with the kernels in front of and behind it (argv[1]: results.db, argv[2]: window in seconds from the end, argv[3]: how many)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
K = [(s, e, re.sub(r"\(anonymous namespace\)::", "", n)[:70]) for s, e, n in db.execute("select start, end, name from kernels order by start")]
t_end = max(k[1] for k in K)
W0 = t_end - int(float(sys.argv[2]) * 1e9)
K = [k for k in K if k[0] >= W0]
cp = [(e - s, i) for i, (s, e, n) in enumerate(K) if "copyBuffer" in n or "fillBuffer" in n]
print("%d copies / fills in the window, %.2f ms in all" % (len(cp), sum(c[0] for c in cp) / 1e6))
for d, i in sorted(cp, reverse=True)[:int(sys.argv[3])]:
    print("---- %.3f ms  %s" % (d / 1e6, K[i][2]))
    for j in range(max(0, i - 2), min(len(K), i + 3)):
        print("   %s %9.2f -> %9.2f  %s" % (">>" if j == i else "  ", (K[j][0] - W0) / 1e6, (K[j][1] - W0) / 1e6, K[j][2]))
