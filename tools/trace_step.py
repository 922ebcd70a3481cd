#!/usr/bin/env python
"""Diagnostic: the kernel launches of the last `window` seconds of a rocprofv3 --kernel-trace rocpd database, in start order, that last at least `min_ms`
(argv[1]: results.db, argv[2]: window in seconds from the end, argv[3]: min_ms).  Shows which launches of a step are long on their own."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
K = [(s, e, re.sub(r"\(anonymous namespace\)::", "", n)[:80]) for s, e, n in db.execute("select start, end, name from kernels order by start")]
t_end = max(k[1] for k in K)
W0 = t_end - int(float(sys.argv[2]) * 1e9)
mn = float(sys.argv[3]) * 1e6
for s, e, n in K:
    if s >= W0 and e - s >= mn:
        print("%9.2f %8.2f  %s" % ((s - W0) / 1e6, (e - s) / 1e6, n))
