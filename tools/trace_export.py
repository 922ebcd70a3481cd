#!/usr/bin/env python
"""Diagnostic: the kernel launches of the last `window` seconds of a rocprofv3 --kernel-trace rocpd database as a compact
tab-separated table (start ms, duration ms, stream, queue, grid, workgroup, lds, name) so that a step's timeline can be read
away from the GPU box.  usage: trace_export.py <results.db> <window seconds> [min_us] > step.tsv"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in ("start", "end", "stream_id", "queue_id", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x", "lds_size", "scratch_size", "name") if c in cols]
rows = db.execute("select %s from kernels order by start" % ", ".join(want)).fetchall()
t_end = max(r[1] for r in rows)
W0 = t_end - int(float(sys.argv[2]) * 1e9)
mn = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0.0
print("# columns of the kernels view: " + " ".join(cols))
print("\t".join(["start_ms", "dur_ms"] + want[2:]))
for r in rows:
    if r[0] < W0 or r[1] - r[0] < mn:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r[-1])
    name = re.sub(r"^void ", "", name)[:70]
    print("\t".join(["%.3f" % ((r[0] - W0) / 1e6), "%.3f" % ((r[1] - r[0]) / 1e6)] + [str(x) for x in r[2:-1]] + [name]))
