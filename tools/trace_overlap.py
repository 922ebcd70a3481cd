#!/usr/bin/env python
"""Diagnostic over a rocprofv3 --kernel-trace CSV: how much of the time do kernels of different queues actually overlap?"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print("columns", list(rows[0].keys()))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
K.sort()
t_end = K[-1][1]
# last 3.2 s = the timed step (plus a little)
W0 = t_end - int(3.2e9)
K = [k for k in K if k[0] >= W0]
by_q = collections.defaultdict(list)
for k in K:
    by_q[(k[3], k[4])].append(k)
for q, v in by_q.items():
    busy = sum(e - s for s, e, *_ in v)
    print("queue/stream", q, "kernels", len(v), "busy ms %.0f" % (busy / 1e6), "first at %.0f last end %.0f" % ((v[0][0] - W0) / 1e6, (max(x[1] for x in v) - W0) / 1e6))
# the long kernels
long = sorted(K, key=lambda k: k[0] - k[1])[:12]
for s, e, n, q, st in long:
    print("%8.0f -> %8.0f ms  %7.1f ms  q=%s s=%s %s" % ((s - W0) / 1e6, (e - W0) / 1e6, (e - s) / 1e6, q, st, n))
# union coverage per queue during the longest kernel
s0, e0, n0, q0, st0 = long[0]
print("during the longest kernel (%s, %.0f ms):" % (n0, (e0 - s0) / 1e6))
for q, v in by_q.items():
    iv = sorted((max(s, s0), min(e, e0)) for s, e, *_ in v if e > s0 and s < e0)
    cov = 0; cur = None
    for a, b in iv:
        if cur is None or a > cur[1]:
            if cur: cov += cur[1] - cur[0]
            cur = [a, b]
        else:
            cur[1] = max(cur[1], b)
    if cur: cov += cur[1] - cur[0]
    print("  queue/stream", q, "kernels", len(iv), "covered %.0f ms" % (cov / 1e6))
