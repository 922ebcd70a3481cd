#!/usr/bin/env python
"""per-kernel-name total duration in a kernel trace CSV over the last W seconds, plus per-stream busy time and the idle time of the device"""
import csv, glob, sys, collections, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
W = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:34], r.get("Stream_Id", "?")) for r in rows]
t_end = max(k[1] for k in K); W0 = t_end - int(W * 1e9)
K = sorted(k for k in K if k[0] >= W0)
tot = collections.Counter(); cnt = collections.Counter(); bys = collections.Counter()
for s, e, n, st in K: tot[n] += e - s; cnt[n] += 1; bys[st] += e - s
# union coverage
cov = 0; cur = None
for s, e, *_ in K:
    if cur is None or s > cur[1]:
        if cur: cov += cur[1] - cur[0]
        cur = [s, e]
    else: cur[1] = max(cur[1], e)
if cur: cov += cur[1] - cur[0]
print("window %.0f ms: device busy (union) %.0f ms; per stream busy: %s" % (W * 1e3, cov / 1e6, {k: round(v / 1e6) for k, v in bys.items()}))
for n, v in tot.most_common(22): print("%-36s %5d %9.1f ms" % (n, cnt[n], v / 1e6))
