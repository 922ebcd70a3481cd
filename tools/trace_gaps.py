#!/usr/bin/env python
"""Diagnostic over a rocprofv3 --kernel-trace CSV: the largest idle gaps of the device (no kernel of any queue running) in the last argv[2] seconds, with the
kernels on either side -- where the host is doing sizing round trips or its own work between launches."""
import csv, glob, sys, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:44]) for r in csv.DictReader(open(f))]
K.sort()
t_end = max(k[1] for k in K); W0 = t_end - int(float(sys.argv[2]) * 1e9)
K = [k for k in K if k[0] >= W0]
gaps = []; cur_end = K[0][1]; last = K[0][2]
busy = 0; seg_start = K[0][0]
for s, e, n in K[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, (cur_end - W0) / 1e6, last, n)); busy += cur_end - seg_start; seg_start = s
    if e > cur_end: cur_end, last = e, n
busy += cur_end - seg_start
print("window %.0f ms, device busy %.0f ms, idle %.0f ms in %d gaps (%d kernels)" % ((t_end - K[0][0]) / 1e6, busy / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps), len(K)))
for g in sorted(gaps, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%7.2f ms idle at %8.1f ms  after %-44s before %s" % (g[0] / 1e6, g[1], g[2], g[3]))
import collections
by = collections.Counter()
for g in gaps: by[(g[2], g[3])] += g[0]
print("--- by (after, before), summed")
for (a, b), v in by.most_common(15): print("%7.2f ms  after %-44s before %s" % (v / 1e6, a, b))
