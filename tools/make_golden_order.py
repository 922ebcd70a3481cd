#!/usr/bin/env python
"""Golden vectors for SegAlignmentGroup::SetFromSegAlignment + AlignmentsOrder::Update from the REFERENCE's own code
(oracle/_ref/order_ref, built from /root/reference in place by oracle/Makefile).  Writes tests/golden/order_golden.json."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260928)
cases = []
for c in range(240):
    ng = int(rng.choice([1, 1, 2, 2, 3, 5]))
    groups = []
    base_val = float(rng.choice([50.0, 1200.5, 30000.25]))
    for g in range(ng):
        nseg = int(rng.choice([1, 1, 2, 3]))
        segs = []
        for s in range(nseg):
            v = np.float32(base_val if rng.random() < 0.3 else rng.uniform(-50, 40000))          # equal values: the NumOfAnchors0 tie-break
            segs.append(dict(value_bits=int(np.float32(v).view(np.uint32)), N0=int(rng.choice([3, 8, 8, 25, 400])), N1=int(rng.integers(0, 900)),
                             qStart=int(rng.integers(0, 5000)), qEnd=int(rng.integers(5000, 30000)), tStart=int(rng.integers(1, 10 ** 6)),
                             tEnd=int(rng.integers(10 ** 6, 2 * 10 ** 6)), nm=int(rng.integers(0, 30000)), nmm=int(rng.integers(0, 900)),
                             ndel=int(rng.integers(0, 900)), nins=int(rng.integers(0, 900)), strand=int(rng.random() < 0.4),
                             supp=int(rng.random() < (0.2 if s == 0 else 0.8)), sec=int(rng.random() < 0.2), typeofaln=int(rng.choice([0, 0, 1, 3])),
                             flag=int(rng.choice([0, 0, 16, 2048]))))
        groups.append(segs)
    cases.append(groups)
KEYS = ("value_bits", "N0", "N1", "qStart", "qEnd", "tStart", "tEnd", "nm", "nmm", "ndel", "nins", "strand", "supp", "sec", "typeofaln", "flag")
lines = []
for groups in cases:
    t = [str(len(groups))]
    for segs in groups:
        t.append(str(len(segs)))
        for s in segs:
            t += [str(s[k]) for k in KEYS]
    lines.append(" ".join(t))
out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "order_ref")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
outs = out.strip().split("\n")
assert len(outs) == len(cases)
json.dump({"source": "oracle/_ref/order_ref (reference Alignment.h:944-983, :1021-1061)", "cases": [{"groups": g, "expected": o} for g, o in zip(cases, outs)]},
          open(os.path.join(ROOT, "tests", "golden", "order_golden.json"), "w"))
print(len(cases), "cases")
