#!/usr/bin/env python
"""Golden vectors for Alignment::CreateAlignmentStrings / AlignmentStringsToMD / PrintPairwise from the REFERENCE's own code
(oracle/_ref/aln_strings_ref).  Writes tests/golden/aln_strings_golden.json."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(20260929)
ALPH = np.frombuffer(b"ACGT", np.uint8)
cases = []
for c in range(160):
    tl = int(rng.integers(60, 400))
    text = ALPH[rng.integers(0, 4, tl)].copy()
    if rng.random() < 0.3:
        text[rng.integers(0, tl, 3)] = ord("N")
    if rng.random() < 0.3:
        text = np.frombuffer(text.tobytes().lower(), np.uint8).copy()            # soft-masked reference: MD upper-cases
    # walk: blocks with substitutions, insertions, deletions, and both at once (a common diagonal stretch)
    read = []; blocks = []
    q = int(rng.integers(0, 5)); t = int(rng.integers(0, 10))
    read.extend(ALPH[rng.integers(0, 4, q)].tolist())
    nb = int(rng.choice([0, 1, 2, 5, 9]))
    for b in range(nb):
        ln = int(rng.integers(1, 25))
        if t + ln + 30 > tl:
            break
        seg = np.frombuffer(bytes(text[t:t + ln]).upper(), np.uint8).copy()
        for i in range(ln):
            if rng.random() < 0.12:
                seg[i] = ALPH[rng.integers(0, 4)]
        blocks.append((q, t, ln)); read.extend(seg.tolist()); q += ln; t += ln
        qg = int(rng.choice([0, 0, 1, 3, 12])); tg = int(rng.choice([0, 0, 2, 7, 20]))
        read.extend(ALPH[rng.integers(0, 4, qg)].tolist()); q += qg; t += tg
    read.extend(ALPH[rng.integers(0, 4, int(rng.integers(0, 6)))].tolist())
    if not read: read = [ord("A")]
    cases.append(dict(name="read%d" % c, chrom="chr%d" % (c % 3 + 1), read=bytes(read).decode(), text=text.tobytes().decode(), blocks=blocks))
lines = []
for k in cases:
    lines.append(" ".join([k["name"], k["chrom"], k["read"], k["text"], str(len(k["blocks"]))] + [str(v) for b in k["blocks"] for v in b]))
out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "aln_strings_ref")], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
chunks = out.split("@@END\n")[:-1]
assert len(chunks) == len(cases)
json.dump({"source": "oracle/_ref/aln_strings_ref (reference Alignment.h:204-331, :564-589)", "cases": [dict(k, expected=ch) for k, ch in zip(cases, chunks)]},
          open(os.path.join(ROOT, "tests", "golden", "aln_strings_golden.json"), "w"))
print(len(cases), "cases")
