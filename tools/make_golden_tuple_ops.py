"""Golden vectors for the k-mer primitives of StoreMinimizers (TupleOps.h:104-138, SeqUtils.h): runs oracle/_ref/tuple_ops_ref (the reference headers compiled in
place) on seeded sequences incl. N, lower case and other IUPAC letters -> tests/golden/tuple_ops_golden.json."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rng = np.random.default_rng(77)
    cases = []
    for i in range(60):
        k = int(rng.choice([7, 10, 15, 17, 19, 25, 31, 32]))
        n = int(rng.integers(k, 300))
        alpha = "ACGT" if i % 3 == 0 else ("ACGTNacgtn" if i % 3 == 1 else "ACGTNacgtnRYKMSWBDHV")
        p = np.array([0.24] * 4 + [0.04 / max(1, len(alpha) - 4)] * (len(alpha) - 4)); p /= p.sum()
        cases.append((k, "".join(rng.choice(list(alpha), n, p=p))))
    exe = os.path.join(ROOT, "oracle", "_ref", "tuple_ops_ref")
    text = "%d\n" % len(cases) + "".join("%d %s\n" % c for c in cases)
    out = subprocess.run([exe], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    gold = []
    for i, (k, s) in enumerate(cases):
        w = out[2 * i].split()
        assert int(w[0]) == len(s) - k + 1 and len(w) == 1 + 3 * int(w[0])
        gold.append(dict(k=k, seq=s, codes=[w[1:][j] for j in range(len(w) - 1)], rc=out[2 * i + 1]))
    json.dump(dict(source="TupleOps.h:95-138, SeqUtils.h:5-158 through oracle/ref_harness/tuple_ops_ref.cpp", cases=gold),
              open(os.path.join(ROOT, "tests", "golden", "tuple_ops_golden.json"), "w"))
    print(len(gold), "cases")


if __name__ == "__main__":
    main()
