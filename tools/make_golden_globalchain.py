"""Golden vectors for GlobalChain (GlobalChain.h / PrioritySearchTree.h): runs the reference's own templates (oracle/_ref/globalchain_ref, built by
oracle/Makefile from oracle/ref_harness/globalchain_ref.cpp + the headers under /root/reference) on seeded fragment sets and stores inputs + answers in
tests/golden/globalchain_golden.json.  The first case is the input of the reference's TestGlobalChain.cpp:27-38."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def problems():
    out = [[[0, 0, 10, 10], [20, 20, 30, 30], [40, 40, 50, 50], [60, 60, 70, 70], [80, 80, 90, 90], [100, 100, 110, 110], [120, 120, 130, 130],
            [140, 140, 150, 150], [81, 31, 91, 41]]]
    rng = np.random.default_rng(20260928)
    for case in range(199):
        n = int(rng.integers(0, 4)) if case < 8 else int(rng.integers(1, 60))
        grid = int(rng.choice([1, 5, 10, 25]))                            # coarse grids: shared corners, equal x with different y, fragments ending where others start
        fr = []
        x = y = 0
        for i in range(n):
            mode = rng.integers(0, 6)
            ln = int(rng.integers(1, 12)) * grid
            if mode <= 2:                                                  # roughly collinear
                x += int(rng.integers(0, 6)) * grid; y += int(rng.integers(0, 6)) * grid
                a, b = x, y
                x += ln; y += ln
            elif mode == 3:                                                # anywhere
                a, b = int(rng.integers(0, 80)) * grid, int(rng.integers(0, 80)) * grid
            elif mode == 4 and fr:                                         # starts exactly where an earlier fragment ends
                a, b = fr[int(rng.integers(0, len(fr)))][2:]
            else:                                                          # off-diagonal competitor
                a, b = x + int(rng.integers(0, 4)) * grid, max(0, y - int(rng.integers(1, 30)) * grid)
            fr.append([int(a), int(b), int(a + ln), int(b + ln)])
        order = rng.permutation(n)
        out.append([fr[i] for i in order])
    return out


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "globalchain_ref")
    P = problems()
    text = "%d\n" % len(P) + "".join("%d\n%s" % (len(p), "".join("%d %d %d %d\n" % tuple(f) for f in p)) for p in P)
    res = subprocess.run([exe], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    cases = []
    for i, p in enumerate(P):
        ch = [int(v) for v in res[2 * i].split()]
        sp = [int(v) for v in res[2 * i + 1].split()]
        assert ch[0] == len(ch) - 1 and len(sp) == 2 * len(p)
        cases.append(dict(fragments=p, chain=ch[1:], score=sp[0::2], prev=sp[1::2]))
    json.dump(dict(source="GlobalChain.h:85-189 + PrioritySearchTree.h through oracle/ref_harness/globalchain_ref.cpp; case 0 = TestGlobalChain.cpp:27-38", cases=cases),
              open(os.path.join(ROOT, "tests", "golden", "globalchain_golden.json"), "w"))
    print(len(cases), "cases;", "case 0 chain:", cases[0]["chain"], "scores", [cases[0]["score"][c] for c in cases[0]["chain"]])


if __name__ == "__main__":
    main()
