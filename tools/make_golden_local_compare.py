"""Generate tests/golden/local_compare_golden.json with the REFERENCE's CompareLists<LocalTuple,SmallTuple>
(Global=false, optional diagonal band) compiled in place (oracle/_ref/comparelists_ref, mode L).
Lists mimic LocalIndex windows: ~40 tuples of 20-bit k-mers sorted by t, positions < 4096."""
import json, os, random, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "comparelists_ref")


def main():
    rng = random.Random(5)
    cases = []
    for _ in range(300):
        nq = rng.choice([0, 1, 2, 3, 10, 40, 90]); nt = rng.choice([0, 1, 2, 5, 40, 120])
        ks = rng.choice([4, 30, 1000, 1 << 20])
        q = sorted([[rng.randrange(ks), rng.randrange(256)] for _ in range(nq)], key=lambda x: x[0])
        t = sorted([[rng.randrange(ks), rng.randrange(256)] for _ in range(nt)], key=lambda x: x[0])
        band = rng.choice([(0, 0), (0, 0), (40, -40), (200, 1), (-1, -300), (5, -5)])
        cases.append({"q": q, "t": t, "maxFreq": rng.choice([1, 2, 6, 15]), "maxDiag": band[0], "minDiag": band[1]})
    inp = []
    for c in cases:
        inp.append("L %d %d %d %d %d" % (len(c["q"]), len(c["t"]), c["maxFreq"], c["maxDiag"], c["minDiag"]))
        inp += ["%d %d" % (a, b) for a, b in c["q"]] + ["%d %d" % (a, b) for a, b in c["t"]]
    out = subprocess.run([BIN], input=("\n".join(inp) + "\n").encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    assert len(out) == len(cases)
    for c, line in zip(cases, out):
        v = [int(x) for x in line.split()]
        assert len(v) == 1 + 4 * v[0]
        c["pairs"] = v[1:]
    path = os.path.join(ROOT, "tests", "golden", "local_compare_golden.json")
    json.dump({"source": "oracle/_ref/comparelists_ref mode L (reference CompareLists.h + TupleOps.h compiled in place)", "cases": cases}, open(path, "w"))
    print("wrote", path, len(cases), sum(len(c["pairs"]) // 4 for c in cases), os.path.getsize(path))


if __name__ == "__main__":
    main()
