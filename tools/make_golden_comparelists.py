"""Generate tests/golden/comparelists_golden.json with the REFERENCE's CompareLists and
std::sort(readmm) (oracle/_ref/comparelists_ref = CompareLists.h compiled in place).
Cases stress equal-key runs with mixed strand bits, maxFreq cut-offs, one-element tails and
disjoint ranges.  Stored: inputs + the reference's emitted (query index, target index) pairs
and its sorted query permutation."""
import json, os, random, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "comparelists_ref")
M = (1 << 63) - 1


def gen_case(rng, nq, nt, keyspace, maxfreq, strand_p):
    def tup(pos):
        k = rng.randrange(keyspace)
        if rng.random() < strand_p:
            k |= 1 << 63
        return [k, pos]
    q = [tup(i * 3 + 1) for i in range(nq)]
    t = [tup(1000000 + i * 7) for i in range(nt)]
    rng.shuffle(t)
    t.sort(key=lambda x: x[0] & M)          # index order: sorted by masked key, ties arbitrary
    return {"q": q, "t": t, "maxFreq": maxfreq}


def main():
    rng = random.Random(11)
    cases = []
    for nq, nt, ks, mf, sp in [(0, 5, 10, 5, 0.5), (5, 0, 10, 5, 0.5), (1, 1, 2, 5, 0.5), (2, 2, 2, 5, 0.5),
                                (3, 10, 4, 5, 0.5), (10, 3, 4, 5, 0.5)]:
        cases.append(gen_case(rng, nq, nt, ks, mf, sp))
    for _ in range(120):
        nq = rng.choice([2, 3, 5, 8, 17, 40, 100, 300])
        nt = rng.choice([1, 2, 5, 20, 100, 400, 1500])
        ks = rng.choice([3, 8, 30, 200, 5000])
        cases.append(gen_case(rng, nq, nt, ks, rng.choice([1, 2, 3, 50]), rng.choice([0.0, 0.3, 0.5])))
    for ks in (2000, 50000):
        cases.append(gen_case(rng, 3000, 20000, ks, 30, 0.5))
    inp = []
    for c in cases:
        inp.append("%d %d %d 1" % (len(c["q"]), len(c["t"]), c["maxFreq"]))
        inp += ["%d %d" % (a, b) for a, b in c["q"]]
        inp += ["%d %d" % (a, b) for a, b in c["t"]]
    out = subprocess.run([BIN], input=("\n".join(inp) + "\n").encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    assert len(out) == 2 * len(cases)
    for i, c in enumerate(cases):
        v = [int(x) for x in out[2 * i].split()]
        assert len(v) == 1 + 2 * v[0]
        c["pairs"] = v[1:]
        c["sorted_perm"] = [int(x) for x in out[2 * i + 1].split()]
    path = os.path.join(ROOT, "tests", "golden", "comparelists_golden.json")
    json.dump({"source": "oracle/_ref/comparelists_ref (reference CompareLists.h + std::sort compiled in place)", "cases": cases}, open(path, "w"))
    print("wrote", path, len(cases), "cases", sum(len(c["pairs"]) // 2 for c in cases), "pairs", os.path.getsize(path))


if __name__ == "__main__":
    main()
