"""Generate tests/golden/aog_golden.json by running the REFERENCE's AffineOneGapAlign
(oracle/_ref/aog_ref, compiled from /root/reference) in this container.

Inputs: (i) the 22 query/target pairs the reference's own TestAffineOneGapAlign.cpp:19-68
holds (string literals = test data), with its parameters (4,-4,-3,15) and with the
preset parameter sets the live path uses; (ii) seeded random pairs covering equal
lengths, long one-sided gaps (the "alignTop" branch), length 0/1 and N bases.
Only inputs + the reference's outputs are stored (no reference source).
"""
import json, os, re, random, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TEST = "/root/reference/TestAffineOneGapAlign.cpp"
BIN = os.path.join(ROOT, "oracle", "_ref", "aog_ref")


def reference_test_pairs():
    src = open(REF_TEST).read()
    src = src.split("/*\n\tstring target")[0]
    pairs = []
    for mobj in re.finditer(r'Test\(\s*"([A-Za-z]*)"\s*,\s*"([A-Za-z]*)"\s*\)', src):
        pairs.append((mobj.group(1), mobj.group(2)))
    return pairs


def mutate(rng, s, sub, ins, dele):
    out = []
    for c in s:
        r = rng.random()
        if r < sub:
            out.append(rng.choice("ACGT"))
        elif r < sub + ins:
            out.append(c); out.append(rng.choice("ACGT"))
        elif r < sub + ins + dele:
            continue
        else:
            out.append(c)
    return "".join(out)


def random_pairs(seed=7, n=260):
    rng = random.Random(seed)
    cases = []
    for x in range(n):
        L = rng.choice([0, 1, 2, 3, 5, 8, 9, 12, 20, 31, 47, 62, 90, 150, 300, 640])
        s = "".join(rng.choice("ACGT") for _ in range(L))
        mode = x % 5
        if mode == 0:
            q, t = s, mutate(rng, s, 0.05, 0.03, 0.03)
        elif mode == 1:   # long insertion in q
            g = "".join(rng.choice("ACGT") for _ in range(rng.choice([40, 90, 200, 500])))
            p = rng.randint(0, L)
            q, t = s[:p] + g + s[p:], mutate(rng, s, 0.03, 0.02, 0.02)
        elif mode == 2:   # long insertion in t
            g = "".join(rng.choice("ACGT") for _ in range(rng.choice([40, 90, 200, 500])))
            p = rng.randint(0, L)
            q, t = mutate(rng, s, 0.03, 0.02, 0.02), s[:p] + g + s[p:]
        elif mode == 3:   # unrelated
            q = s
            t = "".join(rng.choice("ACGT") for _ in range(rng.choice([0, 1, 4, 17, 33, 80])))
        else:             # with N / lowercase
            q = mutate(rng, s, 0.1, 0.05, 0.05).replace("A", "N", 1)
            t = s.lower() if x % 2 else s
        k = rng.choice([1, 2, 3, 5, 7, 15, 30])
        par = rng.choice([(4, -3, -4), (4, -1, -2), (4, -4, -3), (1, -1, -1)])
        cases.append((q, t, par[0], par[1], par[2], k))
    return cases


def main():
    cases = []
    for q, t in reference_test_pairs():
        for (m, mm, indel, k) in [(4, -4, -3, 15), (4, -3, -4, 15), (4, -1, -2, 30), (4, -3, -4, 7)]:
            cases.append((q, t, m, mm, indel, k))
    n_ref = len(cases)
    cases += random_pairs()
    inp = "".join("%s %s %d %d %d %d\n" % (q or "-", t or "-", m, mm, indel, k) for q, t, m, mm, indel, k in cases)
    out = subprocess.run([BIN], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    assert len(out) == len(cases)
    recs = []
    for c, line in zip(cases, out):
        v = [int(x) for x in line.split()]
        assert len(v) == 2 + 3 * v[1]
        recs.append({"q": c[0], "t": c[1], "m": c[2], "mm": c[3], "indel": c[4], "k": c[5], "score": v[0], "blocks": v[2:]})
    path = os.path.join(ROOT, "tests", "golden", "aog_golden.json")
    json.dump({"source": "oracle/_ref/aog_ref (reference AffineOneGapAlign.h compiled in place)",
               "n_from_reference_test_inputs": n_ref, "cases": recs}, open(path, "w"))
    print("wrote", path, len(recs), "cases (", n_ref, "from TestAffineOneGapAlign.cpp inputs )")


if __name__ == "__main__":
    main()
