"""Generate tests/golden/stats_golden.json with the REFERENCE's Alignment::CalculateStatistics
(oracle/_ref/stats_ref = Alignment.h compiled in place): random block lists over mutated sequences,
including zero-length blocks, equal gaps on both sequences, long gaps (>20, >50, >10001 bases),
N / lower-case bases.  Stored: inputs + the reference's CIGAR, counters and the float value's bits."""
import json, os, random, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "stats_ref")


def case(rng, n_blocks, big=False):
    q = t = 0
    blocks = []
    for b in range(n_blocks):
        L = rng.choice([0, 1, 2, 5, 17, 40, 120])
        blocks.append([q, t, L])
        q += L; t += L
        kind = rng.random()
        g1 = rng.choice([0, 1, 2, 3, 7, 11, 21, 30, 50, 51, 80] + ([10002, 12000] if big else []))
        g2 = rng.choice([0, 0, 0, 1, 2, 5])
        if kind < 0.4:
            q += g1; t += g2
        elif kind < 0.8:
            t += g1; q += g2
        else:
            q += g2; t += g2
    qlen = q + rng.randint(0, 30); tlen = t + rng.randint(0, 30)
    alpha = "ACGT" * 6 + "Nacgt"
    read = "".join(rng.choice(alpha) for _ in range(qlen))
    # genome mostly equals the read along the blocks, with some mismatches
    g = [rng.choice("ACGT") for _ in range(tlen)]
    for (bq, bt, L) in blocks:
        for x in range(L):
            if rng.random() < 0.9:
                g[bt + x] = read[bq + x]
    return {"read": read, "genome": "".join(g), "blocks": [v for b in blocks for v in b]}


def main():
    rng = random.Random(23)
    cases = [case(rng, rng.choice([1, 2, 3, 6, 15, 40])) for _ in range(150)] + [case(rng, 6, big=True) for _ in range(4)]
    cases = [c for c in cases if c["read"] and c["genome"]]
    inp = "".join("%s %s %d %s\n" % (c["read"], c["genome"], len(c["blocks"]) // 3, " ".join(map(str, c["blocks"]))) for c in cases)
    out = subprocess.run([BIN], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    assert len(out) == len(cases)
    names = ["nm", "nmm", "nins", "ndel", "tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns", "value_bits",
             "preClip", "sufClip", "qStart", "qEnd", "tStart", "tEnd"]
    for c, line in zip(cases, out):
        f = line.split()
        c["cigar"] = "" if f[0] == "*" else f[0]
        c["out"] = dict(zip(names, [int(x) for x in f[1:]]))
    path = os.path.join(ROOT, "tests", "golden", "stats_golden.json")
    json.dump({"source": "oracle/_ref/stats_ref (reference Alignment.h compiled in place)", "cases": cases}, open(path, "w"))
    print("wrote", path, len(cases), os.path.getsize(path))


if __name__ == "__main__":
    main()
