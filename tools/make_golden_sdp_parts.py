"""Generate tests/golden/sdp_parts_golden.json with the pieces of the REFERENCE's sparse-DP engine that compile here
(oracle/_ref/sdp_parts_ref = Sorting.h, DivideSubBy*.h, SubRountine.h, SubProblem.h ... compiled in place):
  * "divide": point sets -> FNV-1a hash + length of the canonical text listing the H1/H2 sort permutations, the row/column
    tables with their SS_A/SS_B lists and every sub-problem's Di/Ei/Db/Eb of the four decompositions (small cases also
    keep the text itself);
  * "pwl":    InitPWL parameter sets -> SLOPE/INTER tables and PWL_w / w over a grid of gap lengths (float bits);
  * "maxim":  one ascending sub-problem driven by a script of value updates and queries -> FindValueInBlock answers and the
    final Block list.
Inputs are stored with the outputs, so the file is self-contained."""
import json, os, random, struct, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "sdp_parts_ref")


def fnv1a(s: bytes):
    h = 1469598103934665603
    for ch in s:
        h ^= ch
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def frag_points(rng, n_clusters, per_cluster, span, ties):
    """points of SDP#A (SparseDP.h:2152-2169) for random clusters of fragments"""
    pts = []
    for c in range(n_clusters):
        strand = rng.random() < 0.4
        q = rng.randint(0, span); t = rng.randint(0, span)
        m = rng.randint(1, per_cluster)
        for i in range(m):
            L = rng.choice([1, 5, 17, 17, 17, 30, 60]) if not ties else rng.choice([1, 2, 3])
            edge = i == 0 or i == m - 1
            def pair(kind):
                if kind == 0: pts.extend([(q, t, 1, 1), (q + L, t + L, 0, 1)])
                else: pts.extend([(q, t + L, 1, 0), (q + L, t, 0, 0)])
            if not strand:
                pair(0)
                if edge: pair(1)
            else:
                pair(1)
                if edge: pair(0)
            step = rng.randint(0, 3) if ties else rng.randint(0, 120)
            q += L + step
            if strand: t = max(0, t - L - rng.randint(0, 3 if ties else 120))
            else: t += L + (rng.randint(0, 3) if ties else rng.randint(0, 120))
    return pts


def f2b(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def main():
    rng = random.Random(41)
    divide = []
    for k in range(140):
        if k < 30: pts = frag_points(rng, rng.randint(1, 3), 3, 12, True)          # heavy ties, tiny
        elif k < 60: pts = frag_points(rng, rng.randint(1, 4), 6, 40, True)
        elif k < 110: pts = frag_points(rng, rng.randint(1, 6), 12, 3000, False)
        else: pts = frag_points(rng, rng.randint(4, 12), 40, 30000, False)         # realistic sizes (hash only)
        if k % 17 == 0: pts = [(rng.randint(0, 6), rng.randint(0, 6), rng.randint(0, 1), rng.randint(0, 1)) for _ in range(rng.randint(1, 40))]
        divide.append({"pts": [v for p in pts for v in p]})
    pwl_params = [(7.0, 10.0, 1.5, 1500, 3000), (4.0, 20.0, 1.5, 3000, 5000), (4.0, 15.0, 1.5, 2000, 3000), (2.0, 10.0, 2.0, 1500, 3000)]
    xs = sorted(set(list(range(1, 130)) + [rng.randint(1, 120000) for _ in range(300)] +
                    [s + d for s in (5, 10, 20, 40, 80, 100, 200, 300, 500, 1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000, 15000, 20000,
                                     30000, 40000, 50000, 100000) for d in (-1, 0, 1)] + [10**6, 10**7]))
    maxim = []
    for k in range(120):
        nD = rng.randint(1, 40); nE = rng.randint(1, 40)
        rngspan = rng.choice([20, 200, 5000, 60000])
        Di = sorted(rng.sample(range(-rngspan, rngspan), min(nD, 2 * rngspan)))
        Ei = sorted(rng.sample(range(-rngspan, rngspan), min(nE, 2 * rngspan)))
        ops = []
        cur = 0
        for _ in range(rng.randint(1, 120)):
            if rng.random() < 0.55:
                ops.append((0, rng.randrange(len(Di)), f2b(float(rng.choice([20, 340, 340, 1000, 5000, 20000]) * rng.randint(1, 9)))))
            else:
                cur = min(len(Ei) - 1, cur + rng.choice([0, 0, 1, 1, 2, 5]))   # queries arrive in non-decreasing Ei order in ProcessPoint
                ops.append((1, cur, 0))
        maxim.append({"params": pwl_params[k % 4], "Di": Di, "Ei": Ei, "ops": [v for o in ops for v in o]})

    inp = []
    for c in divide:
        p = c["pts"]
        inp.append("D %d %s" % (len(p) // 4, " ".join(map(str, p))))
    for pr in pwl_params:
        inp.append("P %r %r %r %d %d %d %s" % (pr[0], pr[1], pr[2], pr[3], pr[4], len(xs), " ".join(map(str, xs))))
    for c in maxim:
        pr = c["params"]
        inp.append("M %r %r %r %d %d %d %s %d %s %d %s" % (pr[0], pr[1], pr[2], pr[3], pr[4], len(c["Di"]), " ".join(map(str, c["Di"])),
                                                           len(c["Ei"]), " ".join(map(str, c["Ei"])), len(c["ops"]) // 3,
                                                           " ".join(map(str, c["ops"]))))
    out = subprocess.run([BIN], input=("\n".join(inp) + "\n").encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    pos = 0
    for c in divide:
        text = []
        while out[pos] != "END":
            text.append(out[pos]); pos += 1
        pos += 1
        s = "".join(l + "\n" for l in text)
        c["len"] = len(s); c["fnv1a"] = "%016x" % fnv1a(s.encode())
        if len(s) < 1500: c["text"] = s
    pwl = []
    for pr in pwl_params:
        rows = [out[pos + i].split() for i in range(len(xs))]; pos += len(xs)
        tab = [out[pos + i].split() for i in range(25)]; pos += 25
        pwl.append({"params": pr, "pwl_bits": [int(r[0]) for r in rows], "w_bits": [int(r[1]) for r in rows],
                    "slope_bits": [int(r[0]) for r in tab], "inter_bits": [int(r[1]) for r in tab]})
    for c in maxim:
        c["out"] = [int(x) for x in out[pos].split()]; pos += 1
        c["block"] = [int(x) for x in out[pos].split()]; pos += 1
    path = os.path.join(ROOT, "tests", "golden", "sdp_parts_golden.json")
    json.dump({"source": "oracle/_ref/sdp_parts_ref (reference sparse-DP component headers compiled in place)", "divide": divide, "xs": xs,
               "pwl": pwl, "maxim": maxim}, open(path, "w"))
    print("wrote", path, len(divide), len(pwl), len(maxim), os.path.getsize(path))


if __name__ == "__main__":
    main()
