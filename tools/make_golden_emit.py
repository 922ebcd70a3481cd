"""Generate tests/golden/emit_golden.json with the REFERENCE's own emitters (oracle/_ref/emit_ref = Alignment.h compiled in place):
random alignment records (flags, strands, clips, counters, float values incl. large / fractional ones, supplementary groups with
SA tags, hard clipping, unaligned records, missing qualities) -> PrintSAM / SimplePrintSAM / PrintPAF / PrintBed text."""
import json, os, random, struct, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "emit_ref")
FIELDS = ["name", "read", "qual", "chrom", "cigar", "readLen", "genomeLen", "flag", "strand", "mapqv", "supp", "typeofaln", "qStart", "qEnd", "tStart", "tEnd",
          "preClip", "sufClip", "nm", "nmm", "nins", "ndel", "tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns",
          "valueBits", "order", "N0", "N1", "runtime", "nBlocks", "firstBlockQPos", "lastBlockQEnd"]


def f2b(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def record(rng, name, read, qual, unaligned=False):
    L = len(read)
    qs = rng.randint(0, L // 3); qe = rng.randint(qs + 1, L)
    ts = rng.choice([0, 5, 123456, 4_000_000_000]); te = ts + rng.randint(1, 50000)
    strand = rng.randint(0, 1); supp = int(rng.random() < 0.4)
    flag = (16 if strand else 0) | (0x800 if supp else (256 if rng.random() < 0.2 else 0))
    nb = 0 if unaligned else rng.choice([1, 2, 7])
    val = rng.choice([0.0, 1.0, 123.0, 1234567.0, 98765.4321, 0.000123, 3.5e9, -42.25, 1e-7, 16777217.0])
    cig = "".join("%d%s" % (rng.randint(1, 900), rng.choice("=XID")) for _ in range(rng.randint(1, 12)))
    return dict(name=name, read=read, qual=qual, chrom=rng.choice(["chr1", "chr20", "contig_7|x"]), cigar=cig, readLen=L, genomeLen=rng.choice([1000, 64444167]),
                flag=flag, strand=strand, mapqv=rng.choice([0, 1, 30, 60, 255]), supp=supp, typeofaln=rng.choice([0, 1, 3]), qStart=qs, qEnd=qe, tStart=ts,
                tEnd=te, preClip=rng.choice([0, 0, 17, qs]), sufClip=rng.choice([0, 0, 250, L - qe]), nm=rng.randint(0, 30000), nmm=rng.randint(0, 900),
                nins=rng.randint(0, 900), ndel=rng.randint(0, 900), tdel=rng.randint(0, 5000), tins=rng.randint(0, 5000), nSmallDel=rng.randint(0, 9),
                nMedDel=rng.randint(0, 9), nLargeDel=rng.randint(0, 3), nSmallIns=rng.randint(0, 9), nMedIns=rng.randint(0, 9), nLargeIns=rng.randint(0, 3),
                valueBits=f2b(val), order=rng.randint(0, 4), N0=rng.randint(0, 500), N1=rng.choice([0, 0, 37, 1200]), runtime=rng.choice([0, 0, 12]),
                nBlocks=nb, firstBlockQPos=qs, lastBlockQEnd=qe)


def main():
    rng = random.Random(7)
    cases = []
    for k in range(260):
        L = rng.randint(20, 200)
        read = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = rng.choice(["*", "NULL", "".join(chr(rng.randint(35, 73)) for _ in range(L))])
        mode = "SsPpB"[k % 5]
        hard = rng.randint(0, 1)
        if mode == "s" and qual == "NULL" and not hard:
            qual = "*"                                  # SimplePrintSAM dereferences qual on this path
        ng = rng.choice([1, 1, 2, 4]) if mode == "S" else 1
        unal = k % 23 == 0
        if unal and qual == "*":
            qual = "NULL"                               # the unaligned branch copies readLen quality bytes unless qual is NULL
        group = [record(rng, "read/%d" % k, read, qual, unaligned=unal) for _ in range(ng)]
        cases.append(dict(mode=mode, hardClip=hard, passthrough=rng.choice(["-", "-", "BC:Z:ACGT"]), asIdx=rng.randrange(ng), group=group))
    lines = []
    for c in cases:
        parts = [c["mode"], str(c["hardClip"]), c["passthrough"], str(len(c["group"])), str(c["asIdx"])]
        for g in c["group"]:
            parts += [str(g[f]) if g[f] != "" else "-" for f in FIELDS]
        lines.append(" ".join(parts))
    out = subprocess.run([BIN], input=("\n".join(lines) + "\n").encode(), stdout=subprocess.PIPE, check=True).stdout.decode("latin-1").split("\n")
    assert len(out) == len(cases) + 1 and out[-1] == "", (len(out), len(cases))
    for c, line in zip(cases, out):
        c["text"] = line + "\n"
    path = os.path.join(ROOT, "tests", "golden", "emit_golden.json")
    json.dump({"source": "oracle/_ref/emit_ref (reference Alignment.h emitters compiled in place)", "fields": FIELDS, "cases": cases}, open(path, "w"))
    print("wrote", path, len(cases), os.path.getsize(path))


if __name__ == "__main__":
    main()
