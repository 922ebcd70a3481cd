"""a17: SAM / PAF / BED records of liblra_hip.so (host code) vs. the reference's own emitters (golden file from Alignment.h
compiled in place: tools/make_golden_emit.py).  Runs without a GPU."""
import json
import os
import struct

from lra_amd import emit

GOLD = os.path.join(os.path.dirname(__file__), "golden", "emit_golden.json")


def _rec(g):
    val = struct.unpack("<f", struct.pack("<I", g["valueBits"]))[0]
    qual = None if g["qual"] == "NULL" else g["qual"].encode()
    return emit.AlnRecord(g["name"].encode(), g["read"].encode(), qual, g["readLen"], g["chrom"].encode(), g["genomeLen"], g["cigar"].encode(), g["flag"],
                          g["strand"], g["mapqv"], g["supp"], g["typeofaln"], g["qStart"], g["qEnd"], g["tStart"], g["tEnd"], g["preClip"], g["sufClip"],
                          g["nm"], g["nmm"], g["nins"], g["ndel"], g["tdel"], g["tins"], g["nSmallDel"], g["nMedDel"], g["nLargeDel"], g["nSmallIns"],
                          g["nMedIns"], g["nLargeIns"], val, g["order"], g["N0"], g["N1"], g["runtime"], g["nBlocks"], g["firstBlockQPos"], g["lastBlockQEnd"])


def test_emitters_match_reference_text():
    gold = json.load(open(GOLD))
    seen = set()
    for c in gold["cases"]:
        recs = [_rec(g) for g in c["group"]]
        pt = None if c["passthrough"] == "-" else c["passthrough"].encode()
        m = c["mode"]
        if m == "S": got = emit.format_sam(recs, c["asIdx"], c["hardClip"], pt)
        elif m == "s": got = emit.format_sam_simple(recs[0], c["hardClip"], pt)
        elif m == "P": got = emit.format_paf(recs[0], True)
        elif m == "p": got = emit.format_paf(recs[0], False)
        else: got = emit.format_bed(recs[0])
        assert got.decode() == c["text"], (m, c["group"][0]["name"])
        seen.add(m)
    assert seen == set("SsPpB")
