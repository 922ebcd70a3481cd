"""The high-accuracy presets at bench scale against the oracle: the first N reads of a tools/bench_presets.py batch through tests/oracle_pipeline.map_read_highacc (the
Python composition of the oracle's stage functions, a process per host CPU), every SegAlignment field by field as tests/test_highacc_path.py compares them.
usage (on the GPU box): python tools/scale_parity_highacc.py --preset ccs --sample 2000   ->  one JSON line (+ the first mismatches)"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

G = {}                                                                     # what the worker processes inherit by fork


def work(r):
    import oracle_lib as O
    import oracle_pipeline as OP
    rd = G["reads"][int(G["off"][r]):int(G["off"][r + 1])].tobytes()
    exp, unaligned, note = OP.map_read_highacc(rd, G["g"], G["ik"], G["ip"], G["oo"], chrom_pos=G["cp"], g_index=G["g_index"])
    out, na = G["out"], G["na"]
    if note is not None or out["read_status"][r] != 0:
        return (r, "note/status", str(note), int(out["read_status"][r]))
    by_h = {gr["h"]: gr["segs"] for gr in (exp or [])}
    for h in range(na):
        a0, a1 = int(out["job_aln_off"][r * na + h]), int(out["job_aln_off"][r * na + h + 1])
        if bool(out["job_reached"][r * na + h]) != (h in by_h):
            return (r, "reached", h, int(out["job_reached"][r * na + h]), h in by_h)
        e = by_h.get(h, [])
        if a1 - a0 != len(e):
            return (r, "n_segs", h, a1 - a0, len(e))
        for a, s in zip(range(a0, a1), e):
            if (out["strand"][a], out["supp"][a], out["secondary"][a], out["n0"][a], out["n1"][a], out["chrom"][a]) != (s["strand"], s["supp"], s["secondary"], s["n0"], s["n1"], s["chrom"]):
                return (r, "fields", h, a - a0)
            if np.float32(out["first_sdp_value"][a]).view(np.uint32) != np.float32(s["value"]).view(np.uint32):
                return (r, "value", h, a - a0)
            b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
            if not np.array_equal(b, s["blocks"]):
                return (r, "blocks", h, a - a0, len(b), len(s["blocks"]))
            ec, ev, eruns, _ = s["stats"]
            if out["counts"][a].tolist() != [ec[k] for k in O.STAT_NAMES]:
                return (r, "counts", h, a - a0)
            if np.float32(out["value"][a]).view(np.uint32) != np.float32(ev).view(np.uint32):
                return (r, "nv", h, a - a0)
            if not np.array_equal(out["runs"][int(out["run_off"][a]):int(out["run_off"][a + 1])], eruns):
                return (r, "cigar", h, a - a0)
    if unaligned and any(out["job_reached"][r * na:(r + 1) * na]):
        return (r, "unaligned")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", choices=["ccs", "contig"], default="ccs")
    ap.add_argument("--sample", type=int, default=1000)
    ap.add_argument("--err", type=float, default=0.0, help="the reads' error rate (default: the preset's)")
    ap.add_argument("--sv-frac", type=float, default=0.05, help="fraction of reads carrying one planted structural variant")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=0)
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import oracle_pipeline as OP
    from lra_amd.context import Context
    from lra_amd import seed, mapread, index as I, synth_genome as sg
    from bench import host_cpus
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    P = {"ccs": dict(reads=50000, read_len=15000, err=0.01, scale=64.4e6 / 3.09e9), "contig": dict(reads=1024, read_len=1000000, err=0.002, scale=1.0)}[args.preset]
    n_reads = args.reads or P["reads"]
    if args.err > 0:
        P["err"] = args.err
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=P["scale"], seed=3)
    ctx = Context(0)
    mapper = mapread.HighAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, args.preset, gli=True)
    sim = sg.simulate_reads_sv(genome, chrom_pos, n_reads, args.read_len or P["read_len"], (args.read_len or P["read_len"]) / 10, P["err"], (34, 33, 33), 1000, sv_frac=args.sv_frac)
    off_h = sim["off"].cpu().numpy(); total = int(off_h[-1])
    reads_h = np.frombuffer(sim["seq"][:total].cpu().numpy().tobytes(), np.uint8)
    del genome
    lseq = torch.cat([sim["seq"][:total], torch.zeros(64, dtype=torch.uint8, device=dev)])
    rbatch = seed.read_batch_from_device(ctx, lseq, sim["off"].contiguous())
    res = mapper.align(rbatch)
    out = mapper.fetch(res)
    O.lib()
    ik, ipos = I.global_index(ctx)
    g = ctx.to_host(ctx.lib.lra_ctx_genome_ptr(ctx.h), int(chrom_pos[-1]), np.uint8).tobytes() + b"\0" * 64
    oo = dict(OP.CONTIG if args.preset == "contig" else OP.CCS); oo.update(localK=10, localIndexWindow=2048)
    G.update(reads=reads_h, off=off_h, g=g, ik=ik, ip=ipos, oo=oo, cp=[int(x) for x in chrom_pos], g_index=mapper.fetch_local_index(), out=out, na=int(res.num_aln))
    S = min(args.sample, n_reads)
    nt = host_cpus()[0]
    t0 = time.time()
    with mp.get_context("fork").Pool(nt) as pool:
        bad = [x for x in pool.imap_unordered(work, range(S), chunksize=4) if x is not None]
    dt = time.time() - t0
    nal = int(out["job_aln_off"][S * int(res.num_aln)]) if S * int(res.num_aln) < len(out["job_aln_off"]) else int(res.n_alignments)
    print(json.dumps({"preset": args.preset, "reads_in_batch": n_reads, "sample": S, "alignments_in_sample": nal, "flagged_in_batch": int((out["read_status"] != 0).sum()),
                      "mismatching_reads": len(bad), "first_mismatches": [list(map(str, b)) for b in sorted(bad)[:10]], "oracle_seconds": round(dt, 1), "host_processes": nt,
                      "what": "lra_map_reads_highacc_batch against tests/oracle_pipeline.map_read_highacc: per chain the SegAlignmentGroup's existence, per SegAlignment strand, "
                              "Supplymentary, ISsecondary, NumOfAnchors0/1, chromosome, the chain's value, the refined blocks, the 18 counters, NV bits, CIGAR runs"}))


if __name__ == "__main__":
    main()
