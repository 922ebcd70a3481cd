"""The other BASELINE.json configurations on one MI355X (bench.py measures the headline, configs[2]):
  --preset ccs     configs[1]: a chr20-sized reference (64 Mb) + 15 kb CCS reads (1 % error), -CCS       -> lra_map_reads_highacc_batch
  --preset clr     configs[3]: the GRCh38-like reference + 20 kb CLR reads (15 % error), -CLR            -> lra_map_reads_lowacc_batch
  --preset contig  configs[4]: the GRCh38-like reference + long assembly contigs (0.2 % error), -CONTIG  -> lra_map_reads_highacc_batch
One JSON line per run: whole-step throughput (device side + record text), inputs resident in HBM, no CPU baseline (bench.py carries that for the headline).
Results are kept under profiles/ (r02_<preset>_bench.json)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", choices=["ccs", "clr", "contig"], required=True)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--genome-scale", type=float, default=0.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gli", type=int, default=1, help="1: the local index as `lra index` writes it (k = 10, w = 5, windows of 2048 bases: what glIndex.Read hands `lra align`); "
                                                       "0: `lra align` without a .gli file (opts.localK, windows of 256 bases)")
    ap.add_argument("--two-stage", type=int, default=0, help="-CLR only: two-stage batches (lra_map_reads_lowacc_front / _back), as bench.py runs the headline step")
    ap.add_argument("--err", type=float, default=0.0, help="the reads' error rate (default: the preset's)")
    ap.add_argument("--sv-frac", type=float, default=0.05, help="fraction of reads carrying one planted structural variant")
    ap.add_argument("--oracle-sample", type=int, default=0, help="-CLR only: the first N reads of the batch through the oracle's MapRead_lowacc on the host's cores as well "
                                                                  "(bench.py's cpu_baseline with the -CLR options): its rate, and whether its alignments equal the last step's")
    args = ap.parse_args()
    import torch
    from lra_amd.context import Context
    from lra_amd import seed, mapread, synth_genome as sg
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    P = {"ccs": dict(reads=50000, read_len=15000, err=0.01, scale=64.4e6 / 3.09e9, mix=(34, 33, 33)),
         "clr": dict(reads=28672, read_len=20000, err=0.15, scale=1.0, mix=(20, 30, 50)),
         "contig": dict(reads=1024, read_len=1000000, err=0.002, scale=1.0, mix=(34, 33, 33))}[args.preset]
    n_reads = args.reads or P["reads"]; read_len = args.read_len or P["read_len"]; scale = args.genome_scale or P["scale"]
    if args.err > 0:
        P["err"] = args.err
    t0 = time.time()
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=scale, seed=3)
    torch.cuda.synchronize()
    ctx = Context(0)
    if args.preset == "clr":
        mopts = mapread.with_gli(mapread.clr_options()) if args.gli else mapread.clr_options()
        mapper = mapread.LowAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, mopts, index_params=(15, 10, 250, 12, 1), staged=False)
    else:
        mapper = mapread.HighAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, args.preset, gli=bool(args.gli) or None)
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    sim = sg.simulate_reads_sv(genome, chrom_pos, n_reads, read_len, read_len / 10, P["err"], P["mix"], 1000, sv_frac=args.sv_frac)
    off_h = sim["off"].cpu().numpy()
    total = int(off_h[-1])
    reads_h = sim["seq"][:total].cpu().numpy().tobytes()
    del genome
    lseq = torch.cat([sim["seq"][:total], torch.zeros(64, dtype=torch.uint8, device=dev)])
    rbatch = seed.read_batch_from_device(ctx, lseq, sim["off"].contiguous())
    names = [b"read%d" % i for i in range(n_reads)]
    reads_b = [reads_h[int(off_h[i]):int(off_h[i + 1])] for i in range(n_reads)]
    rargs = mapper.record_args(names, reads_b)
    text = [0]

    import threading
    tail = [None]

    def fmt(snap):
        text[0] = mapper.records_host(snap, rargs, as_list=False)

    def join_tail():
        if tail[0] is not None:
            tail[0].join(); tail[0] = None

    def step():
        # the record text of a batch (host threads only) beside the next batch's device side, as in bench.py
        res = mapper.align(rbatch)
        snap = mapper.snapshot(res)
        join_tail()
        tail[0] = threading.Thread(target=fmt, args=(snap,)); tail[0].start()
        return res
    two_stage = bool(args.two_stage) and args.preset == "clr"
    last = [None, mapper]

    def steps_two_stage(n):
        # the front halves on a thread of their own (the mapping context on a low-priority stream), the back halves + snapshot here, the text beside both
        ft = threading.Thread(target=lambda: [mapper.front(rbatch) for _ in range(n)])
        ft.start()
        for _ in range(n):
            r_, bc = mapper.back()
            mb = mapper.on(bc)
            try:
                snap = mb.snapshot(r_)
            finally:
                mapper.release()
            join_tail()
            tail[0] = threading.Thread(target=fmt, args=(snap,)); tail[0].start()
            last[0], last[1] = r_, mb
        ft.join()
    if two_stage:
        lo, _hi = torch.cuda.Stream.priority_range()
        fstream = torch.cuda.Stream(device=0, priority=lo)
        ctx.bind_stream(fstream)
        steps_two_stage(args.warmup)
    else:
        for _ in range(args.warmup):
            res = step()
    join_tail()
    ctx.timing(True); ctx.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if two_stage:
        steps_two_stage(args.steps)
        res = last[0]
    else:
        for _ in range(args.steps):
            res = step()
    join_tail()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = last[1].fetch(res)
    flagged = int((out["read_status"] != 0).sum())
    import hashlib
    hh = hashlib.sha256()
    for k in ("job_aln_off", "read_status", "strand", "chrom", "block_off", "blocks", "counts"):
        if k in out:
            hh.update(np.ascontiguousarray(out[k]).tobytes())
    aligned = int(sum(1 for r in range(n_reads) if out["job_aln_off"][r * int(res.num_aln) + int(res.num_aln)] > out["job_aln_off"][r * int(res.num_aln)]))
    free_b, tot_b = torch.cuda.mem_get_info()
    cpu = None
    if args.oracle_sample and args.preset == "clr":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench
        import oracle_pipeline as OP
        oo = dict(OP.CLR, localIndexWindow=mopts.localIndexWindow)
        cpu = bench.cpu_baseline(mapper, np.frombuffer(reads_h, np.uint8), off_h, None, res, opts=oo, sample=args.oracle_sample, res_mapper=last[1] if two_stage else None)
    print(json.dumps({
        "metric": "aligned Gbp/s (%s)" % {"ccs": "15 kb CCS vs a chr20-sized reference, -CCS, MapRead_highacc end to end incl. SAM text",
                                          "clr": "20 kb CLR vs the GRCh38-like reference, -CLR, MapRead_lowacc end to end incl. SAM text",
                                          "contig": "assembly contigs vs the GRCh38-like reference, -CONTIG, MapRead_highacc end to end incl. SAM text"}[args.preset],
        "value": total * args.steps / dt / 1e9, "unit": "Gbp/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "reads_per_s": n_reads * args.steps / dt, "higher_is_better": True, "data": "synthetic", "dtype": "int32",
        "config": {"workload": "BASELINE configs[%d]" % {"ccs": 1, "clr": 3, "contig": 4}[args.preset], "reads": n_reads, "mean_read_len": read_len, "error": P["err"],
                   "reference_bp": int(chrom_pos[-1]), "index_entries": int(mapper.index_stats.get("n_index", 0)),
                   "local_index": "k 10, w 5, windows of 2048 bases (the .gli file `lra index` writes)" if args.gli else "the options' localK, windows of 256 bases (`lra align` without a .gli file)"},
        "reads_with_an_alignment": aligned, "reads_flagged": flagged, "n_alignments": int(res.n_alignments), "sam_text_mb": text[0] / 1e6,
        "cpu_baseline": cpu, "result_sha256": hh.hexdigest(), "two_stage": two_stage, "setup_s": round(setup_s, 1), "hbm_used_gb": round((tot_b - free_b) / 1e9, 1)}))


if __name__ == "__main__":
    main()
