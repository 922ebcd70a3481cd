"""Which reads of a bench-scale batch the device and the oracle disagree on (the low-accuracy presets): per-read hashes of the device's alignments against the oracle's
checksum over halving ranges of reads.  usage (on the GPU box): python tools/locate_mismatch.py --preset clr [--reads N] [--two-stage 0]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", choices=["ont", "clr"], default="clr")
    ap.add_argument("--reads", type=int, default=28672)
    ap.add_argument("--max-report", type=int, default=12)
    ap.add_argument("--sv-frac", type=float, default=0.05)
    ap.add_argument("--n-frac", type=float, default=0.0, help="this share of the reads gets a run of N (20-400 bases) and a run of lower-case bases")
    ap.add_argument("--alnthres", type=float, default=0.0, help="opts.alnthres (-a): the share of the best chain's value a further primary chain needs; low values make second chains common")
    ap.add_argument("--refine-breakpoints", action="store_true", help="--refineBreakpoints (a15, off by default in lra)")
    ap.add_argument("--only-flagged", action="store_true", help="no oracle: the status words of the reads the device flagged")
    args = ap.parse_args()
    import torch
    import oracle_lib as O
    import oracle_pipeline as OP
    from lra_amd.context import Context
    from lra_amd import seed, mapread, index as I, synth_genome as sg
    from bench import host_cpus
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=1.0, seed=3)
    ctx = Context(0)
    if args.preset == "clr":
        mopts = mapread.with_gli(mapread.clr_options()); ip = (15, 10, 250, 12, 1); rl, err, mix = 20000, 0.15, (20, 30, 50); oo = dict(OP.CLR)
    else:
        mopts = mapread.with_gli(mapread.LowAccOptions()); ip = (17, 10, 150, 12, 1); rl, err, mix = 30000, 0.10, (30, 35, 35); oo = dict(OP.ONT)
    oo["localIndexWindow"] = mopts.localIndexWindow
    if args.alnthres > 0:
        import dataclasses
        mopts = dataclasses.replace(mopts, alnthres=args.alnthres); oo["alnthres"] = args.alnthres
    if args.refine_breakpoints:
        import dataclasses
        mopts = dataclasses.replace(mopts, refineBreakpoint=True); oo["refineBreakpoint"] = True
    mapper = mapread.LowAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, mopts, index_params=ip, staged=False)
    sim = sg.simulate_reads_sv(genome, chrom_pos, args.reads, rl, rl / 10, err, mix, 1000, sv_frac=args.sv_frac)
    off_h = sim["off"].cpu().numpy(); total = int(off_h[-1])
    if args.n_frac > 0:                                                       # N runs (MinCount.h skips windows with an N; IndelRefine / statistics see the bases as they are) and lower case
        rng = np.random.default_rng(7)
        seq = sim["seq"]
        for r in np.nonzero(rng.random(args.reads) < args.n_frac)[0]:
            a, b = int(off_h[r]), int(off_h[r + 1])
            if b - a < 2000:
                continue
            x = a + int(rng.integers(0, b - a - 500)); ln = int(rng.integers(20, 400))
            seq[x:x + ln] = ord("N")
            y = a + int(rng.integers(0, b - a - 500)); l2 = int(rng.integers(20, 400))
            seq[y:y + l2] = seq[y:y + l2] | 32
    reads_h = np.frombuffer(sim["seq"][:total].cpu().numpy().tobytes(), np.uint8)
    del genome
    lseq = torch.cat([sim["seq"][:total], torch.zeros(64, dtype=torch.uint8, device=dev)])
    rbatch = seed.read_batch_from_device(ctx, lseq, sim["off"].contiguous())
    res = mapper.align(rbatch)
    out = mapper.fetch(res)
    na = int(res.num_aln)
    fl = np.nonzero(out["read_status"])[0]
    print(json.dumps({"flagged_reads": [{"read": int(r), "status": "0x%x" % int(out["read_status"][r]), "length": int(off_h[r + 1] - off_h[r])} for r in fl[:20]]}))
    if args.only_flagged:
        return
    P = np.uint64(1099511628211)
    n = args.reads
    hs = np.zeros(n, np.uint64); nal = np.zeros(n, np.int64)
    with np.errstate(over="ignore"):
        reached = out.get("job_reached")
        for r in range(n):
            a0 = int(out["job_aln_off"][r * na]); a1 = a0                  # (the loop over the primary chains as Map_lowacc.h:259-267 runs it: bench.cpu_baseline)
            for p_ in range(na):
                j = r * na + p_
                if reached is not None and len(reached) and not reached[j]:
                    break
                a1 = int(out["job_aln_off"][j + 1])
            if out["read_status"][r] or a1 == a0 or int(out["job_aln_off"][r * na + 1]) == a0:                       # (p = 0 without a SegAlignment: unaligned, :578-581)
                continue
            parts = []
            for a in range(a0, a1):
                b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
                parts.append(b.reshape(-1).astype(np.uint32).astype(np.uint64)); parts.append(out["counts"][a].astype(np.int64).astype(np.uint64))
            x = np.concatenate(parts)
            pw = np.ones(len(x), np.uint64)
            if len(x) > 1:
                pw[1:] = P; pw = np.multiply.accumulate(pw)
            hs[r] = np.sum(x * pw[::-1], dtype=np.uint64); nal[r] = a1 - a0
    O.lib()
    key, pos = I.global_index(ctx)
    g = ctx.to_host(ctx.lib.lra_ctx_genome_ptr(ctx.h), mapper.G, np.uint8).tobytes() + b"\0" * 64
    g_index = mapper.fetch_local_index()
    nt = host_cpus()[0]

    def oracle(first, cnt):
        r = OP.map_reads_lowacc_mt(reads_h, off_h, first, cnt, g, key, pos, g_index, oo, mapper.chrom_pos, n_threads=nt)
        return r["checksum"], r["n_alignments"]

    def gpu(first, cnt):
        with np.errstate(over="ignore"):
            w = np.arange(first + 1, first + cnt + 1, dtype=np.uint64)
            return int(np.sum(hs[first:first + cnt] * w, dtype=np.uint64)), int(nal[first:first + cnt].sum())
    bad = []
    todo = [(0, n)]
    while todo and len(bad) < args.max_report:
        first, cnt = todo.pop()
        oc, on = oracle(first, cnt); gc, gn = gpu(first, cnt)
        if oc == gc and on == gn:
            continue
        if cnt == 1:
            bad.append((first, gn, on)); continue
        h = cnt // 2
        todo.append((first + h, cnt - h)); todo.append((first, h))
    print(json.dumps({"preset": args.preset, "reads": n, "flagged": int((out["read_status"] != 0).sum()), "mismatching_reads": [{"read": r, "gpu_alignments": a, "oracle_alignments": b,
                      "length": int(off_h[r + 1] - off_h[r]), "status": int(out["read_status"][r])} for r, a, b in bad]}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for r, _, _ in bad[:2]:                                                    # the first few in detail: the oracle's segments against the device's
        a0, a1 = int(out["job_aln_off"][r * na]), int(out["job_aln_off"][(r + 1) * na])
        print("read", r, "device:", [(int(out["strand"][a]), int(out["chrom"][a]), int(out["block_off"][a + 1] - out["block_off"][a]), out["blocks"][int(out["block_off"][a])].tolist(),
                                      out["blocks"][int(out["block_off"][a + 1]) - 1].tolist()) for a in range(a0, a1)])
        rd = reads_h[int(off_h[r]):int(off_h[r + 1])].tobytes()
        try:
            groups, unal = OP.map_read_lowacc(rd, g, key, pos, g_index, oo, chrom_pos=mapper.chrom_pos)
            print("read", r, "oracle:", "unaligned" if unal else "", [[(s["strand"], s["chrom"], len(s["blocks"]), s["blocks"][0].tolist(), s["blocks"][-1].tolist()) for s in segs] for segs in groups])
        except Exception as e:                                                  # (the single-read composition's signature differs between presets: the list above is what matters)
            print("read", r, "oracle detail unavailable:", repr(e)[:200])
        # where the oracle's composition loses the read: the Python composition with every stage function's outcome logged
        names = ["split_chain", "refine_splitchain", "refine_btwn_splitchain", "merge_extend", "sdp_chain", "local_refine_alignment", "filter_chain", "indel_refine"]
        orig = {nm: getattr(O, nm) for nm in names}

        def wrap(nm):
            def f(*a, **k):
                v = orig[nm](*a, **k)
                def brief(x):
                    if x is None: return None
                    if isinstance(x, dict): return {kk: (len(vv) if hasattr(vv, "__len__") else vv) for kk, vv in list(x.items())[:8]}
                    if isinstance(x, (list, tuple)): return [brief(y) if isinstance(y, dict) else (len(y) if hasattr(y, "__len__") else y) for y in x[:4]]
                    return x
                print("   ", nm, "->", str(brief(v))[:300])
                return v
            return f
        for nm in names: setattr(O, nm, wrap(nm))
        try:
            al, un = OP.map_read_lowacc_py(rd, g, key, pos, g_index, oo, chrom_pos=mapper.chrom_pos)
            print("read", r, "python composition:", "unaligned" if un else "", [len(x) for x in al])
        except Exception as e:
            print("read", r, "python composition failed:", repr(e)[:300])
        for nm in names: setattr(O, nm, orig[nm])
        with open(os.path.join(ROOT, "gpurun_out", "mismatch_%s_read%d.fa" % (args.preset, r)), "w") as f:
            f.write(">read%d\n%s\n" % (r, rd.decode()))


if __name__ == "__main__":
    main()
