"""The one call + the library's record stage against the same stages driven one call at a time from Python + the Python mirror of the record stage, on reads of the
bench's workload: the SAM text of every read.  usage (on the GPU box): python tools/staged_vs_onecall.py [--preset ont|clr] [--reads N] [--sv-frac F]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", choices=["ont", "clr"], default="ont")
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--sv-frac", type=float, default=0.5)
    args = ap.parse_args()
    import torch
    from lra_amd.context import Context
    from lra_amd import seed, mapread, synth_genome as sg
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=1.0, seed=3)
    ctx = Context(0)
    if args.preset == "clr":
        mopts = mapread.with_gli(mapread.clr_options()); ip = (15, 10, 250, 12, 1); rl, err, mix = 20000, 0.15, (20, 30, 50)
    else:
        mopts = mapread.with_gli(mapread.LowAccOptions()); ip = (17, 10, 150, 12, 1); rl, err, mix = 30000, 0.10, (30, 35, 35)
    mapper = mapread.LowAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, mopts, index_params=ip, staged=True)
    sim = sg.simulate_reads_sv(genome, chrom_pos, args.reads, rl, rl / 10, err, mix, 1000, sv_frac=args.sv_frac)
    off_h = sim["off"].cpu().numpy(); total = int(off_h[-1])
    reads_h = sim["seq"][:total].cpu().numpy().tobytes()
    lseq = torch.cat([sim["seq"][:total], torch.zeros(64, dtype=torch.uint8, device=dev)])
    rbatch = seed.read_batch_from_device(ctx, lseq, sim["off"].contiguous())
    names = [b"read%d" % i for i in range(args.reads)]
    reads = [reads_h[int(off_h[i]):int(off_h[i + 1])] for i in range(args.reads)]
    one = mapper.records(mapper.align(rbatch), names, reads)
    staged = mapper.records_staged(mapper.align_staged(rbatch), names, reads)
    strip = lambda t: b"\n".join(b"\t".join(f for f in l.split(b"\t") if not f.startswith(b"RT:i:")) for l in t.split(b"\n"))
    bad = [i for i in range(args.reads) if strip(one[i]) != strip(staged[i])]
    n_lines = sum(t.count(b"\n") for t in one)
    n_supp = sum(1 for t in one for l in t.split(b"\n") if l and int(l.split(b"\t")[1]) & 2048)
    print(json.dumps({"preset": args.preset, "reads": args.reads, "sam_lines": n_lines, "supplementary_lines": n_supp, "reads_whose_text_differs": len(bad), "first": bad[:5]}))
    for i in bad[:2]:
        a, b = one[i].split(b"\n"), staged[i].split(b"\n")
        print("read", i, len(a), len(b))
        for x, y in zip(a, b):
            if strip(x) != strip(y):
                fx, fy = x.split(b"\t"), y.split(b"\t")
                print("  differs at fields", [k for k in range(min(len(fx), len(fy))) if fx[k] != fy[k]][:8], fx[:9], fy[:9])
                break


if __name__ == "__main__":
    main()
