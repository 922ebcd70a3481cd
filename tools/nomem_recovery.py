"""After LRA_ERR_NOMEM a context must stay usable: a batch too large for the device fails loudly, a smaller one on the SAME context then maps as on a fresh context.
usage (on the GPU box): python tools/nomem_recovery.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from lra_amd.context import Context
    from lra_amd._lib import LraError
    from lra_amd import seed, mapread, synth_genome as sg
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=0.05, seed=3)
    sim_big = sg.simulate_reads_sv(genome, chrom_pos, 28672, 30000, 3000, 0.10, (30, 35, 35), 1000, sv_frac=0.3)
    sim_small = sg.simulate_reads_sv(genome, chrom_pos, 2048, 30000, 3000, 0.10, (30, 35, 35), 1001, sv_frac=0.3)

    def batch(ctx, sim):
        off = sim["off"]; total = int(off[-1])
        lseq = torch.cat([sim["seq"][:total], torch.zeros(64, dtype=torch.uint8, device=dev)])
        return seed.read_batch_from_device(ctx, lseq, off.contiguous()), lseq

    def digest(mapper, res):
        out = mapper.fetch(res)
        import hashlib
        h = hashlib.sha256()
        for k in ("job_aln_off", "read_status", "strand", "chrom", "block_off", "blocks", "counts", "job_reached"):
            h.update(np.ascontiguousarray(out[k]).tobytes())
        return h.hexdigest(), int(res.n_alignments)
    mopts = mapread.with_gli(mapread.LowAccOptions())
    ctx = Context(0)
    mapper = mapread.LowAccMapper(ctx, genome, None, None, chrom_names, chrom_pos, mopts, index_params=(17, 10, 150, 12, 1), staged=False)
    rb_big, keep1 = batch(ctx, sim_big)
    err = None
    try:
        mapper.align(rb_big)
    except LraError as e:
        err = str(e)
    del rb_big, keep1
    free_before = torch.cuda.mem_get_info()[0]
    freed = ctx.release_buffers()                                             # what the failed batch had grown stays with the context until it is asked for
    free_after = torch.cuda.mem_get_info()[0]
    rb, keep2 = batch(ctx, sim_small)
    d1 = digest(mapper, mapper.align(rb))
    ctx2 = Context(0)
    mapper2 = mapread.LowAccMapper(ctx2, genome, None, None, chrom_names, chrom_pos, mopts, index_params=(17, 10, 150, 12, 1), staged=False)
    rb2, keep3 = batch(ctx2, sim_small)
    d2 = digest(mapper2, mapper2.align(rb2))
    print(json.dumps({"large_batch_error": err, "released_gb": round(freed / 1e9, 1), "device_free_gb": [round(free_before / 1e9, 1), round(free_after / 1e9, 1)], "after_error": d1, "fresh_context": d2, "equal": d1 == d2}))


if __name__ == "__main__":
    main()
