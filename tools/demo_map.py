"""Map a few simulated reads with the -ONT low-accuracy path and print the SAM records (needs the GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lra_amd import synth, seed, mapread
from lra_amd.context import Context

ctx = Context(0)
genome = synth.make_genome(800_000, seed=77, repeat_frac=0.2, n_families=3)
CH = [0, 350_000, 800_000]
o = mapread.LowAccOptions()
ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
reads, truth = synth.simulate_reads(genome, 6, 3000, 500, 0.10, seed=5)
mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chrA", b"chrB"], CH, o)
res = mapper.align(seed.ReadBatch(ctx, [r.tobytes() for r in reads]))
sys.stdout.write(mapper.sam_header(b"0.1", b"demo_map.py").decode())
for t, tr in zip(mapper.records(res, [b"r%d" % i for i in range(len(reads))], [r.tobytes() for r in reads]), truth):
    for l in t.decode().split("\n"):
        if l:
            f = l.split("\t")
            print("\t".join(f[:5] + [f[5][:60] + "...", f[6], f[7], f[8], f[9][:20] + "..."] + f[10:]), " # truth", tr)
