"""CPU-only: reads drawn from an alpha-satellite-like array through the oracle's MapRead_lowacc with ORACLE_SDP_DUMP set, so that the inputs of
the heavy sparse-DP calls (SDP#A, the per-merged-cluster sparse DP) can be studied offline (tools/sdp_case_stats.py).  Test infrastructure."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sdp_dump.bin"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if os.path.exists(out):
    os.remove(out)
os.environ["ORACLE_SDP_DUMP"] = out
import oracle_lib as O
import oracle_pipeline as OP
from lra_amd import synth

rng = np.random.default_rng(7)
B = np.frombuffer(b"ACGT", dtype=np.uint8)
flank = synth.make_genome(1_000_000, seed=5, repeat_frac=0.3, n_families=4)
mono = B[rng.integers(0, 4, 171)]
alen = int(os.environ.get("SAT_LEN", "300000"))
arr = np.tile(mono, (alen + 170) // 171)[:alen].copy()
mut = rng.random(alen) < 0.02
arr[mut] = B[rng.integers(0, 4, int(mut.sum()))]
genome = np.concatenate([flank[:600_000], arr, flank[600_000:]])
G = len(genome)
gb = genome.tobytes()
t0 = time.time()
ik, ip, st = O.store_index(gb, [0, G], 17, 10, 150, 15, 1)
print("index", len(ik), "entries in %.1fs" % (time.time() - t0), flush=True)
tup, bnd = O.local_index_seq(gb, 10, 5, 256, 15)
g_index = (OP.seq_offsets(G, 256), bnd, tup)
gbytes = gb + b"\0" * 64
for i in range(n_reads):
    a = 600_000 + int(rng.integers(20_000, alen - 60_000))
    n = 30_000
    rd = synth.simulate_read(rng, genome[a:a + n + 3000], n, 0.10, (30, 35, 35), bool(i & 1))[0]
    t0 = time.time()
    alns, un = OP.map_read_lowacc(rd.tobytes(), gbytes, ik, ip, g_index)
    print("read", i, "len", len(rd), "unaligned", un, "segs", [len(x) for x in alns], "%.1fs" % (time.time() - t0), flush=True)
print("dump", out, os.path.getsize(out))
