import numpy as np, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
from lra_amd import synth
from lra_amd.context import Context
import test_refine as T
ctx = Context(0)
genome = synth.make_genome(300000, seed=22)
reads, blocks = T.make_cases(14, 6, 20000, 0.12, (30, 35, 35), genome, drop=0.0, trim=0.5)
rng = np.random.default_rng(5)
for b in blocks:
    m = rng.random(len(b)) < 0.02
    m[0] = False
    b[m, 0] -= rng.integers(1, 3, size=int(m.sum())).astype(b.dtype)
b = blocks[-1]
b[len(b) // 2, 1] -= 8
(got, status), res = T._run_gpu(ctx, genome, reads, blocks, 7, (4, -1, -2))
g = genome.tobytes()
for i in range(len(reads)):
    exp, st = O.indel_refine(blocks[i], reads[i].tobytes(), g, 7, 4, -1, -2)
    if st == 0 and not np.array_equal(got[i], exp):
        n = min(len(got[i]), len(exp))
        d = np.nonzero((got[i][:n] != exp[:n]).any(axis=1))[0]
        print("read", i, "len", len(reads[i]), "first diff at", d[:5], len(got[i]), len(exp))
        j = d[0]
        print("got", got[i][j - 2:j + 8].tolist())
        print("exp", exp[j - 2:j + 8].tolist())
        q0 = exp[j - 2][0]
        bi = blocks[i]
        k = np.searchsorted(bi[:, 0], q0)
        print("in ", bi[max(0, k - 3):k + 10].tolist())
