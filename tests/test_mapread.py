"""MapRead_lowacc end to end through lra_amd.mapread (GPU): simulated reads come back as text records at the locus they were drawn from.
Every stage has its own parity test against the oracle; this one checks the wiring between them and the record bookkeeping."""
import re

import numpy as np
import pytest

from lra_amd import synth


def _parse_cigar(c):
    return [(int(n), op) for n, op in re.findall(r"(\d+)([=XIDSHMN])", c)]


@pytest.mark.gpu
def test_map_reads_to_sam(ctx):
    from lra_amd import seed, mapread
    genome = synth.make_genome(800_000, seed=77, repeat_frac=0.2, n_families=3)
    CH = [0, 350_000, 800_000]
    names = [b"chrA", b"chrB"]
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    reads, truth = synth.simulate_reads(genome, 40, 7000, 2000, 0.10, seed=5)
    # keep reads inside one chromosome; add a read that cannot align and a chimeric one (two loci -> supplementary records)
    keep = [i for i, (s, l, st) in enumerate(truth) if not (s < CH[1] < s + l)]
    reads = [reads[i] for i in keep]; truth = [truth[i] for i in keep]
    rng = np.random.default_rng(3)
    junk = rng.integers(0, 4, 3000).astype(np.uint8)
    reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[junk].copy()); truth.append(None)
    a = synth.simulate_read(rng, genome[100_000:105_001], 4500, 0.08, (30, 35, 35), False)[0]
    b = synth.simulate_read(rng, genome[600_000:605_001], 4500, 0.08, (30, 35, 35), False)[0]
    reads.append(np.concatenate([a, b])); truth.append("chimera")
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, names, CH, o)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    rnames = [b"read%d" % i for i in range(len(reads))]
    # the same stages driven one library call at a time (what the stage parity tests hook into) ...
    sres = mapper.align_staged(batch)
    staged = mapper.records_staged(sres, rnames, [r.tobytes() for r in reads])
    staged_blocks = sres.block_records.cpu().numpy()
    # ... and behind the C boundary: one call for the device side, one for the records
    res = mapper.align(batch)
    assert np.array_equal(mapper.block_records(res).cpu().numpy(), staged_blocks)
    texts = mapper.records(res, rnames, [r.tobytes() for r in reads])
    assert texts == staged
    for fmt in "pPb":
        mapper.opts.printFormat = fmt; mapper.copts = mapper._c_opts()
        assert mapper.records(res, rnames, [r.tobytes() for r in reads]) == mapper.records_staged(sres, rnames, [r.tobytes() for r in reads]), fmt
    mapper.opts.printFormat = "s"; mapper.copts = mapper._c_opts()
    assert len(texts) == len(reads)
    hdr = mapper.sam_header(b"test", b"lra align")
    assert hdr.count(b"@SQ") == 2 and b"SN:chrB\tLN:450000" in hdr
    n_right = 0
    for i, (t, tr) in enumerate(zip(texts, truth)):
        lines = [l for l in t.decode().split("\n") if l]
        assert lines, i
        f = lines[0].split("\t")
        assert f[0] == "read%d" % i
        if tr is None:
            assert int(f[1]) & 4, lines[0][:200]                                  # unaligned
            continue
        if tr == "chimera":
            chroms = {l.split("\t")[2] for l in lines}
            assert chroms == {"chrA", "chrB"}, chroms
            assert sum(1 for l in lines if int(l.split("\t")[1]) & 2048) == len(lines) - 1
            continue
        s, l, st = tr
        flag, chrom, pos, mapq, cigar = int(f[1]), f[2], int(f[3]), int(f[4]), f[5]
        ci = 0 if s < CH[1] else 1
        ops = _parse_cigar(cigar)
        qlen = sum(n for n, op in ops if op in "=XISH")
        tlen = sum(n for n, op in ops if op in "=XD")
        assert qlen == len(reads[i]), (i, qlen, len(reads[i]))
        assert f[9] == "*" or len(f[9]) == sum(n for n, op in ops if op in "=XIS")
        ok = (chrom == names[ci].decode() and bool(flag & 16) == bool(st) and abs((pos - 1 + CH[ci]) - s) < 200
              and abs(tlen - l) < 400)
        n_right += ok
        tags = dict(x.split(":", 2)[::2] for x in f[11:] if x.count(":") >= 2)
        assert "NM" in tags
    n_sim = sum(1 for t in truth if isinstance(t, tuple))
    assert n_right >= 0.9 * n_sim, (n_right, n_sim)
