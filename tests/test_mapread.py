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
    # the same in two halves (snapshot on the host, then host-only formatting), and through the packed record buffer a rank sends to rank 0
    rargs = mapper.record_args(rnames, [r.tobytes() for r in reads])
    assert mapper.records_host(mapper.snapshot(res), rargs) == texts
    import ctypes as C_
    from lra_amd import parallel
    d_buf, nb = C_.c_void_p(), C_.c_uint64(0)
    ctx.check(ctx.lib.lra_map_pack(ctx.h, C_.byref(res), 0, C_.byref(d_buf), C_.byref(nb)))
    packed = ctx.to_host(d_buf.value, nb.value, np.uint8)
    assert parallel.records_from_packed(ctx.lib, mapper.copts, packed, rnames, [r.tobytes() for r in reads], names) == texts
    # two ranks (here: two batches, one after the other) own hash-partitioned ordinals; rank 0 merges their record buffers by ordinal
    per_rank, ords = [], []
    for rk in range(2):
        o_ = parallel.shard_ordinals(len(reads), rk, 2)
        rb = seed.ReadBatch(ctx, [reads[i].tobytes() for i in o_])
        rr = mapper.align(rb)
        ctx.check(ctx.lib.lra_map_pack(ctx.h, C_.byref(rr), 0, C_.byref(d_buf), C_.byref(nb)))
        pk = ctx.to_host(d_buf.value, nb.value, np.uint8)
        per_rank.append(parallel.records_from_packed(ctx.lib, mapper.copts, pk, [rnames[i] for i in o_], [reads[i].tobytes() for i in o_], names))
        ords.append(o_)
    assert parallel.merge_by_ordinal(per_rank, ords, len(reads)) == texts
    res = mapper.align(batch)
    import os
    os.environ["LRA_RECORD_THREADS"] = "1"                                  # one host thread or many: the same text
    try:
        assert mapper.records(res, rnames, [r.tobytes() for r in reads]) == texts
        os.environ["LRA_RECORD_THREADS"] = "7"
        assert mapper.records(res, rnames, [r.tobytes() for r in reads]) == texts
    finally:
        del os.environ["LRA_RECORD_THREADS"]
    for fmt in "pPb":
        mapper.opts.printFormat = fmt; mapper.copts = mapper._c_opts()
        assert mapper.records(res, rnames, [r.tobytes() for r in reads]) == mapper.records_staged(sres, rnames, [r.tobytes() for r in reads]), fmt
    # print format "a": PrintPairwise of every printed SegAlignment, against the strings built here from the fetched blocks
    import ctypes as C
    from lra_amd._lib import load_library
    lib = load_library()
    mapper.opts.printFormat = "a"; mapper.copts = mapper._c_opts()
    pw = mapper.records(res, rnames, [r.tobytes() for r in reads])
    fo = mapper.fetch(res)
    na_ = int(res.num_aln)
    n_pw = 0
    for i, rd in enumerate(reads):
        j = i * na_
        a0, a1 = int(fo["job_aln_off"][j]), int(fo["job_aln_off"][j + 1])                       # PrintNumAln = 1: the best group = the first job here
        if a1 == a0:
            assert pw[i] == b""
            continue
        if sum(int(fo["job_aln_off"][i * na_ + p + 1] - fo["job_aln_off"][i * na_ + p]) > 0 for p in range(na_)) > 1:
            continue                                                                              # several candidate groups: ordering tested elsewhere
        exp = b""
        for a in range(a1 - 1, a0 - 1, -1):                                                     # segments are printed last to first
            b = np.ascontiguousarray(fo["blocks"][int(fo["block_off"][a]):int(fo["block_off"][a + 1])], np.int32)
            sread = rd.tobytes() if fo["strand"][a] == 0 else synth.revcomp(rd).tobytes()
            ci = int(fo["chrom"][a]); text = genome[CH[ci]:CH[ci + 1]].tobytes()
            n = C.c_uint64(0); rl = C.c_uint32(0)
            bp = b.ctypes.data_as(C.c_void_p)
            lib.lra_alignment_strings(sread, text, bp, len(b), None, None, None, C.c_uint64(0), C.byref(n), C.byref(rl))
            qb = C.create_string_buffer(n.value + 1); ab = C.create_string_buffer(n.value + 1); tb = C.create_string_buffer(n.value + 1)
            assert lib.lra_alignment_strings(sread, text, bp, len(b), qb, ab, tb, n, C.byref(n), C.byref(rl)) == 0
            p_ = C.c_uint64(0)
            args = (rnames[i], names[ci], len(b), int(b[0, 0]), int(b[0, 1]), rl, qb.raw[:n.value], ab.raw[:n.value], tb.raw[:n.value], n)
            lib.lra_format_pairwise(*args, None, C.c_uint64(0), C.byref(p_))
            pb = C.create_string_buffer(p_.value + 1)
            assert lib.lra_format_pairwise(*args, pb, p_, C.byref(p_)) == 0
            exp += pb.raw[:p_.value]
        assert pw[i] == exp, i
        n_pw += 1
    assert n_pw >= 20
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
        # SEQ is Alignment::read = strands[str] (Map_lowacc.h:560, Alignment.h:506-507): the read as it came on the forward strand, its reverse complement on the reverse
        # strand; a hard-clipped supplementary record carries the aligned part of THAT strand's sequence (Alignment.h:708-711)
        for ln_ in lines:
            ff = ln_.split("\t")
            fl_, cg_ = int(ff[1]), _parse_cigar(ff[5])
            sread = mapread.create_rc(reads[i].tobytes()) if fl_ & 16 else reads[i].tobytes()
            h0 = cg_[0][0] if cg_[0][1] == "H" else 0
            h1 = cg_[-1][0] if cg_[-1][1] == "H" else 0
            assert ff[9].encode() == sread[h0:len(sread) - h1], (i, fl_, h0, h1)
        ok = (chrom == names[ci].decode() and bool(flag & 16) == bool(st) and abs((pos - 1 + CH[ci]) - s) < 200
              and abs(tlen - l) < 400)
        n_right += ok
        tags = dict(x.split(":", 2)[::2] for x in f[11:] if x.count(":") >= 2)
        assert "NM" in tags
    n_sim = sum(1 for t in truth if isinstance(t, tuple))
    assert n_right >= 0.9 * n_sim, (n_right, n_sim)


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["ont", "clr", "ont-bp", "ont-2chr", "ont-rep", "ont-defer", "ont-defer-all", "ont-onepass", "ont-gli", "clr-gli", "ont-gli-2chr"])
def test_map_reads_match_oracle_pipeline(ctx, oracle, preset):
    """The C boundary against the oracle's stage functions composed on the CPU (tests/oracle_pipeline.py): every SegAlignment of every
    primary chain -- strand, Supplymentary, NumOfAnchors0/1, FirstSDPValue, the refined blocks -- bit for bit, on plain reads, reads with a
    deletion / an inversion / a translocated half, and a read that cannot align."""
    import oracle_lib
    import oracle_pipeline as OP
    from lra_amd import seed, mapread
    O_STAT_NAMES = oracle_lib.STAT_NAMES
    genome = synth.make_genome(500_000, seed=31, repeat_frac=0.25, n_families=3)
    gli = preset.endswith("-gli") or "-gli-" in preset                     # the local index as `lra index` writes it and glIndex.Read hands it to the path: windows of 2048 bases
    preset = preset.replace("-gli", "")                                    # (LocalIndex(0), MMIndex.h:110-127, lra.cpp:989); without a .gli file `lra align` builds it with opts.localIndexWindow = 256
    o = mapread.clr_options() if preset == "clr" else mapread.LowAccOptions(refineBreakpoint=(preset == "ont-bp"))     # ont-bp: --refineBreakpoints
    o.localIndexWindow = 2048 if gli else 256
    # ont-defer / -all / onepass: lra_map_opts.defer_matches -- some / all / none of the reads go through the driver's second, concurrent pass; same results
    o.deferMatches = {"ont-defer": 800, "ont-defer-all": 1, "ont-onepass": 0}.get(preset)
    oo = dict(OP.CLR if preset == "clr" else dict(OP.ONT, refineBreakpoint=(preset == "ont-bp")), localIndexWindow=o.localIndexWindow)
    err, mix = (0.15, (20, 30, 50)) if preset == "clr" else (0.10, (30, 35, 35))
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    reads, truth = synth.simulate_reads(genome, 10, 8000, 2500, err, mix, seed=11)
    rng = np.random.default_rng(2)
    sim = lambda a, n, rev=False: synth.simulate_read(rng, genome[a:a + n + 1], n, err * 0.8, mix, rev)[0]
    reads.append(np.concatenate([sim(50_000, 4000), sim(60_000, 4000)]))                       # 6 kb deletion
    reads.append(np.concatenate([sim(150_000, 4000), sim(154_000, 2500, True), sim(156_500, 4000)]))   # inversion
    reads.append(np.concatenate([sim(250_000, 4500), sim(400_000, 4500, True)]))               # translocation, second half reversed
    reads.append(synth.revcomp(np.concatenate([sim(300_000, 3000), sim(303_200, 3000)])))      # 200 bp deletion, read on the reverse strand
    reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2500)].copy())      # junk
    CH = [0, 200_100, len(genome)] if preset == "ont-2chr" else [0, len(genome)]      # two chromosomes: the deletion / inversion reads lie in the first, the translocation spans both
    if preset == "ont-rep":
        # low-error 18 kb reads (>= 500 anchors per cluster) inside and outside a segmental duplication (24 kb, 0.5 % diverged): the first
        # sparse DP runs with anchor bonus 3 where a cluster's anchorfreq lies in (1, 2] (Map_lowacc.h:86-89, :184-185), else with 20
        genome = genome.copy()
        dup = genome[100_000:124_000].copy()
        flip = rng.random(len(dup)) < 0.005
        dup[flip] = np.frombuffer(b"CGTA", np.uint8)[np.searchsorted(np.frombuffer(b"ACGT", np.uint8), dup[flip])]     # A->C, C->G, G->T, T->A
        genome[330_000:354_000] = dup
        ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
        for a0 in (101_000, 104_000, 332_000, 336_000, 200_000, 420_000):
            reads.append(synth.simulate_read(rng, genome[a0:a0 + 18_001], 18_000, 0.02, mix, a0 == 104_000)[0])
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr%d" % (i + 1) for i in range(len(CH) - 1)], CH, o)
    res = mapper.align(seed.ReadBatch(ctx, [r.tobytes() for r in reads]))
    out = mapper.fetch(res)
    na = int(res.num_aln)
    nd = mapper.stats["n_deferred_reads"]
    assert {"ont-defer": 0 < nd < len(reads) - 1, "ont-defer-all": nd == len(reads) - 1, "ont-onepass": nd == 0}.get(preset, True), (nd, len(reads))
    g_win, g_bnd, g_tup = mapper.gli.fetch()
    g_index = (mapread.seq_offsets(CH, o.localIndexWindow).astype(np.uint64), g_bnd, g_tup)
    gbytes = genome.tobytes() + b"\0" * 64
    n_seg = n_supp = n_rev = n_multi = n_bp = 0
    rates = []
    for r, rd in enumerate(reads):
        exp, unaligned = OP.map_read_lowacc(rd.tobytes(), gbytes, ik, ip, g_index, oo, chrom_pos=CH)
        rates.append(OP.TRACE.get("match_rate"))
        assert out["read_status"][r] == 0, (r, out["read_status"][r])
        for p in range(na):
            a0, a1 = int(out["job_aln_off"][r * na + p]), int(out["job_aln_off"][r * na + p + 1])
            e = exp[p] if p < len(exp) else []
            assert a1 - a0 == len(e), (r, p, a1 - a0, len(e))
            if p > 0 or not unaligned:                                                        # which chains reach Map_lowacc.h:574
                assert bool(out["job_reached"][r * na + p]) == (p < len(exp)), (r, p, len(exp))
            for a, s in zip(range(a0, a1), e):
                assert (out["strand"][a], out["supp"][a], out["n0"][a], out["n1"][a], out["chrom"][a]) == (s["strand"], s["supp"], s["n0"], s["n1"], s["chrom"]), (r, p, a)
                assert np.float32(out["first_sdp_value"][a]).view(np.uint32) == np.float32(s["value"]).view(np.uint32), (r, p, a)
                assert out["refine_status"][a] == s["refine_status"] == 0, (r, p, a)
                b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
                assert np.array_equal(b, s["blocks"]), (r, p, a, len(b), len(s["blocks"]))
                ec, ev, eruns, _ = s["stats"]                                                  # CalculateStatistics: counters, NV bits, CIGAR runs
                assert out["counts"][a].tolist() == [ec[k] for k in O_STAT_NAMES], (r, p, a)
                assert np.float32(out["value"][a]).view(np.uint32) == np.float32(ev).view(np.uint32), (r, p, a)
                assert np.array_equal(out["runs"][int(out["run_off"][a]):int(out["run_off"][a + 1])], eruns), (r, p, a)
                n_seg += 1; n_supp += int(s["supp"]); n_rev += int(s["strand"]); n_bp += int(s.get("breakpoint", 0) == 1)
            n_multi += len(e) > 1
        if unaligned:
            assert out["job_aln_off"][r * na + 1] == out["job_aln_off"][r * na], r
    assert n_seg >= len(reads) - 1 and n_supp >= 2 and n_rev >= 3 and n_multi >= 2, (n_seg, n_supp, n_rev, n_multi)
    assert preset != "ont-bp" or n_bp >= 1, n_bp
    if preset == "ont-rep":
        # anchorfreq = matches / distinct read k-mers of a diagonal run (AVGfreq, Clustering.h:550): one repeated k-mer among >= 500 anchors
        # puts a low-error read just above 1 (bonus 3), the reads inside the duplication just above 2 (bonus 20), 10 % error reads have
        # fewer than 500 anchors per cluster (bonus 20).  Both values must occur, and the device computes the same rates.
        assert 3.0 in rates and 20.0 in rates, rates
        import ctypes as C
        from lra_amd import cluster
        batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
        seed.seed_batch(ctx, batch, o.globalK, o.globalW, o.globalMaxFreq)
        cres = cluster.clean_matches_batch(ctx, mapper.clean_opts, CH)
        d_rate = C.c_void_p()
        ctx.check(ctx.lib.lra_match_rate_batch(ctx.h, C.byref(cres), C.c_float(o.initial_anchorbonus), C.byref(d_rate)))
        got = ctx.to_host(d_rate.value, len(reads), np.float32)
        assert got.tolist() == [float(x) if x is not None else 20.0 for x in rates], (got.tolist(), rates)


def _same_alignments(a, b):
    if len(a) != len(b):
        return False
    for ga, gb in zip(a, b):
        if len(ga) != len(gb):
            return False
        for x, y in zip(ga, gb):
            if any(x[k] != y[k] for k in ("strand", "supp", "secondary", "n0", "n1", "chrom", "refine_status")) or x.get("breakpoint") != y.get("breakpoint"):
                return False
            if np.float32(x["value"]).view(np.uint32) != np.float32(y["value"]).view(np.uint32):
                return False
            if not np.array_equal(x["blocks"], y["blocks"]) or not np.array_equal(x["a13_blocks"], y["a13_blocks"]) or ("stats" in x) != ("stats" in y):
                return False
            if "stats" in x and (x["stats"][0] != y["stats"][0] or not np.array_equal(x["stats"][2], y["stats"][2]) or
                                 np.float32(x["stats"][1]).view(np.uint32) != np.float32(y["stats"][1]).view(np.uint32)):
                return False
    return True


def test_oracle_cpp_pipeline_equals_python_composition(oracle):
    """oracle/pipeline.cpp (what the GPU path is compared with, and what bench.py's cpu_baseline times on all host cores) against the same
    composition written in Python over the stage wrappers: plain reads, a deletion, an inversion, a translocation, junk; two chromosomes; with
    and without --refineBreakpoints; and the thread pool gives the same checksum on 1 and 4 threads."""
    import oracle_lib as O
    import oracle_pipeline as OP
    genome = synth.make_genome(400_000, seed=31, repeat_frac=0.25, n_families=3)
    CH = [0, 200_100, len(genome)]
    ik, ip = synth.build_global_index(genome, 17, 10, 100)
    tups, bnd = [], [0]
    for c in range(2):
        t, b = O.local_index_seq(genome[CH[c]:CH[c + 1]].tobytes(), 10, 5, 256, 15)
        tups.append(t); bnd.extend((b[1:] + bnd[-1]).tolist())
    g_index = (OP.seq_offsets_multi(CH, 256), np.array(bnd, np.uint64), np.concatenate(tups))
    gbytes = genome.tobytes() + b"\0" * 64
    reads, _ = synth.simulate_reads(genome, 4, 7000, 2000, 0.10, (30, 35, 35), seed=11)
    rng = np.random.default_rng(2)
    sim = lambda a, n, rev=False: synth.simulate_read(rng, genome[a:a + n + 1], n, 0.08, (30, 35, 35), rev)[0]
    reads.append(np.concatenate([sim(50_000, 3000), sim(59_000, 3000)]))
    reads.append(np.concatenate([sim(150_000, 3000), sim(153_000, 2500, True), sim(155_500, 3000)]))
    reads.append(np.concatenate([sim(250_000, 3500), sim(100_000, 3500, True)]))
    reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2500)].copy())
    n_seg = 0
    for oo in (None, dict(refineBreakpoint=True)):
        for rd in reads:
            a, u = OP.map_read_lowacc_py(rd.tobytes(), gbytes, ik, ip, g_index, oo, chrom_pos=CH)
            b, v = OP.map_read_lowacc(rd.tobytes(), gbytes, ik, ip, g_index, oo, chrom_pos=CH)
            assert u == v and _same_alignments(a, b), (u, v, [len(x) for x in a], [len(x) for x in b])
            n_seg += sum(len(x) for x in b)
    assert n_seg >= 16
    allr = np.concatenate(reads); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    r1 = OP.map_reads_lowacc_mt(allr, off, 0, len(reads), gbytes, ik, ip, g_index, chrom_pos=CH, n_threads=1)
    r4 = OP.map_reads_lowacc_mt(allr, off, 0, len(reads), gbytes, ik, ip, g_index, chrom_pos=CH, n_threads=4)
    assert r1["checksum"] == r4["checksum"] and r1["n_alignments"] == r4["n_alignments"] >= 8 and r1["bases"] == int(off[-1])


def test_oracle_pipeline_sanity(oracle):
    """The oracle's MapRead_lowacc composition on the CPU: simulated reads come back at their locus with a plausible alignment (the
    checker itself; the GPU path is compared with it above)."""
    import oracle_lib as O
    import oracle_pipeline as OP
    genome = synth.make_genome(300_000, seed=13, repeat_frac=0.2, n_families=2)
    ik, ip = synth.build_global_index(genome, 17, 10, 100)
    tup, bnd = O.local_index_seq(genome.tobytes(), 10, 5, 256, 15)
    g_index = (OP.seq_offsets(len(genome), 256), bnd, tup)
    gbytes = genome.tobytes() + b"\0" * 64
    reads, truth = synth.simulate_reads(genome, 5, 6000, 1500, 0.10, seed=4)
    rng = np.random.default_rng(1)
    reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2000)].copy()); truth.append(None)
    for rd, tr in zip(reads, truth):
        alns, unaligned = OP.map_read_lowacc(rd.tobytes(), gbytes, ik, ip, g_index)
        if tr is None:
            assert unaligned
            continue
        s, l, st = tr
        assert not unaligned and len(alns[0]) >= 1
        a = alns[0][0]
        b = a["blocks"]
        assert a["strand"] == st and a["refine_status"] == 0
        assert abs(int(b[0, 1]) - s) < 300 and abs(int(b[-1, 1] + b[-1, 2]) - (s + l)) < 300
        assert b[:, 2].sum() > 0.7 * len(rd)
        assert np.all(b[:-1, 0] + b[:-1, 2] <= b[1:, 0]) and np.all(b[:-1, 1] + b[:-1, 2] <= b[1:, 1])


@pytest.mark.gpu
def test_map_reads_edge_batches(ctx, oracle):
    """Batches the path must survive: nothing alignable at all (no alignment arrays downstream), reads shorter than a k-mer / a window, an
    empty read between real ones, a single read; every read still gets its record (flag 4 when unaligned), aligned ones match the oracle."""
    import oracle_pipeline as OP
    from lra_amd import seed, mapread
    genome = synth.make_genome(200_000, seed=9, repeat_frac=0.1, n_families=2)
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], o)
    rng = np.random.default_rng(7)
    junk = lambda n: np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    good = synth.simulate_read(rng, genome[20_000:25_001], 5000, 0.10, (30, 35, 35), False)[0]
    g_win, g_bnd, g_tup = mapper.gli.fetch()
    g_index = (OP.seq_offsets(len(genome), 256), g_bnd, g_tup)
    gbytes = genome.tobytes() + b"\0" * 64
    for reads in ([junk(1500), junk(300)], [junk(12), good, np.zeros(0, np.uint8), junk(40), genome[100:120].copy()], [good]):
        batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
        res = mapper.align(batch)
        out = mapper.fetch(res)
        texts = mapper.records(res, [b"r%d" % i for i in range(len(reads))], [r.tobytes() for r in reads])
        na = max(int(res.num_aln), 1)
        for i, (rd, txt) in enumerate(zip(reads, texts)):
            exp, unaligned = OP.map_read_lowacc(rd.tobytes(), gbytes, ik, ip, g_index) if len(rd) else ([], True)
            f = txt.decode().split("\n")[0].split("\t")
            assert f[0] == "r%d" % i
            n_gpu = int(out["job_aln_off"][i * na + 1] - out["job_aln_off"][i * na]) if int(res.n_jobs) else 0
            assert (n_gpu == 0) == unaligned, (i, n_gpu, unaligned)
            assert bool(int(f[1]) & 4) == unaligned, (i, f[1])
            if not unaligned:
                a = int(out["job_aln_off"][i * na])
                b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])]
                assert np.array_equal(b, exp[0][0]["blocks"]), i


@pytest.mark.gpu
def test_map_reads_bench_like_properties(ctx, oracle):
    """The bench's read profile (30 kb, 10 % error, half of the reads reverse-complemented) on a 16 Mb genome through the C boundary:
    size-independent properties on every read (aligned at the locus it was drawn from, blocks colinear and inside the read / chromosome,
    CIGAR runs consistent with the blocks and the counters), and a sample compared with the oracle pipeline bit for bit."""
    import torch
    import oracle_pipeline as OP
    from lra_amd import synth_torch as st, seed, mapread
    dev = ctx.device
    NR = 512
    genome = st.make_genome(16_000_000, 1, dev)
    o = mapread.LowAccOptions()
    ik, ip = st.build_global_index(genome, o.globalK, o.globalW, 150)
    sim = st.simulate_batch(genome, NR, 30000, 3000, 0.10, (30, 35, 35), 1234)
    pad = torch.zeros(64, dtype=torch.uint8, device=dev)
    g2 = torch.Generator(device=dev).manual_seed(8)
    rev = torch.rand(NR, generator=g2, device=dev) < 0.5
    reads = torch.cat([st.revcomp_some(sim["seq"], sim["off"], rev), pad])
    G = int(genome.numel())
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, G], o)
    res = mapper.align(seed.read_batch_from_device(ctx, reads, sim["off"]))
    out = mapper.fetch(res)
    na = int(res.num_aln)
    off = sim["off"].cpu().numpy(); rev_h = rev.cpu().numpy()
    tb = sim["blocks"].cpu().numpy(); tbo = sim["block_off"].cpu().numpy()
    n_primary = 0
    for r in range(NR):
        L = int(off[r + 1] - off[r])
        a0, a1 = int(out["job_aln_off"][r * na]), int(out["job_aln_off"][r * na + 1])
        if a1 == a0:
            continue
        a = a0
        b = out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])].astype(np.int64)
        assert len(b) and out["refine_status"][a] == 0
        assert np.all(b[:, 2] >= 0) and np.all(b[:-1, 0] + b[:-1, 2] <= b[1:, 0]) and np.all(b[:-1, 1] + b[:-1, 2] <= b[1:, 1]), r
        assert b[0, 0] >= 0 and b[-1, 0] + b[-1, 2] <= L and b[-1, 1] + b[-1, 2] <= G, r
        assert int(out["strand"][a]) == int(rev_h[r]), r
        t0 = int(tb[tbo[r], 1]); t1 = int(tb[tbo[r + 1] - 1, 1] + tb[tbo[r + 1] - 1, 2])          # the locus the read was drawn from
        if a1 - a0 == 1:
            assert abs(int(b[0, 1]) - t0) < 500 and abs(int(b[-1, 1] + b[-1, 2]) - t1) < 500, (r, int(b[0, 1]), t0)
            n_primary += 1
        runs = out["runs"][int(out["run_off"][a]):int(out["run_off"][a + 1])].astype(np.int64)
        ln, op = runs >> 4, runs & 15
        c = out["counts"][a]
        assert ln[op <= 1].sum() == b[:, 2].sum() and ln[op == 0].sum() == c[0] and ln[op == 1].sum() == c[1], r       # '=' + 'X' columns = block bases
        assert ln[op <= 2].sum() == b[-1, 0] + b[-1, 2] - b[0, 0] and ln[(op <= 1) | (op == 3)].sum() == b[-1, 1] + b[-1, 2] - b[0, 1], r
    assert n_primary >= 0.97 * NR, n_primary
    g_win, g_bnd, g_tup = mapper.gli.fetch()
    g_index = (OP.seq_offsets(G, 256), g_bnd, g_tup)
    gbytes = genome.cpu().numpy().tobytes() + b"\0" * 64
    reads_h = reads.cpu().numpy()
    for r in range(0, NR, 64):
        exp, unaligned = OP.map_read_lowacc(reads_h[int(off[r]):int(off[r + 1])].tobytes(), gbytes, ik, ip, g_index)
        for p in range(na):
            a0, a1 = int(out["job_aln_off"][r * na + p]), int(out["job_aln_off"][r * na + p + 1])
            e = exp[p] if p < len(exp) else []
            assert a1 - a0 == len(e), (r, p)
            for a, s in zip(range(a0, a1), e):
                assert np.array_equal(out["blocks"][int(out["block_off"][a]):int(out["block_off"][a + 1])], s["blocks"]), (r, p)


@pytest.mark.gpu
def test_flagged_reads_are_counted_and_can_be_written_as_unaligned(ctx):
    """A read whose status word is non-zero gets no alignment record: counters.n_flagged_reads / lra_map_host_flagged report it, and with
    opts.flagged_unaligned the records keep one entry per input read (the read's unaligned record, output_unaligned)."""
    import ctypes as C
    import torch
    from lra_amd import seed, mapread
    genome = synth.make_genome(300_000, seed=41, repeat_frac=0.2, n_families=2)
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    reads, _ = synth.simulate_reads(genome, 5, 5000, 1000, 0.10, seed=9)
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], o)
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    names = [b"r%d" % i for i in range(len(reads))]
    res = mapper.align(batch)
    assert int(res.counters.n_flagged_reads) == 0
    base = mapper.records(res, names, [r.tobytes() for r in reads])
    assert all(len(x) > 0 and x.split(b"\t")[1] != b"4" for x in base)
    # flag read 2 the way a stage would (LRA_ST_CAPACITY = 8) on the device
    one = torch.tensor([8], dtype=torch.int32, device=ctx.device)
    ctx.check(ctx.lib.lra_copy_device(ctx.h, C.c_void_p(res.d_read_status + 2 * 4), C.c_void_p(one.data_ptr()), C.c_uint64(4)))
    torch.cuda.synchronize()
    snap = mapper.snapshot(res)
    st = C.POINTER(C.c_uint32)()
    assert ctx.lib.lra_map_host_flagged(snap, C.byref(st)) == 1 and [st[i] for i in range(5)] == [0, 0, 8, 0, 0]
    args = mapper.record_args(names, [r.tobytes() for r in reads])
    out0 = mapper.records_host(snap, args, free=False)
    assert out0[2] == b"" and [out0[i] for i in (0, 1, 3, 4)] == [base[i] for i in (0, 1, 3, 4)]
    mapper.copts.flagged_unaligned = 1
    out1 = mapper.records_host(snap, args, free=True)
    mapper.copts.flagged_unaligned = 0
    f = out1[2].split(b"\t")
    assert f[0] == b"r2" and f[1] == b"4" and out1[2].count(b"\n") == 1
    assert [out1[i] for i in (0, 1, 3, 4)] == [base[i] for i in (0, 1, 3, 4)]


@pytest.mark.gpu
def test_borrowed_reference_goes_stale_when_its_owner_reloads_or_dies():
    """lra_ctx_share_reference: a borrower's batch entry points refuse to run once the owner has reloaded its reference data, and once the owner is
    gone altogether (the check reads a generation cell both hold, never the owner's freed state: ADVICE round 3)."""
    from lra_amd import seed, mapread
    from lra_amd.context import Context
    from lra_amd._lib import LraError
    genome = synth.make_genome(120_000, seed=4, repeat_frac=0.1, n_families=2)
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    owner_ctx, b_ctx = Context(0), Context(0)
    owner = mapread.LowAccMapper(owner_ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], o)
    borrower = mapread.LowAccMapper.sharing(b_ctx, owner)
    rng = np.random.default_rng(1)
    rd = synth.simulate_read(rng, genome[30_000:34_001], 4000, 0.10, (30, 35, 35), False)[0]
    batch = seed.ReadBatch(b_ctx, [rd.tobytes()])
    res = borrower.align(batch)
    assert int(res.n_alignments) >= 1
    import ctypes as C
    cp = (C.c_uint64 * 2)(0, len(genome))
    owner_ctx.check(owner_ctx.lib.lra_ctx_load_chromosomes(owner_ctx.h, cp, 1))          # a loader of the owner: the borrower's copy is stale
    with pytest.raises(LraError, match="reloaded"):
        borrower.align(batch)
    b2_ctx = Context(0)
    owner_ctx.check(owner_ctx.lib.lra_ctx_build_local_index(owner_ctx.h, o.localK, o.localW, o.localIndexWindow, o.localMaxFreq))
    borrower2 = mapread.LowAccMapper.sharing(b2_ctx, owner)
    assert int(borrower2.align(seed.ReadBatch(b2_ctx, [rd.tobytes()])).n_alignments) >= 1
    owner_ctx.close()                                                                    # the owner's buffers are freed
    with pytest.raises(LraError, match="destroyed"):
        borrower2.align(seed.ReadBatch(b2_ctx, [rd.tobytes()]))
    b_ctx.close(); b2_ctx.close()


@pytest.mark.gpu
def test_reads_handed_back_by_tier1_matches_map_the_same_in_a_batch_of_their_own(ctx):
    """lra_map_opts.defer_seed_matches (scheduling only): reads with more tier-1 matches than the threshold leave the batch behind the seed stage -- status
    LRA_ST_DEFERRED and nothing else, counted in n_handed_back_reads (not in n_flagged_reads, nor in n_deferred_reads), no record -- every other read's records are unchanged, and the
    handed-back reads mapped as a batch of their own give exactly the records they have in the one-pass batch."""
    from lra_amd import seed, mapread
    genome = synth.make_genome(600_000, seed=12, repeat_frac=0.3, n_families=3)
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    reads, truth = synth.simulate_reads(genome, 24, 9000, 4000, 0.10, seed=21)
    rng = np.random.default_rng(8)
    reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2000)].copy())      # junk: unaligned in both runs
    raw = [r.tobytes() for r in reads]
    names = [b"q%d" % i for i in range(len(reads))]
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], o)
    batch = seed.ReadBatch(ctx, raw)
    s = seed.fetch(ctx, seed.seed_batch(ctx, batch, o.globalK, o.globalW, o.globalMaxFreq))
    per_read = np.diff(s["match_off"]).astype(np.int64)
    T = int(np.sort(per_read)[len(per_read) * 2 // 3])                    # a third of the reads lie above it
    res0 = mapper.align(batch)
    assert mapper.stats["n_handed_back_reads"] == 0 and mapper.stats["n_deferred_reads"] == 0
    texts0 = mapper.records(res0, names, raw)
    mapper.copts.defer_seed_matches = T
    res1 = mapper.align(batch)
    st1 = ctx.to_host(res1.d_read_status, len(reads), np.uint32)
    D = np.nonzero(st1 & 64)[0]
    assert np.array_equal(D, np.nonzero(per_read > T)[0]) and 0 < len(D) < len(reads) - 1
    assert np.all(st1[D] == 64) and np.all(np.delete(st1, D) == 0)
    assert mapper.stats["n_handed_back_reads"] == len(D) and mapper.stats["n_deferred_reads"] == 0 and int(res1.counters.n_flagged_reads) == 0
    texts1 = mapper.records(res1, names, raw)
    for r in range(len(reads)):
        assert texts1[r] == (b"" if r in set(D.tolist()) else texts0[r]), r
    mapper.copts.flagged_unaligned = 1                                      # (a handed-back read is not written as an unaligned one either)
    assert mapper.records(res1, names, raw)[int(D[0])] == b""
    mapper.copts.flagged_unaligned = 0
    mapper.copts.defer_seed_matches = 0
    res2 = mapper.align(seed.ReadBatch(ctx, [raw[i] for i in D]))
    assert mapper.stats["n_handed_back_reads"] == 0 and mapper.stats["n_deferred_reads"] == 0
    texts2 = mapper.records(res2, [names[i] for i in D], [raw[i] for i in D])
    assert [texts2[k] for k in range(len(D))] == [texts0[i] for i in D]
    assert sum(1 for t in texts2 if t and not (int(t.split(b"\t")[1]) & 4)) >= len(D) - 1


@pytest.mark.gpu
def test_a_seed_result_made_ahead_on_a_side_context_gives_the_same_alignments(ctx):
    """lra_seed_prefetch + lra_ctx_adopt_seed (scheduling only): a1-a4 of a batch run on a side context that borrows the reference data, the mapping context adopts the
    result, and its batch call on the same reads starts from it -- same records as the call that seeds itself; the adopted result is used once; a call on other
    reads ignores it and seeds as usual; adopting twice without a prefetch in between is refused; defer_seed_matches does not combine with it."""
    import threading
    from lra_amd import seed, mapread
    from lra_amd.context import Context
    from lra_amd import LraError
    genome = synth.make_genome(500_000, seed=31, repeat_frac=0.3, n_families=3)
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    reads, _ = synth.simulate_reads(genome, 20, 8000, 3000, 0.10, seed=5)
    raw = [r.tobytes() for r in reads]
    names = [b"q%d" % i for i in range(len(reads))]
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], o)
    batch = seed.ReadBatch(ctx, raw)
    other = seed.ReadBatch(ctx, raw[:7])
    texts0 = mapper.records(mapper.align(batch), names, raw)
    texts_other = mapper.records(mapper.align(other), names[:7], raw[:7])
    side = Context(0)
    mapread.LowAccMapper.sharing(side, mapper)
    err = []

    def ahead():                                                           # (a host thread of its own, as a driver would run it beside the previous batch)
        try:
            seed.seed_prefetch(side, batch, o.globalK, o.globalW, o.globalMaxFreq)
        except BaseException as e:
            err.append(e)
    t = threading.Thread(target=ahead); t.start()
    assert mapper.records(mapper.align(other), names[:7], raw[:7]) == texts_other      # the mapping context is busy with another batch meanwhile
    t.join()
    assert not err
    seed.adopt_seed(ctx, side)
    with pytest.raises(LraError):
        seed.adopt_seed(ctx, side)                                         # nothing prefetched since
    ctx.timing(True); ctx.timing_reset()
    assert mapper.records(mapper.align(batch), names, raw) == texts0      # from the adopted result: the mapping context runs no sketch of its own
    assert ctx.timing_get("sketch_emit")[1] == 0
    assert mapper.records(mapper.align(batch), names, raw) == texts0      # (used once: this call seeds itself)
    assert ctx.timing_get("sketch_emit")[1] == 1
    ctx.timing(False)
    seed.seed_prefetch(side, batch, o.globalK, o.globalW, o.globalMaxFreq)
    seed.adopt_seed(ctx, side)
    assert mapper.records(mapper.align(other), names[:7], raw[:7]) == texts_other      # other reads: the adopted result is dropped
    assert mapper.records(mapper.align(batch), names, raw) == texts0
    seed.seed_prefetch(side, batch, o.globalK, o.globalW, o.globalMaxFreq)
    seed.adopt_seed(ctx, side)
    mapper.copts.defer_seed_matches = 100
    with pytest.raises(LraError):
        mapper.align(batch)
    mapper.copts.defer_seed_matches = 0
    assert mapper.records(mapper.align(batch), names, raw) == texts0
    # a stage call on the mapping context between the adoption and the batch call overwrites the adopted buffers: the batch call then seeds itself
    seed.seed_prefetch(side, batch, o.globalK, o.globalW, o.globalMaxFreq)
    seed.adopt_seed(ctx, side)
    seed.seed_batch(ctx, other, o.globalK, o.globalW, o.globalMaxFreq)
    ctx.timing(True); ctx.timing_reset()
    assert mapper.records(mapper.align(batch), names, raw) == texts0
    assert ctx.timing_get("sketch_emit")[1] == 1
    ctx.timing(False)
    # a context that does not share this one's reference data is refused
    stranger = Context(0)
    g2 = synth.make_genome(200_000, seed=77)
    ik2, ip2 = synth.build_global_index(g2, o.globalK, o.globalW, 100)
    mapread.LowAccMapper(stranger, g2, ik2, ip2, [b"chrS"], [0, len(g2)], o)
    seed.seed_prefetch(stranger, seed.ReadBatch(stranger, raw[:3]), o.globalK, o.globalW, o.globalMaxFreq)
    with pytest.raises(LraError):
        seed.adopt_seed(ctx, stranger)
    stranger.close()
    side.close()


@pytest.mark.gpu
def test_two_stage_batches_give_the_same_records_with_the_front_half_of_the_next_batch_beside_the_back_half(ctx):
    """lra_map_reads_lowacc_front / _back / lra_map_back_release (scheduling only): three batches through the two halves -- the front half of batch i + 1 on one host
    thread while the back half of batch i runs on another -- give the records, counters and status words lra_map_reads_lowacc_batch gives for each of them; an empty
    batch passes through; the handback options are refused."""
    import threading
    from lra_amd import seed, mapread
    from lra_amd import LraError
    genome = synth.make_genome(700_000, seed=41, repeat_frac=0.3, n_families=3)
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], o)
    batches = []
    for b in range(3):
        reads, _ = synth.simulate_reads(genome, 14 + 3 * b, 7000, 3000, 0.10, seed=50 + b)
        if b == 1:
            reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(3).integers(0, 4, 1500)].copy())      # an unalignable read
        raw = [r.tobytes() for r in reads]
        batches.append((seed.ReadBatch(ctx, raw), [b"b%d_%d" % (b, i) for i in range(len(raw))], raw))
    want = []
    for rb, names, raw in batches:
        res = mapper.align(rb)
        want.append((mapper.records(res, names, raw), dict(mapper.stats), ctx.to_host(res.d_read_status, rb.n, np.uint32)))
    got, err = [None] * len(batches), []

    def fronts():
        try:
            for rb, _, _ in batches:
                mapper.front(rb)
        except BaseException as e:
            err.append(e)

    def backs():
        try:
            for i, (rb, names, raw) in enumerate(batches):
                res, bctx = mapper.back()
                mb = mapper.on(bctx)
                got[i] = (mb.records(res, names, raw), dict(mapper.stats), bctx.to_host(res.d_read_status, rb.n, np.uint32))
                mapper.release()
        except BaseException as e:
            err.append(e)
    tf, tb = threading.Thread(target=fronts), threading.Thread(target=backs)
    tf.start(); tb.start(); tf.join(); tb.join()
    assert not err, err
    for i in range(len(batches)):
        assert got[i][0] == want[i][0], i
        assert np.array_equal(got[i][2], want[i][2])
        for k in ("n_alignments", "n_blocks", "n_cigar_runs", "n_mm", "n_match", "n_sdp_anchors", "n_sdp2_anchors", "n_refined_after_btwn", "n_segments", "n_cells", "n_flagged_reads"):
            assert got[i][1][k] == want[i][1][k], (i, k)
    # the one-call entry point still works on the same context afterwards, and an empty batch goes through the two halves
    rb, names, raw = batches[0]
    assert mapper.records(mapper.align(rb), names, raw) == want[0][0]
    empty = seed.ReadBatch(ctx, [])
    mapper.front(empty)
    res, bctx = mapper.back()
    assert int(res.n_reads) == 0 and int(res.n_alignments) == 0
    mapper.release()
    with pytest.raises(LraError):
        mapper.release()                                                  # nothing held
    mapper.copts.defer_seed_matches = 50
    with pytest.raises(LraError):
        mapper.front(rb)
    mapper.copts.defer_seed_matches = 0
    # a front call that fails still hands over a batch, an error batch: the back call for it returns the front's code at once, holds nothing (no release), and the
    # next batch goes through -- the thread of the back halves is never left waiting (one back call per front call)
    with pytest.raises(LraError, match="front half of this batch failed"):
        mapper.back()
    with pytest.raises(LraError):
        mapper.release()                                                  # an error batch holds nothing
    got2, err2, failed = [None] * 3, [], []

    def fronts2():
        for i, (rb_, _, _) in enumerate(batches):
            mapper.copts.defer_seed_matches = 50 if i == 1 else 0         # (batch 1's front half is refused)
            try:
                mapper.front(rb_)
            except LraError as e:
                failed.append((i, str(e)))
            except BaseException as e:
                err2.append(e)
        mapper.copts.defer_seed_matches = 0

    def backs2():
        for i, (rb_, names_, raw_) in enumerate(batches):
            try:
                res_, bctx_ = mapper.back()
            except LraError as e:
                got2[i] = str(e)
                continue
            except BaseException as e:
                err2.append(e); return
            got2[i] = mapper.on(bctx_).records(res_, names_, raw_)
            mapper.release()
    tf, tb = threading.Thread(target=fronts2), threading.Thread(target=backs2)
    tf.start(); tb.start(); tf.join(60); tb.join(60)
    assert not tf.is_alive() and not tb.is_alive(), "a failed front half left a thread waiting"
    assert not err2, err2
    assert [i for i, _ in failed] == [1]
    assert got2[0] == want[0][0] and got2[2] == want[2][0] and isinstance(got2[1], str) and "front half of this batch failed" in got2[1]


@pytest.mark.gpu
def test_release_buffers_keeps_the_reference(ctx):
    """lra_ctx_release_buffers (ABI 9): the work buffers go back to the device, the reference stays loaded, the next batch maps as before."""
    import torch
    from lra_amd import seed, mapread
    genome = synth.make_genome(300_000, seed=77, repeat_frac=0.2, n_families=2)
    ik, ip = synth.build_global_index(genome, 17, 10, 100)
    reads, _ = synth.simulate_reads(genome, 24, 9000, 2000, 0.10, (30, 35, 35), seed=5)
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chr1"], [0, len(genome)], mapread.with_gli(mapread.LowAccOptions()))
    batch = seed.ReadBatch(ctx, [r.tobytes() for r in reads])
    a = mapper.fetch(mapper.align(batch))
    free0 = torch.cuda.mem_get_info()[0]
    freed = ctx.release_buffers()
    assert freed > 0 and torch.cuda.mem_get_info()[0] >= free0 + freed // 2
    b = mapper.fetch(mapper.align(batch))
    for k in ("job_aln_off", "job_reached", "read_status", "strand", "chrom", "block_off", "blocks", "counts"):
        assert np.array_equal(a[k], b[k]), k
    assert ctx.release_buffers() > 0
    # the same between two-stage batches: the companions' buffers (the back context's, the handover sets') go too; refused while a batch sits between the halves
    from lra_amd import LraError
    names = [b"r%d" % i for i in range(len(reads))]
    raw = [r.tobytes() for r in reads]
    want = mapper.records(mapper.align(batch), names, raw)
    for rnd in range(2):
        mapper.front(batch)
        with pytest.raises(LraError):
            ctx.release_buffers()                                          # handed over, not taken
        res, bctx = mapper.back()
        with pytest.raises(LraError):
            ctx.release_buffers()                                          # taken, not released
        assert mapper.on(bctx).records(res, names, raw) == want, rnd
        mapper.release()
        assert ctx.release_buffers() > 0


@pytest.mark.gpu
def test_best_chain_without_split_chain_ends_the_read(ctx):
    """Map_lowacc.h:263-267: when SPLITChain + RemoveSpuriousSplitChain leave the FIRST primary chain no split chain the read is unaligned -- the later chains are never
    looked at, even when one of them is the read's true alignment.  Two 19 kb -CLR reads of the bench's own workload that have the case (tests/golden/
    clr_best_chain_spurious_reads.fa; found by tools/locate_mismatch.py: a dozen tier-1 anchors in all, the best chain three of them), against the seeded GRCh38-like
    reference they were simulated from: no chain reaches :574, the device holds no alignment for them, and the record is the unaligned one -- as the oracle's."""
    import os
    import torch
    import oracle_pipeline as OP
    from lra_amd import seed, mapread, index as I, synth_genome as sg
    dev = torch.device("cuda", 0)
    genome, chrom_pos, chrom_names = sg.make_grch38_like(dev, scale=1.0, seed=3)
    c2 = type(ctx)(0)                                                       # (a context of its own: the session's holds other tests' reference)
    try:
        mopts = mapread.with_gli(mapread.clr_options())
        mapper = mapread.LowAccMapper(c2, genome, None, None, chrom_names, chrom_pos, mopts, index_params=(15, 10, 250, 12, 1), staged=False)
        del genome
        fa = open(os.path.join(os.path.dirname(__file__), "golden", "clr_best_chain_spurious_reads.fa")).read().split("\n")
        reads = [fa[i + 1].encode() for i in range(0, len(fa) - 1, 2) if fa[i].startswith(">")]
        assert len(reads) == 2
        res = mapper.align(seed.ReadBatch(c2, reads))
        out = mapper.fetch(res)
        na = int(res.num_aln)
        key, pos = I.global_index(c2)
        g = c2.to_host(c2.lib.lra_ctx_genome_ptr(c2.h), mapper.G, np.uint8).tobytes() + b"\0" * 64
        g_index = mapper.fetch_local_index()
        oo = dict(OP.CLR, localIndexWindow=mopts.localIndexWindow)
        for r, rd in enumerate(reads):
            groups, unaligned = OP.map_read_lowacc(rd, g, key, pos, g_index, oo, chrom_pos=mapper.chrom_pos)
            assert unaligned and sum(len(x) for x in groups) == 0, r        # the oracle: unaligned
            assert out["read_status"][r] == 0 and not out["job_reached"][r * na:(r + 1) * na].any(), r
            assert int(out["job_aln_off"][(r + 1) * na]) == int(out["job_aln_off"][r * na]), r
        assert int(res.n_alignments) == 0
        sam = mapper.records(res, [b"a", b"b"], reads)
        assert all(t.split(b"\t")[1] == b"4" and t.count(b"\n") == 1 for t in sam)
    finally:
        c2.close()
