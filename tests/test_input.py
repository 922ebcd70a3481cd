"""The input side (Input.h:87-168, :182-283, :405-421; lra_amd/csrc/input.hip): FASTA / FASTQ into batches, and the host-buffer boundary.  The expected
batches come from a restatement of the reference's stream logic in plain Python (below: an istream with its eof bit); PARITY UNPINNED (Input.h includes htslib)."""
import numpy as np
import pytest

from lra_amd import synth

EOF_ = -1


class IStream:
    """the parts of std::istream the reader uses: getline, peek, get, the eof bit"""
    def __init__(self, data: bytes):
        self.d, self.p, self.eof = data, 0, False

    def getline(self):
        if self.p >= len(self.d):
            self.eof = True
            return b""
        i = self.d.find(b"\n", self.p)
        if i < 0:
            line, self.p, self.eof = self.d[self.p:], len(self.d), True
        else:
            line, self.p = self.d[self.p:i], i + 1
        return line

    def peek(self):
        if self.p >= len(self.d):
            self.eof = True
            return EOF_
        return self.d[self.p]

    def get(self):
        c = self.peek()
        if c != EOF_:
            self.p += 1
        return c


def _token(header: bytes) -> bytes:                                        # nameStrm >> c >> read.name
    parts = header.lstrip()[1:].split()
    return parts[0] if parts else b""


def ref_batches(files, max_bases):
    """Input::Initialize + GetNext + BufferedRead (FASTA / FASTQ) -> list of batches of (name, seq, qual or None)"""
    datas = [open(f, "rb").read() for f in files]
    cur = [0]

    def initialize():
        s = IStream(datas[cur[0]])
        if s.peek() == ord(">"):
            return s, 0
        if s.peek() == ord("@"):
            t = IStream(datas[cur[0]]); t.getline(); t.getline()
            if t.peek() == ord("+"):
                return s, 1
        return s, -1
    st = initialize()
    state = {"s": st[0], "type": st[1], "ok": st[1] >= 0}

    def get_next():
        if not state["ok"]:
            return None
        s = state["s"]
        if state["type"] == 0 and s.eof:
            cur[0] += 1
            if cur[0] >= len(files):
                state["ok"] = False; return None
            state["s"], state["type"] = initialize(); s = state["s"]
            if state["type"] < 0:
                state["ok"] = False; return None
        if s.eof:
            return None
        if state["type"] == 0:
            name = _token(s.getline())
            seq = b""
            c = s.peek()
            while c != EOF_ and c != ord(">"):
                seq += s.getline().replace(b" ", b"").upper()
                c = s.peek()
            if c == EOF_:
                s.get()
            return name, seq, None
        h, q, sep, ql = s.getline(), s.getline(), s.getline(), s.getline()
        if not (h and q and sep and ql):
            cur[0] += 1
            if cur[0] >= len(files):
                state["ok"] = False; return None
            state["s"], state["type"] = initialize(); s = state["s"]
            if state["type"] < 0:
                state["ok"] = False; return None
            if state["type"] == 1:
                h, q, sep, ql = s.getline(), s.getline(), s.getline(), s.getline()
        if not (h and q and sep and ql):
            return None
        q2, ql2 = q.replace(b" ", b"").upper(), ql.replace(b" ", b"")
        if len(q2) != len(ql2):                       # the reference asserts (Input.h:287) and, with asserts off, writes past its buffer (:288-290): the reader ends the input here
            state["ok"] = False; return None
        return _token(h), q2, ql2
    out = []
    while True:
        batch, total = [], 0
        while total < max_bases:
            r = get_next()
            if r is None:
                break
            batch.append(r); total += len(r[1])
        if not batch:
            return out
        out.append(batch)


def _write_files(tmp_path):
    rng = np.random.default_rng(5)
    base = lambda n: bytes(np.frombuffer(b"ACGTacgtN", np.uint8)[rng.integers(0, 9, n)])
    fa1 = tmp_path / "a.fa"; fa2 = tmp_path / "b.fa"; fq1 = tmp_path / "c.fq"; fq2 = tmp_path / "d.fastq"; fq3 = tmp_path / "e.fq"
    with open(fa1, "wb") as f:
        for i in range(7):
            s = base(int(rng.integers(50, 400)))
            f.write(b">read%d some description %d\n" % (i, i))
            for x in range(0, len(s), 60):
                f.write(s[x:x + 60][:30] + (b" " if i == 3 else b"") + s[x:x + 60][30:] + b"\n")
            if i == 2:
                f.write(b"\n")                                             # a blank line inside the record
        f.write(b"> spaced_name tail\nACGTAC GTTT\nacgt")                   # name behind a blank; no newline at the end of the file
    with open(fa2, "wb") as f:
        f.write(b">second_file_1\nAAAACCCCGGGGTTTT\n>second_file_2\n\nACGT\n")
    with open(fq1, "wb") as f:
        for i in range(5):
            s = base(int(rng.integers(30, 200)))
            f.write(b"@fq%d/1 extra\n%s\n+\n%s\n" % (i, s, bytes([33 + (j % 40) for j in range(len(s))])))
    with open(fq2, "wb") as f:
        f.write(b"@last one\nACgTNN\n+last\nIIII II\n")
        f.write(b"\n")                                                     # trailing blank line: the file is over
    with open(fq3, "wb") as f:
        f.write(b"@x\nACGT\n+\nIIII\n@y\nGGGG\n+\nJJJJ\n@short_quality\nACGTACGT\n+\nIIII\n@never_read\nAC\n+\nII\n")   # a quality string shorter than its read ends the input
    return [str(x) for x in (fa1, fa2, fq1, fq2, fq3)]


@pytest.mark.parametrize("order,max_bases", [([0, 1], 500), ([2, 3, 4], 300), ([0, 1], 10 ** 9), ([2, 1], 100), ([1, 2, 4], 1)])
def test_reader_matches_reference_logic(tmp_path, order, max_bases):
    """FASTA and FASTQ files in sequence: wrapped and blank sequence lines, lower case, blanks, descriptions, no trailing newline, a FASTQ file that ends
    with a blank line, a FASTA file behind a FASTQ file (the reference then stops), batches cut by max_bases"""
    from lra_amd import reads_io
    files = _write_files(tmp_path)
    sel = [files[i] for i in order]
    exp = ref_batches(sel, max_bases)
    rf = reads_io.ReadsFile(sel)
    got = []
    failed = None
    while True:
        try:
            b = rf.next_batch(max_bases)
        except IOError as e:                                               # the corrupt record of e.fq: an error, never a silent end of the input
            failed = str(e)
            if e.partial is not None:
                got.append([(n, s, q) for n, s, q in zip(e.partial["names"], e.partial["seqs"], e.partial["quals"])])
            with pytest.raises(IOError):                                   # and it stays one
                rf.next_batch(max_bases)
            break
        if b is None:
            break
        got.append([(n, s, q) for n, s, q in zip(b["names"], b["seqs"], b["quals"])])
    rf.close()
    assert got == exp, (len(got), len(exp))
    assert sum(len(b) for b in got) >= 3
    if 4 in order:
        assert failed and "short_quality" in failed and "e.fq" in failed, failed
    else:
        assert failed is None


@pytest.mark.gpu
def test_map_reads_host_from_fastq(ctx, tmp_path):
    """reads written to a FASTQ file -> lra_reads_next_batch -> lra_map_reads_host -> lra_map_records: the same text as the device-buffer boundary on the same reads"""
    from lra_amd import seed, mapread, reads_io
    genome = synth.make_genome(400_000, seed=9, repeat_frac=0.2, n_families=3)
    CH = [0, 150_000, len(genome)]
    o = mapread.LowAccOptions()
    ik, ip = synth.build_global_index(genome, o.globalK, o.globalW, 100)
    reads, _ = synth.simulate_reads(genome, 24, 6000, 1500, 0.10, seed=4)
    fq = tmp_path / "reads.fq"
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            s = r.tobytes()
            f.write(b"@r%d len=%d\n%s\n+\n%s\n" % (i, len(s), s.lower() if i % 3 == 0 else s, b"I" * len(s)))
    mapper = mapread.LowAccMapper(ctx, genome, ik, ip, [b"chrA", b"chrB"], CH, o)
    names = [b"r%d" % i for i in range(len(reads))]
    ref = mapper.records(mapper.align(seed.ReadBatch(ctx, [r.tobytes() for r in reads])), names, [r.tobytes() for r in reads], quals=[b"I" * len(r) for r in reads])
    rf = reads_io.ReadsFile([str(fq)])
    texts = []
    while True:
        b = rf.next_batch(60_000)
        if b is None:
            break
        assert all(s == s.upper() for s in b["seqs"])
        res = reads_io.map_reads_host(mapper, b["raw"])
        texts += mapper.records(res, b["names"], b["seqs"], quals=b["quals"])
    rf.close()
    assert len(texts) == len(reads) and texts == ref
