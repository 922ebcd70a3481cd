"""ctypes mirror of the input side (lra_amd/csrc/input.hip): FASTA / FASTQ batches and the host-buffer boundary.  Tests and tools only."""
import ctypes as C

import numpy as np

from ._lib import load_library


class ReadBatchC(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("total_bases", C.c_uint64), ("seq", C.c_void_p), ("off", C.POINTER(C.c_uint64)), ("read_len", C.POINTER(C.c_int32)),
                ("names", C.POINTER(C.c_char_p)), ("reads", C.POINTER(C.c_void_p)), ("quals", C.POINTER(C.c_char_p))]


class ReadsFile:
    def __init__(self, files):
        self.lib = load_library()
        arr = (C.c_char_p * len(files))(*[f.encode() if isinstance(f, str) else f for f in files])
        self.h = C.c_void_p()
        rc = self.lib.lra_reads_open(arr, len(files), C.byref(self.h))
        if rc != 0:
            raise IOError("cannot determine the format of the input reads (%d)" % rc)

    def next_batch(self, max_bases):
        """-> None at the end, else dict(names, seqs, quals (None for FASTA reads), off, raw=(ReadBatchC kept alive until the next call))"""
        b = ReadBatchC()
        rc = self.lib.lra_reads_next_batch(self.h, C.c_uint64(int(max_bases)), C.byref(b))
        n = b.n_reads
        out = None
        if n:
            off = np.ctypeslib.as_array(b.off, shape=(n + 1,)).copy()
            seq = C.string_at(b.seq, int(b.total_bases))
            out = dict(names=[b.names[i] for i in range(n)], seqs=[seq[int(off[i]):int(off[i + 1])] for i in range(n)], quals=[b.quals[i] for i in range(n)], off=off, raw=b)
        if rc != 0:
            e = IOError("lra_reads_next_batch failed (%d): %s" % (rc, (self.lib.lra_reads_last_error(self.h) or b"").decode()))
            e.partial = out                     # the reads in front of the bad record
            raise e
        return out

    def close(self):
        if self.h:
            self.lib.lra_reads_close(self.h)
            self.h = None


def map_reads_host(mapper, raw_batch):
    """lra_map_reads_host on a batch of ReadsFile.next_batch (mapper: LowAccMapper / HighAccMapper) -> MapResult"""
    from .mapread import MapResult
    ctx = mapper.ctx
    res = MapResult()
    ctx.check(ctx.lib.lra_map_reads_host(ctx.h, raw_batch.n_reads, C.c_void_p(raw_batch.seq), raw_batch.off, C.byref(mapper.copts), C.byref(res)))
    return res
