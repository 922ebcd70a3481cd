"""Oracle pieces pinned to reference headers that compile in place (no htslib): the k-mer primitives StoreMinimizers is made of (TupleOps.h, SeqUtils.h).
Golden file: tools/make_golden_tuple_ops.py (oracle/ref_harness/tuple_ops_ref.cpp)."""
import ctypes as C
import json
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tuple_ops_golden.json")


def test_oracle_kmer_primitives_match_reference(oracle):
    L = O.lib()
    L.oracle_kmer_stream.restype = C.c_long
    cases = json.load(open(GOLD))["cases"]
    n_other = 0
    for i, c in enumerate(cases):
        s = c["seq"].encode(); k = c["k"]
        out = np.zeros(3 * max(1, len(s)), np.uint64); rc = C.create_string_buffer(len(s) + 1)
        m = L.oracle_kmer_stream(C.c_char_p(s), C.c_long(len(s)), C.c_int(k), out.ctypes.data_as(C.POINTER(C.c_uint64)), rc)
        assert m == len(s) - k + 1, i
        assert out[:3 * m].tolist() == [int(x) for x in c["codes"]], i          # forward code, reverse-complement code, canonical key of every k-mer
        assert rc.raw[:len(s)].decode() == c["rc"], i                           # CreateRC
        n_other += any(ch not in "ACGT" for ch in c["seq"])
    assert len(cases) == 60 and n_other >= 30
    # the minimizer sketch emits keys from this stream: every stored (key, pos) is the canonical key of the k-mer at pos
    for c in cases[:20]:
        s = c["seq"].encode(); k = c["k"]
        if k > 25 or any(ch not in "ACGT" for ch in c["seq"]) or len(s) < k + 9:
            continue
        keys, pos = O.store_minimizers(s, k, 10)
        for key, p in zip(keys.tolist(), pos.tolist()):
            assert key == int(c["codes"][3 * p + 2])
