"""ctypes bindings to oracle/liblra_oracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "liblra_oracle.so")
        build_oracle()  # make is a no-op when up to date
        _LIB = C.CDLL(path)
    return _LIB


def ref_bin(name):
    """Path of a reference-function driver under oracle/_ref (None if not built)."""
    p = os.path.join(ORACLE_DIR, "_ref", name)
    return p if os.path.exists(p) else None


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def affine_one_gap_align(q: bytes, t: bytes, m, mm, indel, k, cap=4096):
    L = lib()
    blocks = np.zeros(3 * cap, dtype=np.int32)
    nb = C.c_int(0)
    st = C.c_int(0)
    L.oracle_affine_one_gap_align.restype = C.c_int
    score = L.oracle_affine_one_gap_align(C.c_char_p(q), len(q), C.c_char_p(t), len(t), m, mm, indel, k,
                                          _p(blocks, C.c_int), cap, C.byref(nb), C.byref(st))
    n = nb.value
    assert n <= cap
    return score, blocks[:3 * n].reshape(n, 3).copy(), st.value
