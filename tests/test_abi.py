"""The C-ABI library loads on a CPU-only box and exports every symbol include/lra_hip.h declares
(no compute calls here); the ctypes table in lra_amd/_lib.py covers the same set."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lra_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lra_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from lra_amd._lib import library_path, SYMBOLS, load_library
    lib = ctypes.CDLL(library_path())
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(SYMBOLS) == names, (sorted(set(names) ^ set(SYMBOLS)))
    L = load_library()
    assert L.lra_abi_version() == 3


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lra_amd.context import Context
    from lra_amd._lib import LraError
    with pytest.raises(LraError):
        Context(0)


def test_product_does_not_reference_the_oracle():
    """Nothing under lra_amd/ may import, link or execute oracle/ (it is test infrastructure)."""
    for d, _, fs in os.walk(os.path.join(ROOT, "lra_amd")):
        if "build" in d.split(os.sep):
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liblra_oracle" not in txt and "oracle/" not in txt, os.path.join(d, f)
