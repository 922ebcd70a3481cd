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
    assert L.lra_abi_version() == 9


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


# Work-buffer slots (lra_ensure) two source files may both name, and why that is safe.  Everything else belongs to ONE file: a buffer
# that outlives its call (results handed back to the caller) must never be another stage's scratch (the sort's scratch sat in
# fine_clusters.hip's result slots).
_SHARED_SLOTS = {
    12: "the sparse DP arena, reused by IndelRefine / CalculateStatistics once the arena is dead (DESIGN 6b, Memory) and by the minimizer sketch's staging arrays before it is alive (seed.hip)",
    **{s: "AffineOneGapAlign glue of between_anchors / refine_breakpoint: per-call scratch, dead at return" for s in (18, 19, 20, 21)},
    **{s: "tier-2 glue: refine_clusters (high-accuracy driver) / refine_splitchain (low-accuracy driver), results of the last call only" for s in (27, 28, 29)},
    **{s: "the two drivers' own buffers: one driver call per context at a time" for s in list(range(57, 66)) + [81, 82, 171, 173, 175]},
}


def test_work_buffer_slots_have_one_owner():
    import collections
    import glob
    owners = collections.defaultdict(set)
    csrc = os.path.join(ROOT, "lra_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        s = open(f).read()
        consts = {k: int(v) for k, v in re.findall(r"\b([A-Z_]+_SLOT)\s*=\s*(\d+)", s)}
        for m in re.finditer(r"\b(?:lra_ensure|up|room(?:<[^>]*>)?|grow_keep|merge_passes)\(\s*[A-Za-z_>\-]+\s*,\s*([0-9A-Z_]+)\s*[,)]", s):
            t = m.group(1)
            if t.isdigit():
                owners[int(t)].add(os.path.basename(f))
            elif t in consts:
                owners[consts[t]].add(os.path.basename(f))
    assert len(owners) > 100
    assert max(owners) < 192                                     # lra_ctx::gbuf
    bad = {k: sorted(v) for k, v in owners.items() if len(v) > 1 and k not in _SHARED_SLOTS}
    assert not bad, bad
    assert owners[97] == {"seed.hip"} and owners[98] == {"seed.hip"} and owners[86] == {"fine_clusters.hip"} and owners[87] == {"fine_clusters.hip"}
