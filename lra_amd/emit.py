"""Host-side mirror of the record emitters (reference: Alignment.h:591-905) -- plain text formatting, no device work."""
import ctypes as C

from ._lib import load_library


class AlnRecord(C.Structure):
    _fields_ = ([("read_name", C.c_char_p), ("read", C.c_char_p), ("qual", C.c_char_p), ("read_len", C.c_int32), ("chrom", C.c_char_p),
                 ("genome_len", C.c_uint32), ("cigar", C.c_char_p), ("flag", C.c_uint32), ("strand", C.c_int32), ("mapqv", C.c_uint32),
                 ("supplementary", C.c_int32), ("typeofaln", C.c_int32), ("q_start", C.c_uint32), ("q_end", C.c_uint32), ("t_start", C.c_uint32),
                 ("t_end", C.c_uint32), ("pre_clip", C.c_int32), ("suf_clip", C.c_int32)] +
                [(n, C.c_int32) for n in ("nm", "nmm", "nins", "ndel", "tdel", "tins", "nSmallDel", "nMedDel", "nLargeDel", "nSmallIns", "nMedIns", "nLargeIns")] +
                [("value", C.c_float), ("order", C.c_int32), ("NumOfAnchors0", C.c_int32), ("NumOfAnchors1", C.c_int32), ("runtime", C.c_int32),
                 ("n_blocks", C.c_int32), ("first_block_qpos", C.c_uint32), ("last_block_qend", C.c_uint32), ("is_secondary", C.c_int32), ("md", C.c_char_p), ("blocks", C.c_void_p), ("strand_read", C.c_char_p), ("chrom_text", C.c_void_p)])


class AlnGroup(C.Structure):
    _fields_ = [("first", C.c_int32), ("count", C.c_int32), ("q_start", C.c_uint32), ("q_end", C.c_uint32), ("t_start", C.c_uint32), ("t_end", C.c_uint32),
                ("nm", C.c_int32), ("nmm", C.c_int32), ("ndel", C.c_int32), ("nins", C.c_int32), ("is_secondary", C.c_int32), ("value", C.c_float),
                ("NumOfAnchors0", C.c_int32), ("NumOfAnchors1", C.c_int32)]


def finish_read(recs, seg_off, bypass_clustering=True, read_type="ont", globalK=10, print_num_aln=1, fmt="s", hard_clip=False, passthrough=None,
                unaligned=None):
    """SetFromSegAlignment -> AlignmentsOrder::Update -> SimpleMapQV -> OUTPUT (Map_lowacc.h:600-618) for one read.
    recs: list of AlnRecord (modified in place through the returned array), seg_off: CSR of alignments over them.
    Returns (text bytes, records array, groups array, index list)."""
    lib = load_library()
    n = len(seg_off) - 1
    arr = (AlnRecord * max(1, len(recs)))(*recs)
    so = (C.c_int32 * (n + 1))(*seg_off)
    groups = (AlnGroup * max(1, n))()
    index = (C.c_int32 * max(1, n))()
    assert lib.lra_group_alignments(arr, so, n, groups) == 0
    assert lib.lra_order_alignments(groups, n, arr, index, 0) == 0
    assert lib.lra_simple_mapqv(groups, index, n, arr, int(bypass_clustering), int(read_type == "clr"), int(read_type == "ont"), int(globalK)) == 0
    ln = C.c_uint64(0)
    un = C.byref(unaligned) if unaligned is not None else None
    args = (groups, index, n, arr, int(print_num_aln), C.c_char(fmt.encode()), int(hard_clip), passthrough, int(unaligned is not None), un)
    lib.lra_output_read(*args, None, C.c_uint64(0), C.byref(ln))
    buf = C.create_string_buffer(ln.value + 1)
    assert lib.lra_output_read(*args, buf, C.c_uint64(ln.value), C.byref(ln)) == 0
    return buf.raw[:ln.value], arr, groups, list(index)[:n]


def _call(fn, *args):
    lib = load_library()
    n = C.c_uint64(0)
    getattr(lib, fn)(*args, None, C.c_uint64(0), C.byref(n))
    buf = C.create_string_buffer(n.value + 1)
    rc = getattr(lib, fn)(*args, buf, C.c_uint64(n.value), C.byref(n))
    if rc != 0:
        raise RuntimeError("%s failed (%d)" % (fn, rc))
    return buf.raw[:n.value]


def format_sam(group, as_idx, hard_clip=False, passthrough=None):
    arr = (AlnRecord * len(group))(*group)
    return _call("lra_format_sam", arr, len(group), int(as_idx), int(hard_clip), passthrough)


def format_sam_simple(rec, hard_clip=False, passthrough=None):
    return _call("lra_format_sam_simple", C.byref(rec), int(hard_clip), passthrough)


def format_paf(rec, print_cigar=False):
    return _call("lra_format_paf", C.byref(rec), int(print_cigar))


def format_bed(rec):
    return _call("lra_format_bed", C.byref(rec))
