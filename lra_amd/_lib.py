"""ctypes loader for liblra_hip.so (the HIP kernels + C ABI).  Fails loudly if absent."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ABI_VERSION = 9


class LraError(RuntimeError):
    pass


def library_path():
    return os.path.join(HERE, "liblra_hip.so")


# name -> (restype, argtypes); every symbol include/lra_hip.h declares
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p
SYMBOLS = {
    "lra_abi_version": (C.c_int, []),
    "lra_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lra_ctx_destroy": (None, [_vp]),
    "lra_ctx_release_buffers": (C.c_int, [_vp, _u64p]),
    "lra_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "lra_ctx_last_error": (C.c_char_p, [_vp]),
    "lra_copy_to_host": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "lra_copy_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "lra_ctx_timing_enable": (C.c_int, [_vp, C.c_int]),
    "lra_ctx_timing_reset": (C.c_int, [_vp]),
    "lra_ctx_timing_get": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "lra_ctx_load_genome": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lra_ctx_load_global_index": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "lra_ctx_load_genome_device": (C.c_int, [_vp, _vp, C.c_uint64]),
    "lra_ctx_build_global_index": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "lra_ctx_global_index": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lra_write_mms": (C.c_int, [C.c_char_p, C.c_int, _vp, _vp, C.c_int, _vp, _vp, C.c_uint64]),
    "lra_read_mms": (C.c_int, [C.c_char_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lra_write_gli": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, _vp, _vp, _vp]),
    "lra_read_gli": (C.c_int, [C.c_char_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lra_create_rc_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "lra_sort_minimizers_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "lra_sort_pairs_batch": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "lra_seed_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_map_reads_lowacc_front": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, _vp]),
    "lra_map_reads_lowacc_back": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lra_map_back_release": (C.c_int, [_vp]),
    "lra_seed_prefetch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "lra_ctx_adopt_seed": (C.c_int, [_vp, _vp]),
    "lra_clean_matches_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "lra_match_rate_batch": (C.c_int, [_vp, _vp, C.c_float, _vp]),
    "lra_fine_clusters_batch": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "lra_refine_btwn_clusters_batch": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp, _vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp,
                                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_linear_extend_clusters_batch": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_linear_extend_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "lra_sparse_dp_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lra_reads_open": (C.c_int, [_vp, C.c_int, _vp]),
    "lra_reads_next_batch": (C.c_int, [_vp, C.c_uint64, _vp]),
    "lra_reads_close": (None, [_vp]),
    "lra_reads_last_error": (C.c_char_p, [_vp]),
    "lra_host_thread_budget": (C.c_int, []),
    "lra_map_host_trim": (C.c_uint64, [C.c_uint64]),
    "lra_map_reads_host": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "lra_global_chain_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lra_split_chains_highacc_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "lra_split_chains_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_format_sam": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_char_p, _vp, C.c_uint64, _vp]),
    "lra_format_sam_simple": (C.c_int, [_vp, C.c_int, C.c_char_p, _vp, C.c_uint64, _vp]),
    "lra_format_paf": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64, _vp]),
    "lra_format_bed": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "lra_refine_space_batch": (C.c_int, [_vp, C.c_int] + [_vp] * 13 + [C.c_int] * 4 + [_vp]),
    "lra_refine_space_batch_mf": (C.c_int, [_vp, C.c_int] + [_vp] * 13 + [C.c_int] * 3 + [_vp, _vp]),
    "lra_between_anchors_batch": (C.c_int, [_vp, C.c_int] + [_vp] * 8 + [C.c_int] * 5 + [_vp]),
    "lra_refine_breakpoint_batch": (C.c_int, [_vp, C.c_int] + [_vp] * 16),
    "lra_sparse_dp_boxes_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lra_split_clusters_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "lra_refine_splitchain_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    "lra_refine_btwn_splitchain_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_int, _vp, _vp]),
    "lra_trim_anchor_pairs_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp]),
    "lra_merge_extend_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "lra_format_sam_header": (C.c_int, [C.c_char_p, C.c_char_p, _vp, _vp, C.c_int, _vp, C.c_uint64, _vp]),
    "lra_alignment_strings": (C.c_int, [C.c_char_p, C.c_char_p, _vp, C.c_int, _vp, _vp, _vp, C.c_uint64, _vp, _vp]),
    "lra_md_string": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, _vp, C.c_uint64, _vp]),
    "lra_format_pairwise": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, _vp,
                                      C.c_uint64, _vp]),
    "lra_group_alignments": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "lra_order_alignments": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int]),
    "lra_simple_mapqv": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lra_output_read": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_char, C.c_int, C.c_char_p, C.c_int, _vp, _vp, C.c_uint64, _vp]),
    "lra_refine_clusters_batch": (C.c_int, [_vp, C.c_int] + [_vp] * 10 + [C.c_uint64, _vp, _vp, C.c_int, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    "lra_local_refine_highacc_batch": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, C.c_uint64,
                                                 _vp, _vp, C.c_int, _vp, _vp]),
    "lra_local_refine_batch": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp,
                                         C.c_int, _vp, _vp]),
    "lra_local_refine_inputs_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "lra_trim_overlapped_anchors_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "lra_switch_to_original_anchors_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp]),
    "lra_refine_btwn_space_batch": (C.c_int, [_vp, C.c_int] + [_vp] * 12 + [C.c_uint64, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_merge_same_diag_batch": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "lra_switchindex_batch": (C.c_int, [_vp, C.c_uint64] + [_vp] * 9 + [C.c_uint64, _vp]),
    "lra_map_opts_preset_ont": (None, [_vp]),
    "lra_map_opts_preset_ccs": (None, [_vp]),
    "lra_map_opts_preset_contig": (None, [_vp]),
    "lra_map_opts_preset_clr": (None, [_vp]),
    "lra_ctx_load_chromosomes": (C.c_int, [_vp, _vp, C.c_int]),
    "lra_ctx_build_local_index": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lra_map_opts_apply_local_index": (None, [_vp, C.c_int, C.c_int, C.c_int]),
    "lra_ctx_local_index_params": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lra_ctx_load_local_index": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_uint64, _vp, _vp, C.c_uint64, _vp]),
    "lra_map_reads_lowacc_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, _vp, _vp]),
    "lra_map_reads_highacc_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, _vp, _vp]),
    "lra_ctx_genome_ptr": (_vp, [_vp]),
    "lra_ctx_share_reference": (C.c_int, [_vp, _vp]),
    "lra_ctx_local_index": (C.c_int, [_vp, _vp, _vp]),
    "lra_map_snapshot": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "lra_map_records_host": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_char_p, C.c_int, _vp, _vp, _vp]),
    "lra_map_host_free": (None, [_vp]),
    "lra_map_host_flagged": (C.c_uint64, [_vp, _vp]),
    "lra_map_pack": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "lra_map_unpack_host": (C.c_int, [_vp, C.c_uint64, _vp]),
    "lra_map_records": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_char_p, _vp, C.c_uint64, _vp, _vp]),
    "lra_filter_chains_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "lra_filter_chains_ex_batch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "lra_calculate_statistics_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "lra_local_index_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_local_index_masked_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "lra_local_compare_batch": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "lra_indel_refine_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, _vp]),
    "lra_affine_one_gap_align_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                                  C.c_int, _vp, _vp, _vp, _vp, _vp]),
}


def load_library():
    """Load liblra_hip.so; raise LraError if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch (device memory / streams plumbing) bundles its own HIP runtime; it must be the one
    # already mapped when liblra_hip.so resolves libamdhip64, so both share one runtime
    # instance (device pointers and streams cross the boundary).
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise LraError("liblra_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "-- lra_amd has no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.lra_abi_version() != ABI_VERSION:
        raise LraError("liblra_hip.so ABI %d != expected %d" % (lib.lra_abi_version(), ABI_VERSION))
    _LIB = lib
    return lib
