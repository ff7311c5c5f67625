"""ctypes binding of libporechop_amd.so (the C ABI of include/porechop_amd.h).

The library is built in-tree by porechop_amd/csrc/Makefile (``__graft_entry__.build()``);
it is never pip-installed, so the GPU box loads exactly the .so that sits in this directory.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (PC_LIBRARY: another build of the same library, for A/B timing of kernel variants on one GPU box -- tools/ab_build.sh)
LIB_PATH = os.environ.get("PC_LIBRARY") or os.path.join(HERE, "libporechop_amd.so")

# every symbol include/porechop_amd.h declares (tests check the list against the header)
EXPORTS = [
    "adapterAlignment", "freeCString",
    "pc_version", "pc_strerror", "pc_scores_supported", "pc_create", "pc_destroy",
    "pc_set_scores", "pc_set_adapters", "pc_align_batch_host", "pc_scan_device", "pc_sync",
    "pc_format_result", "pc_format_results", "pc_jit_async", "pc_jit_shutdown", "pc_jit_precompile", "pc_jit_stats", "pc_prefilter_max_edits", "pc_prefilter_device", "pc_prefetch", "pc_memo_clear", "pc_memo_stats", "pc_set_timing", "pc_get_timing", "pc_set_length_hint", "pc_set_int16_only", "pc_copy_windows", "pc_trace_ops_x100", "pc_phase_b_reduce", "pc_phase_b_reduce_masked", "pc_phase_b_select", "pc_phase_b_gather", "pc_phase_b_scatter", "pc_gather_records", "pc_debug_value_range",
    "pc_readset_load", "pc_readset_free", "pc_readset_error", "pc_readset_count", "pc_readset_is_fastq",
    "pc_readset_arena", "pc_readset_offsets", "pc_readset_lengths", "pc_readset_name", "pc_readset_quals",
    "pc_readset_is_rna", "pc_readset_load_many", "pc_readset_file_index", "pc_readset_write",
    "pc_readset_load_segment", "pc_readset_write_at", "pc_io_set_thread_limit", "pc_pack_reads", "pc_unpack_device", "pc_fastq_find_record", "pc_readset_write_sizes", "pc_readset_write_shared",
    "pc_readset_compress", "pc_gzimage_sizes", "pc_gzimage_write", "pc_gzimage_free", "pc_gz_finish", "pc_gzip_file",
    "pc_gzstream_open", "pc_gzstream_next", "pc_gzstream_close", "pc_gz_member_start", "pc_gzstream_open_range", "pc_prefilter_packed", "pc_unpack_windows", "pc_prefilter_defer_count", "pc_prefilter_overflowed", "pc_trim_windows", "pc_middle_hits", "pc_group_survivors", "pc_round_consume",
    "pc_gz_sized_size", "pc_gz_sized_find_record", "pc_readset_load_gz_range",
]


class LibraryMissing(RuntimeError):
    pass


_lib = None


def load_library():
    """Load (once) and type the shared library.  No fallback: a missing build is an error."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, same SONAME as the
    # system one).  Whichever copy is loaded first is the one a later NEEDED "libamdhip64.so.7"
    # binds to, but torch itself asks for it by file name and would load a SECOND runtime if ours
    # came first -- two runtimes in one process share neither streams nor devices.  So when torch
    # is available it goes first; without torch (plain Porechop drop-in) the system runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch-less deployments
        pass
    if not os.path.isfile(LIB_PATH):
        raise LibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C porechop_amd/csrc`). porechop_amd has no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_i64, c_vp, c_cp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p
    L.adapterAlignment.argtypes = [c_cp, c_cp, c_int, c_int, c_int, c_int]
    L.adapterAlignment.restype = c_vp
    L.freeCString.argtypes = [c_vp]
    L.freeCString.restype = None
    L.pc_version.restype = c_cp
    L.pc_strerror.argtypes = [c_int]
    L.pc_strerror.restype = c_cp
    L.pc_scores_supported.argtypes = [c_int] * 5
    L.pc_scores_supported.restype = c_int
    L.pc_create.argtypes = [ctypes.POINTER(c_vp), c_int]
    L.pc_create.restype = c_int
    L.pc_destroy.argtypes = [c_vp]
    L.pc_destroy.restype = None
    L.pc_set_scores.argtypes = [c_vp, c_int, c_int, c_int, c_int]
    L.pc_set_scores.restype = c_int
    L.pc_set_adapters.argtypes = [c_vp, ctypes.POINTER(c_cp), c_int]
    L.pc_set_adapters.restype = c_int
    L.pc_align_batch_host.argtypes = [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]
    L.pc_align_batch_host.restype = c_int
    L.pc_scan_device.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]
    L.pc_scan_device.restype = c_int
    L.pc_sync.argtypes = [c_vp, c_vp]
    L.pc_sync.restype = c_int
    L.pc_copy_windows.argtypes = [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, c_int, c_vp]
    L.pc_copy_windows.restype = c_int
    L.pc_set_length_hint.argtypes = [c_vp, c_int]
    L.pc_set_length_hint.restype = c_int
    L.pc_set_int16_only.argtypes = [c_vp, c_int]
    L.pc_set_int16_only.restype = c_int
    L.pc_set_timing.argtypes = [c_vp, c_int]
    L.pc_set_timing.restype = c_int
    L.pc_get_timing.argtypes = [c_vp, c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    L.pc_get_timing.restype = c_int
    L.pc_debug_value_range.argtypes = [c_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.pc_debug_value_range.restype = c_int
    L.pc_trace_ops_x100.argtypes = [c_vp]
    L.pc_trace_ops_x100.restype = c_int
    L.pc_phase_b_reduce.argtypes = [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_int, c_int, ctypes.c_double, c_vp, c_vp,
                                    c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double, c_int, c_vp, c_vp]
    L.pc_phase_b_reduce.restype = c_int
    L.pc_phase_b_reduce_masked.argtypes = [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_int, c_int, ctypes.c_double, c_vp, c_vp,
                                           c_int, c_vp, c_vp, ctypes.c_double, ctypes.c_double, c_int, c_vp, c_vp, c_vp]
    L.pc_phase_b_reduce_masked.restype = c_int
    c_dbl = ctypes.c_double
    L.pc_phase_b_select.argtypes = [c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_dbl,
                                    c_int, c_dbl, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.pc_phase_b_select.restype = c_int
    L.pc_phase_b_gather.argtypes = [c_vp, c_vp, c_i64, c_int] + [c_vp] * 14
    L.pc_phase_b_gather.restype = c_int
    L.pc_phase_b_scatter.argtypes = [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]
    L.pc_phase_b_scatter.restype = c_int
    L.pc_gather_records.argtypes = [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]
    L.pc_gather_records.restype = c_int
    L.pc_jit_async.argtypes = [c_int]
    L.pc_jit_async.restype = None
    L.pc_jit_shutdown.argtypes = []
    L.pc_jit_shutdown.restype = None
    L.pc_jit_precompile.argtypes = [c_cp, c_cp, c_int, c_int, c_int, c_int, c_cp]
    L.pc_jit_precompile.restype = c_int
    L.pc_jit_stats.argtypes = [ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    L.pc_jit_stats.restype = None
    L.pc_prefilter_max_edits.argtypes = [c_int, ctypes.c_double]
    L.pc_prefilter_max_edits.restype = c_int
    L.pc_prefilter_device.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp]
    L.pc_prefilter_device.restype = c_int
    import atexit
    atexit.register(L.pc_jit_shutdown)       # no worker thread inside hiprtc while the process is torn down
    L.pc_format_result.argtypes = [c_vp, c_cp, ctypes.c_size_t]
    L.pc_format_result.restype = c_int
    L.pc_format_results.argtypes = [c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(c_i64)]
    L.pc_format_results.restype = c_int
    L.pc_prefetch.argtypes = [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(c_cp), c_vp, c_i64, c_int, c_int, c_int, c_int]
    L.pc_prefetch.restype = c_int
    L.pc_readset_load.argtypes = [c_cp, ctypes.POINTER(c_vp)]
    L.pc_readset_load.restype = c_int
    L.pc_readset_free.argtypes = [c_vp]
    L.pc_readset_free.restype = None
    L.pc_readset_error.argtypes = [c_vp]
    L.pc_readset_error.restype = c_cp
    L.pc_readset_count.argtypes = [c_vp]
    L.pc_readset_count.restype = c_i64
    L.pc_readset_is_fastq.argtypes = [c_vp]
    L.pc_readset_is_fastq.restype = c_int
    L.pc_readset_arena.argtypes = [c_vp, ctypes.POINTER(c_i64)]
    L.pc_readset_arena.restype = c_vp
    L.pc_readset_offsets.argtypes = [c_vp]
    L.pc_readset_offsets.restype = c_vp
    L.pc_readset_lengths.argtypes = [c_vp]
    L.pc_readset_lengths.restype = c_vp
    L.pc_readset_name.argtypes = [c_vp, c_i64]
    L.pc_readset_name.restype = c_cp
    L.pc_readset_quals.argtypes = [c_vp, c_i64]
    L.pc_readset_quals.restype = c_cp
    L.pc_readset_is_rna.argtypes = [c_vp, c_i64]
    L.pc_readset_is_rna.restype = c_int
    L.pc_readset_load_many.argtypes = [ctypes.POINTER(c_cp), c_int, ctypes.POINTER(c_vp)]
    L.pc_readset_load_many.restype = c_int
    L.pc_readset_file_index.argtypes = [c_vp]
    L.pc_readset_file_index.restype = c_vp
    L.pc_readset_write.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, ctypes.POINTER(c_cp), c_int,
                                   ctypes.POINTER(c_i64)]
    L.pc_readset_write.restype = c_int
    L.pc_readset_load_segment.argtypes = [c_cp, c_i64, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_vp)]
    L.pc_readset_load_segment.restype = c_int
    L.pc_readset_write_at.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, ctypes.POINTER(c_cp), c_int,
                                      ctypes.POINTER(c_i64)]
    L.pc_readset_write_at.restype = c_int
    L.pc_fastq_find_record.argtypes = [c_cp, c_i64, ctypes.POINTER(c_i64)]
    L.pc_fastq_find_record.restype = c_int
    L.pc_readset_write_sizes.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]
    L.pc_readset_write_sizes.restype = c_int
    L.pc_readset_write_shared.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, ctypes.POINTER(c_cp), c_int,
                                          ctypes.POINTER(c_i64)]
    L.pc_readset_write_shared.restype = c_int
    L.pc_io_set_thread_limit.argtypes = [c_int]
    L.pc_io_set_thread_limit.restype = None
    L.pc_prefilter_packed.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp]
    L.pc_prefilter_packed.restype = c_int
    L.pc_prefilter_defer_count.argtypes = [c_vp, c_int]
    L.pc_prefilter_defer_count.restype = c_int
    L.pc_prefilter_overflowed.argtypes = [c_vp]
    L.pc_prefilter_overflowed.restype = c_int
    L.pc_trim_windows.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]
    L.pc_trim_windows.restype = c_int
    L.pc_middle_hits.argtypes = [c_vp, c_vp, c_i64, ctypes.c_double, c_vp, c_vp, c_vp]
    L.pc_middle_hits.restype = c_int
    L.pc_group_survivors.argtypes = [c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_vp, c_vp]
    L.pc_group_survivors.restype = c_int
    L.pc_round_consume.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_i64, ctypes.c_double, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.pc_round_consume.restype = c_int
    L.pc_unpack_windows.argtypes = [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_int, c_vp]
    L.pc_unpack_windows.restype = c_int
    L.pc_readset_compress.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, ctypes.POINTER(c_vp)]
    L.pc_readset_compress.restype = c_int
    L.pc_gzimage_sizes.argtypes = [c_vp, c_int, c_vp, c_vp]
    L.pc_gzimage_sizes.restype = c_int
    L.pc_gzimage_write.argtypes = [c_vp, c_int, c_vp, c_vp, c_int]
    L.pc_gzimage_write.restype = c_int
    L.pc_gzimage_free.argtypes = [c_vp]
    L.pc_gzimage_free.restype = None
    L.pc_gz_finish.argtypes = [c_cp]
    L.pc_gz_finish.restype = c_int
    L.pc_gzip_file.argtypes = [c_cp, c_cp, c_int, c_int]
    L.pc_gzip_file.restype = c_int
    L.pc_gzstream_open.argtypes = [c_cp, ctypes.POINTER(c_vp)]
    L.pc_gzstream_open.restype = c_int
    L.pc_gzstream_open_range.argtypes = [c_cp, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(c_vp)]
    L.pc_gzstream_open_range.restype = c_int
    L.pc_gz_member_start.argtypes = [c_cp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.pc_gz_member_start.restype = c_int
    L.pc_gzstream_next.argtypes = [c_vp, c_i64, c_i64, ctypes.POINTER(c_vp), ctypes.POINTER(c_int)]
    L.pc_gzstream_next.restype = c_int
    L.pc_gzstream_close.argtypes = [c_vp]
    L.pc_gzstream_close.restype = None
    L.pc_gz_sized_size.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    L.pc_gz_sized_size.restype = ctypes.c_int
    L.pc_gz_sized_find_record.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.pc_gz_sized_find_record.restype = ctypes.c_int
    L.pc_readset_load_gz_range.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(c_vp)]
    L.pc_readset_load_gz_range.restype = ctypes.c_int
    L.pc_pack_reads.argtypes = [c_vp, c_i64, c_vp, c_vp, c_i64, ctypes.POINTER(c_i64)]
    L.pc_pack_reads.restype = c_int
    L.pc_unpack_device.argtypes = [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_vp]
    L.pc_unpack_device.restype = c_int
    L.pc_memo_clear.argtypes = []
    L.pc_memo_clear.restype = None
    L.pc_memo_stats.argtypes = [ctypes.POINTER(c_i64)] * 3
    L.pc_memo_stats.restype = None
    _lib = L
    return L


def check(rc, what="porechop_amd"):
    if rc != 0:
        L = load_library()
        raise RuntimeError("%s failed: %s (%d)" % (what, L.pc_strerror(rc).decode(), rc))
