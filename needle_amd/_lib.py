"""ctypes binding of libneedle_hip.so (include/needle_hip.h).  Fails loudly if the library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEEDLE_LIB", os.path.join(_HERE, "libneedle_hip.so"))  # NEEDLE_LIB: kernel-variant A/B runs only

NEEDLE_OK, ERR_INVALID, ERR_SYNTAX, ERR_COMPILE, ERR_UNSUPPORTED, ERR_DEVICE = 0, 1, 2, 3, 4, 5

EXPORTS = [
    "needle_version", "needle_last_error", "needle_device_count", "needle_trim_scratch", "needle_tuning_info", "needle_compile", "needle_pattern_from_tables",
    "needle_pattern_destroy", "needle_pattern_serialize", "needle_pattern_deserialize", "needle_pattern_get_info", "needle_pattern_program_info", "needle_pattern_prefilter_info", "needle_pattern_prefilter_info2", "needle_pattern_set_prefilter", "needle_pattern_prefilter_state", "needle_pattern_utf16_route", "needle_pattern_match_lengths", "needle_pattern_find_all_transducer", "needle_pattern_get_class_map", "needle_pattern_get_table",
    "needle_matches_dev", "needle_contained_in_dev", "needle_find_dev", "needle_find_packed16_dev", "needle_find_packed8_dev", "needle_find_next_dev", "needle_find_all_dev", "needle_find_all_packed16_dev", "needle_find_all_blocked16_dev", "needle_find_all_compact16_dev", "needle_count_matches_dev", "needle_find_all_csr_dev", "needle_find_all_host", "needle_find_all_packed16_host", "needle_find_all_csr_host",
    "needle_pack_start_end16_dev", "needle_unpack_start_end16_dev", "needle_matches_host",
    "needle_contained_in_host", "needle_find_host", "needle_find_compact_dev", "needle_find_compact_host", "needle_find_packed16_host", "needle_find_packed8_host", "needle_matcher_create", "needle_matcher_destroy",
    "needle_matcher_matches", "needle_matcher_contained_in", "needle_matcher_find", "needle_matcher_find_range",
    "needle_matcher_start", "needle_matcher_end", "needle_rows_from_packed_dev", "needle_matches_packed_host",
    "needle_contained_in_packed_host", "needle_find_packed_host",
    "needle_multi_create", "needle_multi_destroy", "needle_multi_device_count", "needle_multi_stream", "needle_multi_transport", "needle_multi_scan",
    "needle_multi_sync", "needle_scan_host_multi", "needle_multi_unique_id", "needle_multi_create_rank",
    "needle_multi_all_gather_u64", "needle_multi_gather_i32",
]


class DfaDesc(ctypes.Structure):
    _fields_ = [("n_states", ctypes.c_int32), ("max_char", ctypes.c_int32), ("table", ctypes.c_void_p),
                ("table_string", ctypes.c_char_p), ("accepting", ctypes.c_void_p)]


class TableDesc(ctypes.Structure):
    _fields_ = [("class_map", ctypes.c_void_p), ("stride", ctypes.c_int32), ("matches", DfaDesc),
                ("contained_in", DfaDesc), ("forwards", DfaDesc), ("backwards", DfaDesc), ("fixed_len", ctypes.c_int32)]


class BatchView(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_void_p), ("char_width", ctypes.c_uint32), ("n_rows", ctypes.c_uint64),
                ("row_stride", ctypes.c_uint64), ("row_len", ctypes.c_uint32), ("lengths", ctypes.c_void_p)]


class PackedView(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("char_width", ctypes.c_uint32), ("n_rows", ctypes.c_uint64),
                ("offsets", ctypes.c_void_p)]


class PatternInfo(ctypes.Structure):
    _fields_ = [("stride", ctypes.c_int32), ("n_states", ctypes.c_int32 * 4), ("max_char", ctypes.c_int32 * 4),
                ("fixed_len", ctypes.c_int32), ("min_len", ctypes.c_int32), ("max_len", ctypes.c_int32),
                ("kernel_mode", ctypes.c_int32 * 4)]


class ProgramInfo(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("mode", "n_states", "lds_bytes", "blob_bytes", "waves", "tile_bytes",
                                              "dense_rows", "records", "chains", "hot_rows", "window", "window_lo", "window_hi", "lengths_form")]


class PrefilterState(ctypes.Structure):
    _fields_ = ([(k, ctypes.c_int32) for k in ("mode", "has_filter", "suspended_calls_left", "backoff")] + [("last_candidates_per_kib", ctypes.c_float)] +
                [(k, ctypes.c_uint64) for k in ("filter_launches", "suspended_calls")])


class PrefilterInfo(ctypes.Structure):
    _fields_ = ([(k, ctypes.c_int32) for k in ("on", "mode", "stride", "warm", "min_len", "n_windows", "bitmap_bytes")] +
                [(k, ctypes.c_uint32) for k in ("m1", "m2", "addr_shift", "addr_mask")] + [("why", ctypes.c_char * 96)] +
                [(k, ctypes.c_int32) for k in ("on2", "n_windows2", "bitmap2_bytes")] + [(k, ctypes.c_uint32) for k in ("m3", "addr_mask2")])


class PrefilterInfo2(ctypes.Structure):
    _fields_ = [("base", PrefilterInfo), ("wide", ctypes.c_int32), ("m1b", ctypes.c_uint32), ("m2b", ctypes.c_uint32)]


_lib = None


def lib():
    """The loaded library.  Import torch first when both are used so that one HIP runtime (torch's) serves both."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libneedle_hip.so is not built (%s). Run `python -m needle_amd.build` "
                          "(needs hipcc); there is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever is mapped first serves
    # the whole process, and two HSA runtimes in one process cannot both open the GPU -- so when torch is
    # installed make sure ITS runtime is the one already loaded before our DT_NEEDED is resolved.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    P, I, VP = ctypes.POINTER, ctypes.c_int, ctypes.c_void_p
    L.needle_version.restype = ctypes.c_char_p
    L.needle_last_error.restype = ctypes.c_char_p
    L.needle_device_count.restype = I
    L.needle_compile.argtypes = [VP, ctypes.c_size_t, I, P(VP)]
    L.needle_pattern_from_tables.argtypes = [P(TableDesc), P(VP)]
    L.needle_pattern_destroy.argtypes = [VP]
    L.needle_pattern_destroy.restype = None
    L.needle_pattern_serialize.argtypes = [VP, VP, ctypes.c_size_t, P(ctypes.c_size_t)]
    L.needle_pattern_deserialize.argtypes = [VP, ctypes.c_size_t, P(VP)]
    L.needle_pattern_get_info.argtypes = [VP, P(PatternInfo)]
    L.needle_pattern_match_lengths.argtypes = [VP, P(ctypes.c_int32), P(ctypes.c_int32), P(ctypes.c_int32), P(ctypes.c_int32), VP, VP, VP]
    L.needle_pattern_find_all_transducer.argtypes = [VP, ctypes.c_int, P(ctypes.c_int32), P(ctypes.c_int32), VP, ctypes.c_size_t, P(ctypes.c_size_t)]
    L.needle_pattern_program_info.argtypes = [VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, P(ProgramInfo)]
    L.needle_pattern_prefilter_info.argtypes = [VP, ctypes.c_int, P(PrefilterInfo), VP]
    L.needle_pattern_prefilter_info2.argtypes = [VP, ctypes.c_int, ctypes.c_int, P(PrefilterInfo2), VP, ctypes.c_size_t]
    L.needle_pattern_set_prefilter.argtypes = [VP, ctypes.c_int]
    L.needle_pattern_prefilter_state.argtypes = [VP, ctypes.c_int, P(PrefilterState)]
    L.needle_pattern_utf16_route.argtypes = [VP, P(ctypes.c_int32), P(ctypes.c_int32)]
    L.needle_pattern_get_class_map.argtypes = [VP, VP]
    L.needle_pattern_get_table.argtypes = [VP, I, VP, VP]
    for n in ("needle_matches_dev", "needle_contained_in_dev"):
        getattr(L, n).argtypes = [VP, P(BatchView), VP, VP]
    L.needle_find_dev.argtypes = [VP, P(BatchView), VP, VP, VP, VP]
    L.needle_find_packed16_dev.argtypes = [VP, P(BatchView), VP, VP, VP]
    L.needle_find_packed8_dev.argtypes = [VP, P(BatchView), VP, VP, VP]
    L.needle_find_next_dev.argtypes = [VP, P(BatchView), VP, VP, VP, VP, VP]
    L.needle_find_all_dev.argtypes = [VP, P(BatchView), ctypes.c_uint32, VP, VP, VP, P(I), VP]
    L.needle_find_all_packed16_dev.argtypes = [VP, P(BatchView), ctypes.c_uint32, VP, VP, P(I), VP]
    L.needle_find_all_blocked16_dev.argtypes = [VP, P(BatchView), ctypes.c_uint32, VP, VP, P(I), VP]
    L.needle_find_all_compact16_dev.argtypes = [VP, P(BatchView), ctypes.c_uint32, VP, VP, ctypes.c_uint64, VP, P(I), VP]
    L.needle_find_all_packed16_host.argtypes = [VP, P(BatchView), ctypes.c_uint32, VP, VP, P(I)]
    L.needle_count_matches_dev.argtypes = [VP, P(BatchView), VP, VP]
    L.needle_find_all_csr_dev.argtypes = [VP, P(BatchView), VP, VP, VP, P(I), VP]
    L.needle_pack_start_end16_dev.argtypes = [VP, VP, ctypes.c_uint64, VP, VP]
    L.needle_unpack_start_end16_dev.argtypes = [VP, ctypes.c_uint64, VP, VP, VP]
    L.needle_find_all_csr_host.argtypes = [VP, P(BatchView), VP, VP, VP, ctypes.c_uint64, P(ctypes.c_uint64)]
    L.needle_find_all_host.argtypes = [VP, P(BatchView), ctypes.c_uint32, VP, VP, VP, P(I)]
    for n in ("needle_matches_host", "needle_contained_in_host"):
        getattr(L, n).argtypes = [VP, P(BatchView), VP]
    L.needle_find_host.argtypes = [VP, P(BatchView), VP, VP, VP]
    L.needle_find_compact_dev.argtypes = [VP, P(BatchView), VP, VP, ctypes.c_uint64, VP, VP]
    L.needle_find_compact_host.argtypes = [VP, P(BatchView), VP, VP, ctypes.c_uint64, P(ctypes.c_uint64)]
    L.needle_find_packed16_host.argtypes = [VP, P(BatchView), VP, VP]
    L.needle_find_packed8_host.argtypes = [VP, P(BatchView), VP, VP]
    L.needle_rows_from_packed_dev.argtypes = [P(PackedView), VP, ctypes.c_uint64, VP, VP, VP]
    for n in ("needle_matches_packed_host", "needle_contained_in_packed_host"):
        getattr(L, n).argtypes = [VP, P(PackedView), VP]
    L.needle_find_packed_host.argtypes = [VP, P(PackedView), VP, VP, VP]
    L.needle_matcher_create.argtypes = [VP, VP, ctypes.c_size_t, P(VP)]
    L.needle_matcher_destroy.argtypes = [VP]
    L.needle_matcher_destroy.restype = None
    for n in ("needle_matcher_matches", "needle_matcher_contained_in", "needle_matcher_find"):
        getattr(L, n).argtypes = [VP, P(I)]
    L.needle_matcher_find_range.argtypes = [VP, I, I, P(I)]
    L.needle_matcher_start.argtypes = [VP]
    L.needle_matcher_end.argtypes = [VP]
    L.needle_multi_create.argtypes = [P(I), I, ctypes.c_uint, P(VP)]
    L.needle_multi_destroy.argtypes = [VP]
    L.needle_multi_destroy.restype = None
    L.needle_multi_device_count.argtypes = [VP]
    L.needle_multi_stream.argtypes = [VP, I]
    L.needle_multi_stream.restype = VP
    L.needle_multi_transport.argtypes = [VP]
    L.needle_multi_transport.restype = ctypes.c_char_p
    L.needle_multi_scan.argtypes = [VP, VP, I, P(BatchView), VP, VP, VP]
    L.needle_multi_sync.argtypes = [VP]
    L.needle_scan_host_multi.argtypes = [VP, VP, I, P(BatchView), VP, VP, VP]
    L.needle_multi_unique_id.argtypes = [VP]
    L.needle_multi_create_rank.argtypes = [VP, I, I, I, P(VP)]
    L.needle_multi_all_gather_u64.argtypes = [VP, VP, ctypes.c_uint64, VP, VP]
    L.needle_multi_gather_i32.argtypes = [VP, VP, ctypes.c_uint64, VP, VP]
    _lib = L
    return L


def last_error():
    return lib().needle_last_error().decode("utf-8", "replace")
