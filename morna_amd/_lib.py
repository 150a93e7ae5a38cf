"""ctypes binding of libmorna_hip.so (C ABI: include/morna_hip.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible when an index is created, the error is raised to the caller.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MORNA_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmorna_hip.so")   # MORNA_LIB: an experimental build

OK, E_INVALID, E_HIP, E_STATE, E_RANGE, E_IO, E_EMPTY = 0, -1, -2, -3, -4, -5, -6

T_FEATURES, T_TWO_MEANS, T_SPLIT, T_PARTITION, T_QUERY, T_EXACT, T_QUERY_FILTER, T_EXACT_SCAN, T_SPLIT_MM, T_TM_STRIP, T_TM_WAVE = range(11)
COMM_ID_BYTES = 128
TIMER_NAMES = ["features", "two_means", "split", "partition", "query", "exact", "query_filter", "exact_scan", "split_mm",
               "tm_strip", "tm_wave"]


class ForestStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_items", "dim", "leaf_capacity", "n_trees", "n_nodes", "n_split", "n_leaves", "max_depth",
        "split_attempts", "split_rows", "fallback_nodes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# every symbol include/morna_hip.h declares: (restype, argtypes)
_p, _i32, _i64, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
SIGNATURES = {
    "morna_index_create": (C.c_int, [_i32, _i32, C.POINTER(_p)]),
    "morna_index_destroy": (C.c_int, [_p]),
    "morna_device_count": (C.c_int, [C.POINTER(_i32)]),
    "morna_last_error": (C.c_char_p, []),
    "morna_hash32": (_i32, [_p, _i64]),
    "morna_add_item": (C.c_int, [_p, _i32, _p]),
    "morna_add_items_f32": (C.c_int, [_p, _i32, _p, _i64]),
    "morna_stage_junctions": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p, _p]),
    "morna_build_features": (C.c_int, [_p, _i64]),
    "morna_stage_item_order": (C.c_int, [_p, _p, _i64]),
    "morna_unstage_junctions": (C.c_int, [_p]),
    "morna_hash_keys": (C.c_int, [_p, _p, _p, _i64, _p, _p, _p]),
    "morna_parse_intropolis": (C.c_int, [C.c_char_p, _i64, _i64, C.POINTER(_p)]),
    "morna_lines_counts": (C.c_int, [_p, _p]),
    "morna_lines_arrays": (C.c_int, [_p] + [C.POINTER(_p)] * 7),
    "morna_lines_freq_entry": (C.c_int, [_p, _i64, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i64)]),
    "morna_stage_lines": (C.c_int, [_p, _p]),
    "morna_lines_save": (C.c_int, [_p, C.c_char_p, _p]),
    "morna_lines_load": (C.c_int, [C.c_char_p, _p, C.POINTER(_p)]),
    "morna_lines_free": (C.c_int, [_p]),
    "morna_lines_shard": (C.c_int, [_p, _i32, _i32, C.POINTER(_p)]),
    "morna_lines_shard_info": (C.c_int, [_p, _p]),
    "morna_lines_from_arrays": (C.c_int, [_p, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, C.POINTER(_p)]),
    "morna_write_intropolis": (C.c_int, [C.c_char_p, _p, _p, _i64, _p, _p, _p]),
    "morna_merge_topk": (C.c_int, [_p, _p, _i32, _i64, _i32, _i32, _p, _p, _p]),
    "morna_get_nns_by_vector_packed": (C.c_int, [_p, _p, _i64, _i32, _i32, _i64, _p]),
    "morna_merge_topk_packed": (C.c_int, [_p, _p, _i32, _i64, _i32, _i32, _p, _p, _p]),
    "morna_get_item_vectors_dev": (C.c_int, [_p, _p, _i64, _p]),
    "morna_get_stream": (C.c_int, [_p, _p]),
    "morna_get_n_items": (_i64, [_p]),
    "morna_get_item_vector": (C.c_int, [_p, _i32, _p]),
    "morna_get_item_vectors": (C.c_int, [_p, _p, _i64, _p]),
    "morna_get_items": (C.c_int, [_p, _p]),
    "morna_get_norms2": (C.c_int, [_p, _p]),
    "morna_build": (C.c_int, [_p, _i32, _u32]),
    "morna_get_n_trees": (_i32, [_p]),
    "morna_get_forest_stats": (C.c_int, [_p, C.POINTER(ForestStats)]),
    "morna_get_forest": (C.c_int, [_p, _p, _p, _p, _p]),
    "morna_get_nns_by_vector": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "morna_get_nns_by_item": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "morna_exact_search": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "morna_exact_search_by_item": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "morna_comm_unique_id": (C.c_int, [_p]),
    "morna_comm_init": (C.c_int, [_p, _p, _i32, _i32]),
    "morna_comm_destroy": (C.c_int, [_p]),
    "morna_comm_info": (C.c_int, [_p, C.POINTER(_i32), C.POINTER(_i32), _p]),
    "morna_get_nns_by_vector_sharded": (C.c_int, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    "morna_get_nns_by_item_sharded": (C.c_int, [_p, _p, _i64, _p, _i32, _i32, _p, _p, _p]),
    "morna_exact_search_sharded": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "morna_exact_search_by_item_sharded": (C.c_int, [_p, _p, _i64, _p, _i32, _p, _p, _p]),
    "morna_exact_packed_bytes": (_i64, [_i64, _i32]),
    "morna_exact_search_packed": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i64, _p]),
    "morna_merge_exact_packed": (C.c_int, [_p, _p, _i32, _i64, _i32, _i32, _p, _p, _p]),
    "morna_save": (C.c_int, [_p, C.c_char_p]),
    "morna_load": (C.c_int, [_p, C.c_char_p]),
    "morna_timer_enable": (C.c_int, [_p, _i32]),
    "morna_timer_reset": (C.c_int, [_p]),
    "morna_timer_read": (C.c_int, [_p, _i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64)]),
    "morna_synchronize": (C.c_int, [_p]),
}

_lib = None


def lib():
    """Load libmorna_hip.so; raises if it has not been built (python -m morna_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -m morna_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_EXC = {E_INVALID: ValueError, E_HIP: RuntimeError, E_STATE: RuntimeError, E_RANGE: IndexError,
        E_IO: IOError, E_EMPTY: ValueError}


def check(rc):
    """Map a MORNA_E_* code to the exception type the reference call sites raise."""
    if rc != OK:
        msg = lib().morna_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, RuntimeError)(msg)


def device_count():
    """HIP devices the library sees (raises when there is none)."""
    n = _i32(0)
    check(lib().morna_device_count(C.byref(n)))
    return int(n.value)


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
