"""ctypes binding of librad_hip.so (the C ABI declared in include/rad_hip.h).

There is no CPU fallback: if the shared library is missing this module raises,
and every device entry point raises ``RadHipError`` when no gfx950 GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADHIP_LIB") or os.path.join(_HERE, "_build", "librad_hip.so")

NO_SLOT = 0xFFFFFFFF
TRAV_LOG_POPS = 1
SHARD_OWN_STREAM = 2
TRAV_SLOTS = 4          # heavy traversal state per resident row of the kernel instead of per traversal
TRAV_OWN_STREAM = 8     # the traversal object gets a HIP stream of its own (start / finish overlap)

E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM, E_STATE, E_CAPACITY, E_RANGE, E_COMM = range(-1, -9, -1)


class RadHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"librad_hip error {code}: {message}")
        self.code = code


class IndexInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("ndim_bits", C.c_uint32), ("row_bytes", C.c_uint32),
                ("row_stride", C.c_uint32), ("connectivity", C.c_uint32),
                ("connectivity_base", C.c_uint32), ("expansion_add", C.c_uint32),
                ("max_level", C.c_int32), ("entry", C.c_uint32), ("n_upper_rows", C.c_uint64),
                ("device_bytes", C.c_uint64), ("device", C.c_int32), ("has_vectors", C.c_int32),
                ("has_graph", C.c_int32), ("sharded", C.c_int32), ("shard_first", C.c_uint64),
                ("shard_rows", C.c_uint64)]


class CommInfo(C.Structure):
    _fields_ = [("rccl_version", C.c_int32), ("comm_count", C.c_int32), ("comm_rank", C.c_int32),
                ("device", C.c_int32), ("pci_bus_id", C.c_char * 32)]


class LayoutInfo(C.Structure):
    _fields_ = [("valid", C.c_int32), ("group", C.c_uint32), ("id_limit", C.c_uint64),
                ("groups_per_row", C.c_double), ("degree", C.c_double), ("seconds", C.c_double)]


class TravStats(C.Structure):
    _fields_ = [("n_scored", C.c_uint64), ("n_pops", C.c_uint64), ("n_nbr", C.c_uint64),
                ("n_repivot", C.c_uint64), ("n_flush", C.c_uint64),
                ("status", C.c_int32), ("n_remid", C.c_int32), ("n_upper", C.c_uint64)]


_P = C.c_void_p
_U32, _U64, _I32 = C.c_uint32, C.c_uint64, C.c_int32

# name -> (restype, argtypes); every symbol include/rad_hip.h declares
SIGNATURES = {
    "radhip_last_error": (C.c_char_p, []),
    "radhip_backend_name": (C.c_char_p, []),
    "radhip_abi_version": (C.c_int, []),
    "radhip_build_id": (C.c_char_p, []),
    "radhip_arm_last_words": (C.c_int, [C.c_char_p]),
    "radhip_traverse_build_id": (C.c_char_p, []),
    "radhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "radhip_index_create": (C.c_int, [_U32, _U32, _U32, _U32, C.c_int, C.POINTER(_P)]),
    "radhip_index_destroy": (C.c_int, [_P]),
    "radhip_index_info": (C.c_int, [_P, C.POINTER(IndexInfo)]),
    "radhip_index_load_vectors": (C.c_int, [_P, _P, _U64]),
    "radhip_index_synth_vectors": (C.c_int, [_P, _U64, _U64, _U64, _U64, C.c_int]),
    "radhip_index_read_vectors": (C.c_int, [_P, _U64, _U64, _P]),
    "radhip_index_load_vectors_shard": (C.c_int, [_P, _P, _U64, _U64, _U64]),
    "radhip_index_synth_vectors_shard": (C.c_int, [_P, _U64, _U64, _U64, _U64, C.c_int]),
    "radhip_index_link_resident": (C.c_int, [_P, _U64, _U32]),
    "radhip_index_broadcast_graph": (C.c_int, [_P, _P, C.c_int]),
    "radhip_comm_info": (C.c_int, [_P, C.POINTER(CommInfo)]),
    "radhip_shard_run_pair": (C.c_int, [_P, _P, _P, _P, _U64, C.POINTER(_U64)]),
    "radhip_shard_speculation": (C.c_int, [_P, C.POINTER(_U32), C.POINTER(_U64), C.POINTER(_U64), C.POINTER(_U64)]),
    "radhip_index_load_graph": (C.c_int, [_P, _U64, _I32, _U32, _P, _P, _P, _P, _U64]),
    "radhip_index_synth_graph": (C.c_int, [_P, _U64]),
    "radhip_index_read_graph": (C.c_int, [_P, _P, _P, _P, _P]),
    "radhip_get_neighbors": (C.c_int, [_P, _U32, _I32, _P, _U32, C.POINTER(_U32)]),
    "radhip_get_top_level_nodes": (C.c_int, [_P, _P, _U64, C.POINTER(_U64)]),
    "radhip_index_optimize_layout": (C.c_int, [_P, _U32]),
    "radhip_index_layout_info": (C.c_int, [_P, C.POINTER(LayoutInfo)]),
    "radhip_index_set_layout": (C.c_int, [_P, _P]),
    "radhip_index_read_layout": (C.c_int, [_P, _P]),
    "radhip_index_set_keys": (C.c_int, [_P, _U64, _P, _U64]),
    "radhip_index_read_keys": (C.c_int, [_P, _U64, _U64, _P]),
    "radhip_keys_from_slots": (C.c_int, [_P, _P, _U64, _P]),
    "radhip_slots_from_keys": (C.c_int, [_P, _P, _U64, _P, C.POINTER(_U64)]),
    "radhip_get_neighbors_keyed": (C.c_int, [_P, _U32, _I32, _P, _U32, C.POINTER(_U32)]),
    "radhip_tanimoto_scan": (C.c_int, [_P, _P, _U32, _U64, _U64, _P, _P]),
    "radhip_tanimoto_gather": (C.c_int, [_P, _P, _U32, _P, _P, _P, _P]),
    "radhip_last_kernel_ms": (C.c_double, []),
    "radhip_tanimoto_topk": (C.c_int, [_P, _P, _U32, _U32, _U64, _U64, _P, _P, _P, _P]),
    "radhip_distance_f32": (C.c_float, [_U32, _U32]),
    "radhip_index_add": (C.c_int, [_P, _P, _U64, _U64, _U32]),
    "radhip_search": (C.c_int, [_P, _P, _U32, _U32, _U32, _P, _P, _P, _P, _P, _P]),
    "radhip_level_of": (C.c_int, [_U64, _U64, _U32]),
    "radhip_traversal_create": (C.c_int, [_P, _P, _U32, _U64, _U32, C.POINTER(_P)]),
    "radhip_traversal_destroy": (C.c_int, [_P]),
    "radhip_traversal_reset": (C.c_int, [_P, _P]),
    "radhip_index_peer_create": (C.c_int, [_P, C.c_int, C.c_int, _U64, C.POINTER(_U64)]),
    "radhip_index_peer_fill_synth": (C.c_int, [_P, _U64, C.c_int]),
    "radhip_index_peer_fill_rows": (C.c_int, [_P, _P, _U64]),
    "radhip_index_peer_export": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "radhip_index_peer_import": (C.c_int, [_P, C.c_int, C.c_int]),
    "radhip_index_peer_seal": (C.c_int, [_P]),
    "radhip_index_copy_graph_from": (C.c_int, [_P, _P]),
    "radhip_traversal_run": (C.c_int, [_P, _U64, C.POINTER(_U32)]),
    "radhip_traversal_reset_count": (C.c_int, [_P, _P, _U32]),
    "radhip_traversal_create_ring": (C.c_int, [_P, _P, _U32, _U64, _U32, _U32, C.POINTER(_P)]),
    "radhip_traversal_list_ring": (_U32, [_P]),
    "radhip_traversal_start": (C.c_int, [_P]),
    "radhip_traversal_finish": (C.c_int, [_P, C.POINTER(_U32)]),
    "radhip_traversal_elapsed_between": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "radhip_traversal_slots": (_U32, [_P]),
    "radhip_traversal_launch_interval": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "radhip_traversal_stats": (C.c_int, [_P, _P]),
    "radhip_traversal_results": (C.c_int, [_P, _U32, _P, _P, _P, _U64, C.POINTER(_U64)]),
    "radhip_traversal_pop_log": (C.c_int, [_P, _U32, _P, _P, _U64, C.POINTER(_U64)]),
    "radhip_traversal_result_hashes": (C.c_int, [_P, _U32, _U32, _P]),
    "radhip_traversal_kernel_time": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(_U64)]),
    "radhip_traversal_resident_capacity": (C.c_int, [_P, C.POINTER(_U32)]),
    "radhip_traversal_state_bytes": (_U64, [_P]),
    "radhip_traversal_kernel": (C.c_int, [_P]),
    "radhip_traversal_table": (C.c_int, [_P]),
    "radhip_traversal_set_targets": (C.c_int, [_P, _P]),
    "radhip_traversal_frontier": (C.c_int, [_P, _P, _P]),
    "radhip_index_keep_rows": (C.c_int, [_P, _U64, _U64]),
    "radhip_shard_create": (C.c_int, [_P, C.c_int, C.c_int, _U64, _U64, _P, _U32, _U64, _U32, C.POINTER(_P)]),
    "radhip_shard_destroy": (C.c_int, [_P]),
    "radhip_shard_reset": (C.c_int, [_P, _P]),
    "radhip_shard_run": (C.c_int, [_P, _P, _U64, C.POINTER(_U64)]),
    "radhip_shard_width": (_U32, [_P]),
    "radhip_shard_slots": (_U32, [_P]),
    "radhip_shard_engine": (C.c_int, [_P]),
    "radhip_shard_step": (C.c_int, [_P, C.POINTER(_U32)]),
    "radhip_shard_get_requests": (C.c_int, [_P, _P]),
    "radhip_shard_set_requests_all": (C.c_int, [_P, _P]),
    "radhip_shard_evaluate": (C.c_int, [_P]),
    "radhip_shard_get_scores_out": (C.c_int, [_P, _P]),
    "radhip_shard_set_scores_in": (C.c_int, [_P, _P]),
    "radhip_shard_stats": (C.c_int, [_P, _P]),
    "radhip_shard_results": (C.c_int, [_P, _U32, _P, _P, _P, _U64, C.POINTER(_U64)]),
    "radhip_shard_pop_log": (C.c_int, [_P, _U32, _P, _P, _U64, C.POINTER(_U64)]),
    "radhip_shard_timing": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_U64), C.POINTER(_U64)]),
    "radhip_shard_state_bytes": (_U64, [_P]),
    "radhip_comm_unique_id": (C.c_int, [_P]),
    "radhip_comm_create": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, C.POINTER(_P)]),
    "radhip_comm_destroy": (C.c_int, [_P]),
    "radhip_comm_allgather_u64": (C.c_int, [_P, _P, _U64, _P]),
    "radhip_comm_rank": (C.c_int, [_P]),
    "radhip_comm_world": (C.c_int, [_P]),
    "radhip_rad_key": (_U64, [_U32, _U32, _U32, _U32]),
    "radhip_debug_device_keys": (C.c_int, [_P, _P, _P, _P, _P, _U64, _P]),
    "radhip_debug_sort_staging": (C.c_int, [_P, _P, _P, _U32, _P]),
    "radhip_debug_staging_capacity": (_U32, []),
    "radhip_rad_key_decode": (None, [_U64, C.POINTER(_U32), C.POINTER(_U32)]),
}

_lib = None


def lib() -> C.CDLL:
    """Load librad_hip.so; raises if it was not built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C rad_amd/csrc` (hipcc, gfx950). rad_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.radhip_abi_version() != 1:
            raise ImportError("librad_hip ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RadHipError(rc, lib().radhip_last_error().decode("utf-8", "replace"))


def build_id() -> str:
    return lib().radhip_build_id().decode("ascii", "replace")


def traverse_build_id() -> str:
    return lib().radhip_traverse_build_id().decode("ascii", "replace")


def device_count() -> int:
    n = C.c_int(0)
    check(lib().radhip_device_count(C.byref(n)))
    return n.value


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def as_rows(a, row_bytes: int, what: str = "vectors") -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != row_bytes:
        raise ValueError(f"{what} must have shape (n, {row_bytes}) uint8 (np.packbits output), got {a.shape}")
    return a
