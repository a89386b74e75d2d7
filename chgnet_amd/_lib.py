"""ctypes binding of libchgnet_hip.so (include/chgnet_hip.h).  No fallback: a missing or
unloadable library raises ``RuntimeError`` -- the product never computes on the CPU."""

from __future__ import annotations

import ctypes
import os

c_int_p = ctypes.POINTER(ctypes.c_int32)
c_float_p = ctypes.POINTER(ctypes.c_float)

TASK_BITS = {"e": 1, "f": 2, "s": 4, "m": 8}

EXPORTED_SYMBOLS = (
    "chg_abi_version", "chg_device_count", "chg_weights_required", "chg_engine_create", "chg_engine_destroy", "chg_last_error",
    "chg_batch_upload", "chg_batch_build", "chg_batch_build_predict", "chg_debug_fetch_i32", "chg_batch_update_geometry", "chg_batch_free", "chg_batch_device_bytes",
    "chg_predict", "chg_synchronize", "chg_batch_download", "chg_timer_start", "chg_timer_stop_ms",
    "chg_profile_enable", "chg_profile_reset", "chg_profile_count", "chg_profile_read",
    "chg_debug_fetch", "chg_test_rows_gemm", "chg_test_split_gemm",
    "chg_engine_set_memory_limit", "chg_engine_memory_info", "chg_batch_bytes_required",
    "chg_stream_copy", "chg_backward", "chg_engine_build_stats", "chg_engine_update_weights",
    "chg_engine_set_graph_search", "chg_engine_cell_stats",
    "chg_comm_unique_id", "chg_comm_create", "chg_comm_all_gather_f32", "chg_comm_all_reduce_sum_f32", "chg_comm_barrier",
    "chg_comm_destroy", "chg_comm_last_error",
    "chg_comm_all_gather_f32_device", "chg_comm_all_reduce_sum_f32_device", "chg_comm_reserve", "chg_comm_info",
    "chg_backward_allreduce", "chg_batch_all_gather_energy", "chg_engine_stream", "chg_engine_device",
    "chg_host_alloc", "chg_host_free",
)


class ModelDesc(ctypes.Structure):
    _fields_ = [
        ("n_conv", ctypes.c_int32), ("cutoff_coeff", ctypes.c_int32), ("is_intensive", ctypes.c_int32),
        ("has_composition", ctypes.c_int32), ("atom_graph_cutoff", ctypes.c_float),
        ("bond_graph_cutoff", ctypes.c_float), ("n_weights", ctypes.c_int64),
        ("n_mlp_hidden", ctypes.c_int32), ("mlp_out_bias", ctypes.c_int32),
    ]


class BatchHost(ctypes.Structure):
    _fields_ = [
        ("n_struct", ctypes.c_int32), ("n_atoms", ctypes.c_int32), ("n_directed", ctypes.c_int32),
        ("n_undirected", ctypes.c_int32), ("n_angles", ctypes.c_int32), ("n_bnodes", ctypes.c_int32),
        ("z", c_int_p), ("frac", c_float_p), ("lattice", c_float_p), ("atom_owner", c_int_p), ("atom_off", c_int_p),
        ("e_center", c_int_p), ("e_nbr", c_int_p), ("e_image", c_float_p), ("e_d2u", c_int_p), ("e_owner", c_int_p),
        ("e_rev", c_int_p), ("p_center", c_int_p), ("p_nbr", c_int_p),
        ("u_u2d", c_int_p), ("u_bnode", c_int_p), ("bn_und", c_int_p),
        ("a_ctr", c_int_p), ("a_b1c", c_int_p), ("a_b2c", c_int_p), ("a_d1", c_int_p), ("a_d2", c_int_p),
    ]


class StructsHost(ctypes.Structure):
    _fields_ = [("n_struct", ctypes.c_int32), ("n_atoms", ctypes.c_int32), ("z", c_int_p),
                ("frac", ctypes.POINTER(ctypes.c_double)), ("lattice", ctypes.POINTER(ctypes.c_double)), ("atom_off", c_int_p)]


class OutHost(ctypes.Structure):
    _fields_ = [(n, c_float_p) for n in ("energy", "force", "stress", "magmom", "site_energy", "atom_fea", "crystal_fea")]


_LIB = None


def hip_lib_path() -> str:
    from chgnet_amd.build import HIP_LIB

    return os.environ.get("CHGNET_HIP_LIB", HIP_LIB)   # override: kernel timing experiments only


ABI_VERSION = 3   # include/chgnet_hip.h CHG_ABI_VERSION


def load() -> ctypes.CDLL:
    """Load the HIP engine library; raise loudly if it is not built or cannot be loaded."""
    global _LIB  # noqa: PLW0603
    if _LIB is not None:
        return _LIB
    path = hip_lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"chgnet_amd: HIP extension {path} is missing. Build it with `python -m chgnet_amd.build` "
            "(needs hipcc). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as exc:
        raise RuntimeError(f"chgnet_amd: cannot load HIP extension {path}: {exc}") from exc
    # the struct layouts below are those of include/chgnet_hip.h at this interface version: a library built from another one is refused
    try:
        lib.chg_abi_version.restype = ctypes.c_int
        found = int(lib.chg_abi_version())
    except AttributeError:
        found = -1
    if found != ABI_VERSION:
        raise RuntimeError(f"chgnet_amd: HIP extension {path} has interface version {found}, this binding was written for {ABI_VERSION}: "
                           "rebuild it with `python -m chgnet_amd.build --force`")
    vp = ctypes.c_void_p
    lib.chg_device_count.restype = ctypes.c_int
    lib.chg_weights_required.argtypes = [ctypes.c_int32]
    lib.chg_weights_required.restype = ctypes.c_int64
    lib.chg_engine_create.argtypes = [ctypes.POINTER(ModelDesc), c_float_p, ctypes.c_int, ctypes.POINTER(vp)]
    lib.chg_engine_destroy.argtypes = [vp]
    lib.chg_last_error.argtypes = [vp]
    lib.chg_last_error.restype = ctypes.c_char_p
    lib.chg_engine_build_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.chg_engine_set_graph_search.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
    u8p = ctypes.POINTER(ctypes.c_uint8)
    lib.chg_comm_unique_id.argtypes = [u8p]
    lib.chg_comm_create.argtypes = [u8p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(vp)]
    lib.chg_comm_all_gather_f32.argtypes = [vp, c_float_p, ctypes.c_int64, c_float_p]
    lib.chg_comm_all_reduce_sum_f32.argtypes = [vp, c_float_p, ctypes.c_int64]
    lib.chg_comm_all_gather_f32_device.argtypes = [vp, vp, ctypes.c_int64, vp, vp]
    lib.chg_comm_all_reduce_sum_f32_device.argtypes = [vp, vp, ctypes.c_int64, vp]
    lib.chg_comm_reserve.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(vp)]
    lib.chg_comm_info.argtypes = [vp, c_int_p, c_int_p, c_int_p]
    lib.chg_comm_barrier.argtypes = [vp]
    lib.chg_comm_destroy.argtypes = [vp]
    lib.chg_comm_last_error.argtypes = [vp]
    lib.chg_comm_last_error.restype = ctypes.c_char_p
    lib.chg_engine_cell_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.chg_engine_set_memory_limit.argtypes = [vp, ctypes.c_int64]
    lib.chg_engine_memory_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.chg_batch_bytes_required.argtypes = [ctypes.c_int32] * 6
    lib.chg_batch_bytes_required.restype = ctypes.c_int64
    lib.chg_batch_upload.argtypes = [vp, ctypes.POINTER(BatchHost), ctypes.POINTER(vp)]
    lib.chg_host_alloc.argtypes = [ctypes.c_int64, ctypes.POINTER(vp)]
    lib.chg_host_free.argtypes = [vp]
    lib.chg_batch_build.argtypes = [vp, ctypes.POINTER(StructsHost), ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.POINTER(vp), c_int_p]
    lib.chg_batch_build_predict.argtypes = [vp, ctypes.POINTER(StructsHost), ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint32,
                                            ctypes.POINTER(vp), c_int_p]
    lib.chg_debug_fetch_i32.argtypes = [vp, vp, ctypes.c_char_p, c_int_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    lib.chg_batch_update_geometry.argtypes = [vp, vp, c_float_p, c_float_p]
    lib.chg_batch_free.argtypes = [vp, vp]
    lib.chg_batch_device_bytes.argtypes = [vp]
    lib.chg_batch_device_bytes.restype = ctypes.c_int64
    lib.chg_predict.argtypes = [vp, vp, ctypes.c_uint32]
    lib.chg_synchronize.argtypes = [vp]
    lib.chg_backward.argtypes = [vp, vp, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
    lib.chg_backward_allreduce.argtypes = [vp, vp, c_float_p, c_float_p, c_float_p, c_float_p, vp, c_float_p]
    lib.chg_batch_all_gather_energy.argtypes = [vp, vp, vp, ctypes.c_int64, c_float_p]
    lib.chg_engine_stream.argtypes = [vp]
    lib.chg_engine_stream.restype = ctypes.c_void_p
    lib.chg_engine_device.argtypes = [vp]
    lib.chg_engine_update_weights.argtypes = [vp, c_float_p]
    lib.chg_batch_download.argtypes = [vp, vp, ctypes.POINTER(OutHost)]
    lib.chg_timer_start.argtypes = [vp]
    lib.chg_timer_stop_ms.argtypes = [vp, c_float_p]
    lib.chg_stream_copy.argtypes = [vp, ctypes.c_int64, ctypes.c_int, c_float_p]
    lib.chg_profile_enable.argtypes = [vp, ctypes.c_int]
    lib.chg_profile_reset.argtypes = [vp]
    lib.chg_profile_count.argtypes = [vp]
    lib.chg_profile_read.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]
    lib.chg_debug_fetch.argtypes = [vp, vp, ctypes.c_char_p, c_float_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    lib.chg_test_rows_gemm.argtypes = [vp, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.chg_test_split_gemm.argtypes = [vp, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is ctypes.c_int and name not in ("chg_device_count", "chg_profile_count"):
            fn.restype = ctypes.c_int
    _LIB = lib
    return lib


def task_mask(task: str) -> int:
    mask = 0
    for ch in task:
        mask |= TASK_BITS[ch]
    return mask
