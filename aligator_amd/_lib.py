"""Loader of the C-ABI shared library (include/gar_hip.h) via ctypes.

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C aligator_amd/csrc`` into ``aligator_amd/libgar_hip.so``.  There is no
CPU fallback: if the library is missing, or no HIP device is visible when a
solver is created, the error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "libgar_hip.so")

_PD = C.POINTER(C.c_double)
_PI32 = C.POINTER(C.c_int32)
_PI64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes): every symbol include/gar_hip.h declares
SIGNATURES = {
    "gar_hip_version": (C.c_char_p, []),
    "gar_hip_last_error": (C.c_char_p, []),
    "gar_hip_device_count": (C.c_int, []),
    "gar_hip_stream_ceiling_ms": (C.c_double, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int]),
    "gar_hip_copy_ceiling_ms": (C.c_double, [C.c_int, C.c_int64, C.c_int]),
    "gar_hip_knot_doubles": (C.c_int64, [_PI32]),
    "gar_hip_factor_doubles": (C.c_int64, [_PI32]),
    "gar_hip_solver_create": (C.c_void_p, [C.c_int, C.c_int, _PI32, C.c_int, C.c_int, C.c_int]),
    "gar_hip_solver_create_sharded": (C.c_void_p, [C.c_int, C.c_int, _PI32, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_int]),
    "gar_hip_solver_create_ranked": (C.c_void_p, [C.c_int, C.c_int, _PI32, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_int]),
    "gar_hip_solver_create_dense": (C.c_void_p, [C.c_int, C.c_int, _PI32, C.c_int, C.c_int]),
    "gar_hip_solver_destroy": (None, [C.c_void_p]),
    "gar_hip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gar_hip_sync": (C.c_int, [C.c_void_p]),
    "gar_hip_problem_doubles": (C.c_int64, [C.c_void_p]),
    "gar_hip_factors_doubles": (C.c_int64, [C.c_void_p]),
    "gar_hip_solution_doubles": (C.c_int64, [C.c_void_p]),
    "gar_hip_batch": (C.c_int, [C.c_void_p]),
    "gar_hip_horizon": (C.c_int, [C.c_void_p]),
    "gar_hip_kernel_name": (C.c_char_p, [C.c_void_p]),
    "gar_hip_suggest_num_legs": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "gar_hip_stage_offsets": (C.c_int, [C.c_void_p, C.c_int, _PI64]),
    "gar_hip_init_offsets": (C.c_int, [C.c_void_p, _PI64]),
    "gar_hip_upload_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [_PD] * 16),
    "gar_hip_set_init": (C.c_int, [C.c_void_p, C.c_int, _PD, _PD]),
    "gar_hip_upload_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _PD]),
    "gar_hip_upload_packed_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gar_hip_commit": (C.c_int, [C.c_void_p]),
    "gar_hip_device_problems": (C.c_void_p, [C.c_void_p]),
    "gar_hip_device_factors": (C.c_void_p, [C.c_void_p]),
    "gar_hip_device_solutions": (C.c_void_p, [C.c_void_p]),
    "gar_hip_device_stage_layout": (C.c_int, [C.c_void_p, C.c_int, _PI64]),
    "gar_hip_device_sizes": (C.c_int, [C.c_void_p, _PI64]),
    "gar_hip_device_record_format": (C.c_int, [C.c_void_p]),
    "gar_hip_packed_stage_dims": (C.c_int, [C.c_void_p, C.c_int, _PI32]),
    "gar_hip_upload_packed_device_fmt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "gar_hip_backward": (C.c_int, [C.c_void_p, C.c_double]),
    "gar_hip_backward_async": (C.c_int, [C.c_void_p, C.c_double]),
    "gar_hip_backward_blocks": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), _PD, _PD, C.c_double]),
    "gar_hip_forward": (C.c_int, [C.c_void_p, _PD]),
    "gar_hip_forward_async": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gar_hip_num_failed": (C.c_int, [C.c_void_p]),
    "gar_hip_slow_path_stages": (C.c_int, [C.c_void_p, _PI64]),
    "gar_hip_constrained_bk_stages": (C.c_int, [C.c_void_p, _PI64]),
    "gar_hip_boundary_doubles": (C.c_int64, [C.c_void_p]),
    "gar_hip_device_boundary_local": (C.c_void_p, [C.c_void_p]),
    "gar_hip_device_boundary_all": (C.c_void_p, [C.c_void_p]),
    "gar_hip_backward_legs_async": (C.c_int, [C.c_void_p, C.c_double]),
    "gar_hip_condensed_solve_async": (C.c_int, [C.c_void_p]),
    "gar_hip_forward_legs_async": (C.c_int, [C.c_void_p]),
    "gar_hip_set_refinement": (C.c_int, [C.c_void_p, C.c_double, C.c_int]),
    "gar_hip_condensed_info": (C.c_int, [C.c_void_p, C.c_int, _PD]),
    "gar_hip_condensed_backward_error": (C.c_int, [C.c_void_p, C.c_int, _PD]),
    "gar_hip_condensed_resolved": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gar_hip_condensed_solver_name": (C.c_char_p, [C.c_void_p]),
    "gar_hip_set_condensed_backward_ok": (C.c_int, [C.c_void_p, C.c_double]),
    "gar_hip_get_solution": (C.c_int, [C.c_void_p, C.c_int, _PD, _PD, _PD, _PD]),
    "gar_hip_get_gains": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _PD, _PD, _PD]),
    "gar_hip_gains_doubles": (C.c_int, [C.c_void_p, _PI64]),
    "gar_hip_gains_offsets": (C.c_int, [C.c_void_p, C.c_int, _PI64]),
    "gar_hip_fetch_results": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gar_hip_prefetch_gains": (C.c_int, [C.c_void_p, C.c_int]),
    "gar_hip_host_results": (C.c_void_p, [C.c_void_p, _PI64]),
    "gar_hip_get_gains_all": (C.c_int, [C.c_void_p, C.c_int, _PD, _PD]),
    "gar_hip_get_value": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _PD, _PD, _PD, _PD, _PD]),
    "gar_hip_get_kkt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, _PD]),
    "gar_hip_get_initial": (C.c_int, [C.c_void_p, C.c_int, _PD, _PD, _PD, _PD]),
    "gar_hip_collapse_feedback": (C.c_int, [C.c_void_p]),
    "gar_hip_debug_trace": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]),
    "gar_hip_download_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _PD]),
    "gar_hip_deriv_doubles": (C.c_int64, [C.c_void_p]),
    "gar_hip_deriv_offsets": (C.c_int, [C.c_void_p, C.c_int, _PI64]),
    "gar_hip_update_lq_subproblem_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int]),
    "gar_hip_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "gar_hip_get_option": (C.c_char_p, [C.c_char_p]),
    "gar_hip_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "gar_hip_pipeline": (C.c_int, [C.c_void_p]),
    "gar_hip_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "gar_hip_last_kernel_ms": (C.c_int, [C.c_void_p, _PD]),
    "gar_hip_cycle_append": (C.c_int, [C.c_void_p, _PI32]),
    "gar_hip_multi_create": (C.c_void_p, [C.c_int, C.POINTER(C.c_int), C.c_int, _PI32, C.c_int, C.c_int, C.c_int]),
    "gar_hip_num_devices": (C.c_int, [C.c_void_p]),
    "gar_hip_stage_device": (C.c_int, [C.c_void_p, C.c_int]),
    "gar_hip_multi_exchange_name": (C.c_char_p, [C.c_void_p]),
    "gar_hip_debug_alloc_count": (C.c_longlong, []),
}


class GarLibraryError(RuntimeError):
    pass


_cache = {}


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64
    (SONAME libamdhip64.so.7, same as /opt/rocm's); if torch initialises after a
    different copy is already mapped, the process ends up with two runtimes and
    the second one sees no GPU.  Importing torch first makes the backend bind
    (by SONAME) to the copy torch uses, so torch tensors, streams and RCCL
    buffers are valid in our kernels' address space."""
    try:
        import torch  # noqa: F401  (plumbing: device memory, streams, RCCL)
    except ImportError:
        pass  # plain C/C++ consumers use the system runtime



def load(path: str | None = None):
    """dlopen the backend and bind every entry point of include/gar_hip.h."""
    path = os.path.abspath(path or DEFAULT_PATH)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise GarLibraryError(
            f"{path} not found: build the HIP backend first "
            "(python -c 'import __graft_entry__ as g; g.build()' or "
            "make -C aligator_amd/csrc). There is no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
