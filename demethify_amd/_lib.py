"""ctypes binding of libdemethify_hip.so (C-ABI declared in include/demethify_hip.h).

The product path has no CPU fallback: if the shared library is missing or no gfx950 device
is visible, every solver call raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libdemethify_hip.so"

DMF_OK = 0
DMF_PTR_DEVICE = 1
DMF_INIT_IN_UNIT_RANGE = 4
DMF_SELECT_COUNTS_F32_EXACT, DMF_SELECT_PURITY, DMF_SELECT_ALPHA_OUTSIDE_UNIT, DMF_SELECT_V_UNALIGNED = 1, 2, 4, 8
DMF_COUNTS_F64 = 2
DMF_MODE_PARTIAL = 0
DMF_MODE_UNSUPERVISED = 1
MAX_K = 64  # dmf::kMaxK: largest n_c + n_u the kernels are built for (DMF_ERR_UNSUPPORTED beyond)
KERNEL_ROWPASS, KERNEL_GRAM, KERNEL_ALPHA, KERNEL_COST = 0, 1, 2, 3
KERNEL_FAMILIES = ("rowpass", "gram", "alpha", "cost")

_p = C.c_void_p
_i64 = C.c_int64
_dbl_p = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/demethify_hip.h declares
SIGNATURES = {
    "dmf_status_string": (C.c_char_p, [C.c_int]),
    "dmf_last_error": (C.c_char_p, []),
    "dmf_abi_version": (C.c_int, []),
    "dmf_context_create": (C.c_int, [C.c_int, _p, C.POINTER(_p)]),
    "dmf_context_destroy": (C.c_int, [_p]),
    "dmf_context_synchronize": (C.c_int, [_p]),
    "dmf_context_set_profiling": (C.c_int, [_p, C.c_int]),
    "dmf_context_kernel_time": (C.c_int, [_p, C.c_int, _dbl_p, C.POINTER(_i64)]),
    "dmf_context_reset_kernel_time": (C.c_int, [_p]),
    "dmf_context_set_generic": (C.c_int, [_p, C.c_int]),
    "dmf_context_set_stop_confirmation": (C.c_int, [_p, C.c_int]),
    "dmf_problem_create": (C.c_int, [_p, _i64, _i64, _i64, _p, _p, _p, C.c_int, C.POINTER(_p)]),
    "dmf_problem_gather": (C.c_int, [_p, _p, _p, _i64, C.POINTER(_p)]),
    "dmf_problem_gather_device": (C.c_int, [_p, _p, _p, _i64, C.POINTER(_p)]),
    "dmf_problem_destroy": (C.c_int, [_p]),
    "dmf_problem_shape": (C.c_int, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "dmf_cost": (C.c_int, [_p, _p, _p, _i64, _p, C.c_int, _dbl_p]),
    "dmf_project_simplex": (C.c_int, [_p, _p, _i64, _i64, C.c_double, C.c_int, _p]),
    "dmf_update_u": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, C.c_int, C.c_int, _dbl_p, _p, _p]),
    "dmf_update_alpha": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, C.c_int, _dbl_p, _p, _p]),
    "dmf_percentile_axis0": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, C.c_int, _p]),
    "dmf_solver_create": (C.c_int, [_p, _p, _p, _p, _i64, C.c_int, C.c_int, C.POINTER(_p)]),
    "dmf_solver_set_purity": (C.c_int, [_p, _p, C.c_int]),
    "dmf_solver_step": (C.c_int, [_p, _i64, _i64, C.c_double, C.POINTER(_i64), C.POINTER(C.c_int)]),
    "dmf_solver_get": (C.c_int, [_p, C.c_int, _p, _p, _dbl_p, C.POINTER(_i64)]),
    "dmf_solver_cost": (C.c_int, [_p, _dbl_p]),
    "dmf_solver_cost_begin": (C.c_int, [_p]),
    "dmf_solver_cost_end": (C.c_int, [_p, _dbl_p]),
    "dmf_solver_destroy": (C.c_int, [_p]),
    "dmf_solver_describe": (C.c_int, [_p, _i64, C.c_char_p, _i64]),
    "dmf_select_describe": (C.c_int, [_i64, _i64, _i64, _i64, C.c_int, C.c_int, _i64, C.c_int, C.c_char_p, _i64]),
    "dmf_solver_stop_info": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(_i64), C.POINTER(_i64), _dbl_p]),
    "dmf_table_scan": (C.c_int, [C.c_char_p, C.c_char, C.POINTER(_i64), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]),
    "dmf_table_read": (C.c_int, [C.c_char_p, C.c_char, C.c_int, C.c_int, _i64, _p, _i64, C.c_double, _p, _i64, C.c_int]),
    "dmf_host_alloc": (_p, [C.c_size_t, C.POINTER(C.c_int)]),
    "dmf_host_free": (None, [_p, C.c_int]),
    "dmf_write_interval_csv": (C.c_int, [C.c_char_p, C.c_char_p, _p, _p, _i64, C.c_int, C.c_int, C.c_int]),
    "dmf_stage_upload": (C.c_int, [_p, _p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dmf_stage_free": (C.c_int, [_p, _p]),
    "dmf_solve": (C.c_int, [_p, _p, _p, _p, _i64, C.c_int, _i64, _i64, C.c_double, C.c_int, _p, _p,
                            _dbl_p, C.POINTER(_i64)]),
}

_lib = None


class DemethifyHipError(RuntimeError):
    """A C-ABI call returned a non-zero dmf_status."""

    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        msg = f"{where}: status {status}"
        if detail:
            msg += f" ({detail})"
        super().__init__(msg)


def _preload_torch_hip_runtime():
    import importlib.util
    import os

    if os.environ.get("DEMETHIFY_SYSTEM_HIP") == "1":  # (opt out: use the runtime this library was linked against)
        return
    try:
        spec = importlib.util.find_spec("torch")  # (does not import torch)
        if spec is None or not spec.submodule_search_locations:
            return
        cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except Exception:  # pragma: no cover - no torch, or a wheel without the bundled runtime: nothing to align
        pass


def load():
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m demethify_amd._build` "
            "(there is no CPU fallback)")
    # If PyTorch-ROCm lives in this process (bench.py, the multi-GPU drivers), let it bring up its HIP
    # runtime first: torch's wheel bundles its own libamdhip64, and initialising ours before it leaves torch
    # without a device ("no ROCm-capable device is detected").
    import sys

    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # pragma: no cover - torch without a usable GPU: our own checks will report it
            pass
    else:
        # torch is installed but not imported yet (the CLI): a later `import torch` -- the bootstrap's replicate stack,
        # torch.distributed -- would bring a SECOND HIP runtime into the process and find no GPU with it (measured: two
        # libamdhip64 in /proc/self/maps, torch.cuda.is_available() False).  Loading torch's copy of the runtime first
        # makes it the one this library binds to as well -- the configuration bench.py and the test-suite run in --
        # without paying for `import torch` (1.5 s) in runs that never need it.
        _preload_torch_hip_runtime()
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, where: str):
    if status != DMF_OK:
        lib = load()
        text = lib.dmf_status_string(status).decode()
        err = lib.dmf_last_error().decode()
        raise DemethifyHipError(status, where, f"{text}; {err}" if err else text)
