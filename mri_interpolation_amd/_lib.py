"""ctypes binding of include/mri_inr.h (the C-ABI drop-in boundary).

The library is required: there is no CPU or PyTorch fallback for the hot path.  If
`libmri_inr.so` is missing or fails to load, every op raises.
"""
import ctypes as C
import os

MAX_LEVELS = 32
MAX_DIM = 7

ACT_IDENTITY, ACT_RELU, ACT_SINE, ACT_GELU = 0, 1, 2, 3
DERIV_NONE, DERIV_MUL, DERIV_RELU_MASK = 0, 1, 2
BWD_PREPARED = 16
BWD_OVERWRITE = 32
ERR_UNSUPPORTED = -2

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmri_inr.so")
if os.environ.get("MRI_LIB"):  # tuning: an alternative build of the same sources (tools/)
    _LIB_PATH = os.environ["MRI_LIB"]


class GridDesc(C.Structure):
    """struct mri_grid_desc (host memory)."""
    _fields_ = [
        ("dim", C.c_int32),
        ("n_levels", C.c_int32),
        ("n_features", C.c_int32),
        ("reserved", C.c_int32),
        ("resolution", (C.c_float * (MAX_DIM + 1)) * MAX_LEVELS),
        ("table_size", C.c_uint32 * MAX_LEVELS),
        ("table_offset", C.c_uint64 * MAX_LEVELS),
    ]


class FusedStepArgs(C.Structure):
    """struct mri_fused_step_args (host memory; every pointer a device pointer unless noted)."""
    _fields_ = (
        [("grid", C.POINTER(GridDesc)), ("table", C.c_void_p)]
        + [(k, C.c_void_p) for k in ("w1", "b1", "w2", "b2", "w3", "b3")]
        + [(k, C.c_void_p) for k in ("d_table", "d_w1", "d_b1", "d_w2", "d_b2", "d_w3", "d_b3", "loss")]
        + [(k, C.c_int32) for k in ("hidden", "bwd_method", "counted", "join_pending")]
        + [("coords", C.c_void_p), ("target", C.c_void_p), ("n", C.c_int64), ("enc", C.c_void_p),
           ("d_enc", C.c_void_p), ("tiny_ws", C.c_void_p), ("tiny_ws_bytes", C.c_int64),
           ("bwd_ws", C.c_void_p), ("bwd_ws_bytes", C.c_int64), ("absmax", C.c_void_p)]
        + [(k, C.c_void_p) for k in ("param", "grad", "exp_avg", "exp_avg_sq")]
        + [("n_params", C.c_int64)] + [(k, C.c_double) for k in ("lr", "beta1", "beta2", "eps")]
        + [("step", C.c_int32), ("grad_scale", C.c_float), ("next_idx", C.c_void_p), ("next_coords", C.c_void_p),
           ("next_target", C.c_void_p), ("next_n", C.c_int64), ("next_bwd_ws", C.c_void_p),
           ("next_bwd_ws_bytes", C.c_int64), ("next_absmax", C.c_void_p), ("seed", C.c_uint64),
           ("first", C.c_int64), ("lo", C.c_int64), ("hi", C.c_int64), ("dim", C.c_int32), ("reserved", C.c_int32),
           ("shape", C.c_int64 * MAX_DIM), ("axis_offset", C.c_int64 * MAX_DIM), ("axes", C.c_void_p),
           ("volume", C.c_void_p)]
        + [(k, C.c_void_p) for k in ("stream", "stream_side", "ev_fork", "ev_join")]
        + [("ev_phase", C.c_void_p * 5), ("grad_divisor", C.c_float), ("reserved2", C.c_float)])


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_F = C.c_float
_D = C.c_double

# name -> argtypes; every function returns int except the two string getters
SIGNATURES = {
    "mri_set_option": [C.c_char_p, _I32],
    "mri_hashgrid_forward": [C.POINTER(GridDesc), _P, _I64, _P, _P, _I64, _I64, _I64, _P],
    "mri_hashgrid_backward": [C.POINTER(GridDesc), _P, _P, _I64, _I64, _I64, _I64, _P, _I32, _P,
                              _I64, _P],
    "mri_hashgrid_backward_levels": [C.POINTER(GridDesc), _P, _P, _I64, _I64, _I64, _I64, _P, _I32,
                                     C.c_uint32, _P, _I64, _P],
    "mri_hashgrid_backward_prepare": [C.POINTER(GridDesc), _P, _I64, _I32, _P, _I64, _P],
    "mri_hashgrid_backward_adam": [C.POINTER(GridDesc), _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _D, _D,
                                   _D, _D, _I32, _F, _I32, _P, _I64, _P],
    "mri_linear_forward": [_P, _I64, _I64, _P, _P, _I64, _I32, _I32, _I32, _F, _P, _I64, _P,
                           _I64, _P],
    "mri_linear_backward_data": [_P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I64, _P, _I64, _I64,
                                 _P],
    "mri_linear_backward_weight": [_P, _I64, _P, _I64, _I64, _I64, _I32, _I32, _P, _P, _P],
    "mri_apply_deriv": [_P, _I64, _I32, _P, _I64, _I64, _I32, _P],
    "mri_frequency_forward": [_P, _I64, _I64, _I32, _I32, _P, _I64, _P],
    "mri_frequency_backward": [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _I64, _P],
    "mri_mse_loss": [_P, _P, _I64, _F, _P, _P, _P],
    "mri_tiny_mlp_forward": [_P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P],
    "mri_tiny_mlp_train_slice": [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _F, _P,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _P],
    "mri_tiny_mlp_train": [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                           _P, _P, _P, _P, _P, _P, _I64, _P],
    "mri_tiny_mlp_train_overwrite": [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _F, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "mri_tiny_mlp_train_dx_absmax": [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _I32, _P, _P, _I64, _P],
    "mri_hashgrid_backward_scaled": [C.POINTER(GridDesc), _P, _P, _I64, _I64, _I64, _I64, _P, _I32,
                                     C.c_uint32, _P, _P, _I64, _P],
    "mri_hash_tiny_mlp_train": [C.POINTER(GridDesc), _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _F,
                                _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I32, _P, _I64, _P],
    "mri_hashgrid_backward_input": [C.POINTER(GridDesc), _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P],
    "mri_hashgrid_forward_signal": [C.POINTER(GridDesc), _P, _I64, _P, _P, _I64, _I64, _P, _P],
    "mri_tiny_mlp_train_overlapped": [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _F, _P, _P,
                                      _P, _P, _P, _P, _P, _P, _I32, _P, C.c_uint64, _P, _P, _I64,
                                      _P],
    "mri_siren_forward": [_P, _I64, _I32, _I32, _I32, C.POINTER(_P), C.POINTER(_P), _F, _F,
                          C.POINTER(_P), C.POINTER(_P), _P, _P, _I64, _P],
    "mri_siren_backward": [_P, _P, _I64, _I32, _I32, _I32, C.POINTER(_P), C.POINTER(_P),
                           C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _I32, _P, _I64,
                           _P],
    "mri_siren_forward_loss": [_P, _P, _I64, _I64, _I32, _I32, _I32, C.POINTER(_P), C.POINTER(_P), _F,
                               _F, _F, C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P,
                               _I64, _P],
    "mri_adam_step": [_P, _P, _P, _P, _I64, _D, _D, _D, _D, _I32, _F, _P],
    "mri_sample_indices": [C.c_uint64, _I64, _I64, _I64, _I64, _P, _P],
    "mri_fused_step": [_P],
    "mri_gather_batch": [_P, _I64, _I32, C.POINTER(_I64), _P, C.POINTER(_I64), _P, _P, _P, _P],
}
STRING_GETTERS = ["mri_version", "mri_last_error"]
INT64_GETTERS = {"mri_hashgrid_backward_workspace_bytes": [C.POINTER(GridDesc), _I64],
                 "mri_fused_step_args_bytes": [],
                 "mri_tiny_mlp_workspace_bytes": [_I32, _I32, _I64],
                 "mri_siren_backward_workspace_bytes": [_I64, _I32, _I32],
                 "mri_siren_forward_workspace_bytes": [_I32, _I32],
                 "mri_hashgrid_forward_signal_blocks": [C.POINTER(GridDesc), _I64],
                 "mri_tiny_mlp_round_rows": [_I32, _I32, _I64]}
INT_GETTERS = {"mri_tiny_mlp_supported": [_I32, _I32, _I32],
               "mri_hash_tiny_mlp_supported": [C.POINTER(GridDesc), _I32],
               "mri_tiny_mlp_dx_absmax_supported": [_I32, _I32],
               "mri_siren_supported": [_I32, _I32, _I32, _I32]}  # return a plain value, not a status

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libmri_inr.so once; raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be in the process BEFORE this library is
    # dlopen'ed, otherwise a second HIP runtime (the system one) gets loaded and kernels are
    # launched on a runtime that owns no device context.
    import torch  # noqa: F401
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build it with `python -m mri_interpolation_amd.build` "
            "(hipcc, gfx950).  The MI355X hot path has no CPU/PyTorch fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name in STRING_GETTERS:
        getattr(lib, name).restype = C.c_char_p
        getattr(lib, name).argtypes = []
    for name, argtypes in INT64_GETTERS.items():
        getattr(lib, name).restype = C.c_int64
        getattr(lib, name).argtypes = argtypes
    for name, argtypes in INT_GETTERS.items():
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mri_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libmri_inr {what} failed ({rc}): {msg}")


def call(name: str, *args):
    check(getattr(load(), name)(*args), name)


def version() -> str:
    return load().mri_version().decode()


def set_option(name: str, value: int):
    call("mri_set_option", name.encode(), int(value))
