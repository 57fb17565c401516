"""Loader and ctypes prototypes for libvarpro_hip.so (the C ABI of include/varpro_hip.h).

There is deliberately NO fallback: if the HIP library is missing or cannot be loaded the
import of any compute entry point fails loudly (``VarproHipUnavailable``).
"""
import ctypes as C
import os

# torch ships its own libamdhip64.so (same soname as /opt/rocm's).  Importing torch first makes
# the dynamic linker resolve our library's libamdhip64.so.7 dependency to the copy torch already
# loaded, so that torch tensors and our kernels share ONE HIP runtime in the process.
try:  # pragma: no cover - depends on environment
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

VP_MAX_BASIS = 8
VP_MAX_PARAMS = 8
VP_MAX_BASIS_PARAMS = 2
VP_MAX_PAIRS = 16

VP_F64, VP_F32 = 0, 1
VP_FLAG_DEVICE_PTRS, VP_FLAG_T_PER_PROBLEM, VP_FLAG_W_PER_PROBLEM, VP_FLAG_OWN_STREAM = 1, 2, 4, 8
VP_FLAG_NO_GRID_RECURRENCE = 16
VP_FLAG_STREAM_ROWS = 32
VP_BASIS_SKIP_INVARIANT = 1
VP_FIT_DERIVATIVES_ON_ACCEPT = 1
VP_WANT_BASIS, VP_WANT_DERIVATIVES = 1, 2
VP_KERNEL_EVALUATE, VP_KERNEL_BASIS, VP_KERNEL_FIT = 0, 1, 2
VP_ST_OK, VP_ST_NONFINITE, VP_ST_NOT_EVALUATED = 0, 1, 2

VP_ERR_OK, VP_ERR_INVALID, VP_ERR_UNSUPPORTED, VP_ERR_HIP, VP_ERR_NO_DEVICE = 0, -1, -2, -3, -4

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvarpro_hip.so")
if os.environ.get("VARPRO_HIP_LIBRARY"):  # developer A/B builds of the same library (tools/); still no fallback
    LIB_PATH = os.path.abspath(os.environ["VARPRO_HIP_LIBRARY"])

# vp_allreduce_fn: int (*)(void *dev_doubles, int64_t count, void *hip_stream, void *user)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)

# every symbol include/varpro_hip.h declares (tests check the library exports all of them)
VP_FIT_KERNEL_AUTO, VP_FIT_KERNEL_WAVE, VP_FIT_KERNEL_SLOTS = 0, 1, 2

ABI_SYMBOLS = [
    "vp_batch_create", "vp_batch_destroy", "vp_set_params", "vp_params", "vp_residuals", "vp_jacobian",
    "vp_linear_coeffs", "vp_weighted_data", "vp_set_observations", "vp_cost", "vp_evaluate", "vp_basis", "vp_lm_opts_default", "vp_fit", "vp_fit_trace",
    "vp_best_fit", "vp_statistics", "vp_summary", "vp_summary_device", "vp_global_fit_condition", "vp_set_rhs_allreduce", "vp_set_fit_kernel", "vp_set_timing", "vp_last_kernel_ms", "vp_synchronize", "vp_last_error",
    "vp_last_error_detail", "vp_version", "vp_device_count",
    "vp_batch_create_external", "vp_set_params_with_basis", "vp_jacobian_with_derivatives", "vp_evaluate_with_basis",
    "vp_reduce_cost", "vp_fit_begin", "vp_fit_step_with_basis", "vp_fit_end", "vp_fit_active_set",
]


# test hooks (include/varpro_hip_debug.h): exported by the library, not part of the drop-in boundary
DEBUG_SYMBOLS = ["vp_debug_gram_evaluate", "vp_debug_lmpar_gram", "vp_debug_set_refit"]


class VarproHipUnavailable(ImportError):
    pass


class VarproHipError(RuntimeError):
    def __init__(self, code, message, detail=0):
        super().__init__("varpro_hip error %d: %s" % (code, message))
        self.code = code
        self.detail = detail


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_basis", C.c_int32),
        ("n_params", C.c_int32),
        ("kind", C.c_int32 * VP_MAX_BASIS),
        ("param", (C.c_int32 * VP_MAX_BASIS_PARAMS) * VP_MAX_BASIS),
    ]


class LmOpts(C.Structure):
    _fields_ = [
        ("ftol", C.c_double),
        ("xtol", C.c_double),
        ("gtol", C.c_double),
        ("stepbound", C.c_double),
        ("patience", C.c_int32),
        ("scale_diag", C.c_int32),
    ]


class Report(C.Structure):
    _fields_ = [("termination", C.c_int32), ("n_evals", C.c_int32), ("objective", C.c_double)]


_lib = None


def load():
    """Load libvarpro_hip.so; raises VarproHipUnavailable (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VarproHipUnavailable(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C varpro_amd/csrc`). varpro_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise VarproHipUnavailable("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32p, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.vp_batch_create.argtypes = [C.POINTER(vp), C.POINTER(ModelDesc), C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                    vp, vp, vp, C.c_double, C.c_int, C.c_int, vp]
    lib.vp_batch_create_external.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, i32p, i32p, C.c_int, C.c_int64,
                                             C.c_int64, C.c_int64, vp, vp, C.c_double, C.c_int, C.c_int, vp]
    lib.vp_set_params_with_basis.argtypes = [vp, vp, vp, vp]
    lib.vp_jacobian_with_derivatives.argtypes = [vp, vp, vp, vp]
    lib.vp_evaluate_with_basis.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.vp_fit_begin.argtypes = [vp, C.POINTER(LmOpts), vp, C.c_int]
    lib.vp_fit_step_with_basis.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int64)]
    lib.vp_fit_end.argtypes = [vp, vp, vp, vp]
    lib.vp_fit_active_set.argtypes = [vp, vp, vp]
    lib.vp_batch_destroy.argtypes = [vp]
    lib.vp_batch_destroy.restype = None
    lib.vp_set_params.argtypes = [vp, vp]
    lib.vp_set_observations.argtypes = [vp, vp]
    lib.vp_params.argtypes = [vp, vp]
    lib.vp_residuals.argtypes = [vp, vp, vp]
    lib.vp_jacobian.argtypes = [vp, vp, vp]
    lib.vp_linear_coeffs.argtypes = [vp, vp, vp]
    lib.vp_weighted_data.argtypes = [vp, vp]
    lib.vp_cost.argtypes = [vp, vp]
    lib.vp_evaluate.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.vp_basis.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.vp_lm_opts_default.argtypes = [C.POINTER(LmOpts), C.c_int]
    lib.vp_lm_opts_default.restype = None
    lib.vp_fit.argtypes = [vp, C.POINTER(LmOpts), vp, vp, vp]
    lib.vp_fit_trace.argtypes = [vp, C.POINTER(LmOpts), vp, vp, vp, vp, C.c_int]
    lib.vp_best_fit.argtypes = [vp, vp]
    lib.vp_debug_gram_evaluate.argtypes = [vp, vp, vp]
    lib.vp_debug_lmpar_gram.argtypes = [C.c_int64, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.vp_debug_set_refit.argtypes = [vp, C.c_int]
    lib.vp_statistics.argtypes = [vp, vp, vp, vp, vp]
    lib.vp_summary.argtypes = [vp, dp]
    lib.vp_summary_device.argtypes = [vp, vp]
    lib.vp_global_fit_condition.argtypes = [vp, vp]
    lib.vp_reduce_cost.argtypes = [vp, vp, dp]
    lib.vp_set_rhs_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, C.c_int64]
    lib.vp_set_fit_kernel.argtypes = [vp, C.c_int]
    lib.vp_set_timing.argtypes = [vp, C.c_int]
    lib.vp_last_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    lib.vp_synchronize.argtypes = [vp]
    lib.vp_last_error.restype = C.c_char_p
    lib.vp_last_error_detail.restype = C.c_int
    lib.vp_version.restype = C.c_char_p
    lib.vp_device_count.restype = C.c_int
    _ = i32p
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        lib = load()
        raise VarproHipError(rc, lib.vp_last_error().decode("utf-8", "replace"), lib.vp_last_error_detail())


def device_count():
    return load().vp_device_count()
