"""ctypes binding of libexpv_mi.so (the C ABI of include/expv_mi.h).

The library is the product; there is no CPU fallback.  Importing this module without the built
shared object raises, and creating a context without a GPU fails loudly (EXPV_MI_HIP_ERROR).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXPV_MI_LIB") or os.path.join(HERE, "libexpv_mi.so")   # EXPV_MI_LIB: A/B builds when profiling

F64, C64, F32, C32 = 0, 1, 2, 3
HOST, DEVICE = 0, 1
ORTHO_AUTO, ORTHO_MGS, ORTHO_LOWSYNC, ORTHO_PIPELINED = 0, 1, 2, 3
PATH_FLAGS = {"modular": 1, "two_kernel": 2, "pipeline": 4, "wave": 8, "overlapped": 16, "redo_serial": 32,
              "redo_wave_off": 64, "resident": 128, "patch": 256, "pipelined_lanczos": 512}

STATUS_NAMES = {
    0: "OK", 1: "DimensionMismatch", 2: "ArgumentError", 3: "AssertionError", 4: "SingularException",
    5: "Unsupported", 6: "OutOfMemory", 7: "HIPError", 8: "BoundsError",
}

KERNEL_IDS = {"firststep": 0, "matvec": 1, "dots": 2, "update": 3, "scale": 4, "combine": 5,
              "fused_a": 6, "fused_b": 7, "lincomb": 8, "aug": 9, "batch": 10}


class ArnoldiOpts(C.Structure):
    _fields_ = [("m", C.c_int32), ("iop", C.c_int32), ("init", C.c_int32), ("ishermitian", C.c_int32),
                ("ortho", C.c_int32), ("flags", C.c_int32), ("tol", C.c_double)]


class ExpvStats(C.Structure):
    _fields_ = [("m_used", C.c_int32), ("wasbreakdown", C.c_int32), ("matvecs", C.c_int32),
                ("path_flags", C.c_int32), ("beta", C.c_double)]


ABI_KINDS = {"ArnoldiOpts": 0, "ExpvStats": 1, "TimestepOpts": 2, "TimestepStats": 3, "KiopsOpts": 4}
PRINT_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
MATVEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


class TimestepOpts(C.Structure):
    _fields_ = [("tau", C.c_double), ("tol", C.c_double), ("delta", C.c_double), ("gamma", C.c_double),
                ("opnorm", C.c_double), ("has_opnorm", C.c_int32), ("m", C.c_int32), ("iop", C.c_int32),
                ("correct", C.c_int32), ("adaptive", C.c_int32), ("ishermitian", C.c_int32),
                ("verbose", C.c_int32), ("ortho", C.c_int32), ("no_basis_reuse", C.c_int32), ("reserved", C.c_int32),
                ("NA", C.c_int64), ("print", PRINT_FN),
                ("print_user", C.c_void_p)]


class TimestepStats(C.Structure):
    _fields_ = [("num_timesteps", C.c_int32), ("matvecs", C.c_int32), ("m_final", C.c_int32),
                ("arnoldi_calls", C.c_int32), ("arnoldi_reused", C.c_int32), ("stalled_steps", C.c_int32)]


class KiopsOpts(C.Structure):
    _fields_ = [("mmin", C.c_int32), ("mmax", C.c_int32), ("m", C.c_int32), ("iop", C.c_int32),
                ("ishermitian", C.c_int32), ("task1", C.c_int32), ("ortho", C.c_int32),
                ("reserved", C.c_int32), ("tol", C.c_double)]


# every symbol include/expv_mi.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pi, _pi64, _pd, _pvp = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_void_p)
PROTOTYPES = {
    "expv_mi_ctx_create": (_i, [_i, _vp, _pvp]),
    "expv_mi_ctx_destroy": (_i, [_vp]),
    "expv_mi_ctx_sync": (_i, [_vp]),
    "expv_mi_ctx_set_async_outputs": (_i, [_vp, _i]),
    "expv_mi_ctx_set_pipeline_overlap": (_i, [_vp, _i]),
    "expv_mi_ctx_counters": (_i, [_vp, _pi64]),
    "expv_mi_ctx_selftest": (_i, [_vp, _pi64]),
    "expv_mi_ctx_set_option": (_i, [_vp, C.c_char_p, _i64]),
    "expv_mi_ctx_get_option": (_i, [_vp, C.c_char_p, _pi64]),
    "expv_mi_last_error": (C.c_char_p, [_vp]),
    "expv_mi_version": (C.c_char_p, []),
    "expv_mi_malloc": (_i, [_vp, C.c_size_t, _pvp]),
    "expv_mi_free": (_i, [_vp, _vp]),
    "expv_mi_memcpy_h2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "expv_mi_memcpy_d2h": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "expv_mi_prof_enable": (_i, [_vp, _i]),
    "expv_mi_prof_reset": (_i, [_vp]),
    "expv_mi_prof_get": (_i, [_vp, _i, _pi64, _pd]),
    "expv_mi_prof_name": (C.c_char_p, [_i]),
    "expv_mi_op_create_csc": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _pvp]),
    "expv_mi_op_create_csr": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _i, _pvp]),
    "expv_mi_op_create_dense": (_i, [_vp, _i, _i64, _vp, _i64, _i, _pvp]),
    "expv_mi_op_create_callback": (_i, [_vp, _i, _i64, MATVEC_FN, _vp, _i, _i64, _pvp]),
    "expv_mi_op_destroy": (_i, [_vp]),
    "expv_mi_op_update_values": (_i, [_vp, _vp, _i]),
    "expv_mi_op_info": (_i, [_vp, _pi64, _pi64, _pi, _pd, _pi]),
    "expv_mi_op_reorder_info": (_i, [_vp, _vp]),
    "expv_mi_op_patch_info": (_i, [_vp, _vp]),
    "expv_mi_plan_cache": (_i, [_i, _i64, _vp]),
    "expv_mi_host_patch_order": (_i, [C.c_int64, _vp, _vp, _i, _vp, _vp, _vp]),
    "expv_mi_host_mesh_patch_order": (_i, [C.c_int64, _vp, _vp, _i, _vp, _vp, _vp]),
    "expv_mi_op_apply": (_i, [_vp, _vp, _i, _vp, _i]),
    "expv_mi_gemv_block": (_i, [_vp, _i, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i]),
    "expv_mi_ks_create": (_i, [_vp, _i, _i, _i64, _i, _i, _pvp]),
    "expv_mi_ks_destroy": (_i, [_vp]),
    "expv_mi_ks_resize": (_i, [_vp, _i]),
    "expv_mi_ks_get": (_i, [_vp, _pi, _pi, _pi, _pd, _pi]),
    "expv_mi_ks_set_m": (_i, [_vp, _i]),
    "expv_mi_ks_H": (_i, [_vp, _pvp, _pi, _pi, _pi]),
    "expv_mi_ks_V_download": (_i, [_vp, _i, _i, _vp, _i64]),
    "expv_mi_ks_V_upload": (_i, [_vp, _i, _i, _vp, _i64]),
    "expv_mi_ks_V_devptr": (_i, [_vp, _pvp, _pi64]),
    "expv_mi_arnoldi_opts_default": (None, [C.POINTER(ArnoldiOpts)]),
    "expv_mi_arnoldi": (_i, [_vp, _vp, _vp, _i, C.POINTER(ArnoldiOpts)]),
    "expv_mi_lanczos": (_i, [_vp, _vp, _vp, _i, C.POINTER(ArnoldiOpts)]),
    "expv_mi_arnoldi_aug": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _pd, _d, _d, C.POINTER(ArnoldiOpts)]),
    "expv_mi_expv_ks": (_i, [_vp, _d, _d, _vp, _i, _i]),
    "expv_mi_phiv_ks": (_i, [_vp, _d, _d, _i, _i, _vp, _i64, _i, _i, _pd]),
    "expv_mi_combine": (_i, [_vp, _i, _i, _vp, _i, _i, _d, _vp, _i64, _i, _i]),
    "expv_mi_expv": (_i, [_vp, _vp, _d, _d, _vp, _i, _vp, _i, _i, C.POINTER(ArnoldiOpts), C.POINTER(ExpvStats)]),
    "expv_mi_expv_error_estimate": (_i, [_vp, _vp, _d, _d, _vp, _i, _vp, _i, _d, _d, _i, _i]),
    "expv_mi_timestep_opts_default": (None, [C.POINTER(TimestepOpts)]),
    "expv_mi_timestep_caches_create": (_i, [_vp, _i, _i64, _i, _i, _pvp]),
    "expv_mi_timestep_caches_destroy": (_i, [_vp]),
    "expv_mi_phiv_timestep": (_i, [_vp, _vp, _i, _pd, _vp, _i64, _i, _i, _vp, _i64, _i, C.POINTER(TimestepOpts),
                                   _vp, C.POINTER(TimestepStats)]),
    "expv_mi_kiops_opts_default": (None, [C.POINTER(KiopsOpts)]),
    "expv_mi_kiops": (_i, [_vp, _vp, _pd, _i, _i, _vp, _i64, _i, _i, _vp, _i64, _i, C.POINTER(KiopsOpts), _pi64]),
    "expv_mi_expv_batch": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _i64, _i, _pd, _vp, _i64, _i, _vp, _i64, _i,
                                C.POINTER(ArnoldiOpts), C.POINTER(C.c_int32)]),
    "expv_mi_expv_batch_multi": (_i, [_pvp, _i, _i, _i64, _i, _vp, _vp, _vp, _i64, _pd, _vp, _i64, _vp, _i64, _i,
                                      C.POINTER(ArnoldiOpts), C.POINTER(C.c_int32)]),
    "expv_mi_rccl_available": (_i, []),
    "expv_mi_rccl_unique_id": (_i, [_vp]),
    "expv_mi_comm_create": (_i, [_vp, _vp, _i, _i, _pvp]),
    "expv_mi_gather_rccl": (_i, [_vp, _vp, _vp, _i64, _i]),
    "expv_mi_comm_destroy": (_i, [_vp]),
    "expv_mi_abi_sizeof": (C.c_size_t, [_i]),
    "expv_mi_abi_layout": (C.c_char_p, [_i]),
    "expv_mi_host_pattern_info": (_i, [C.c_int64, _vp, _vp, _i, _vp]),
    "expv_mi_host_rcm": (_i, [C.c_int64, _vp, _vp, _i, _vp, _vp]),
    "expv_mi_host_wrapsum": (_i, [_vp, C.c_uint64, _vp]),
    "expv_mi_host_expm": (_i, [_i, _i, _vp, _i]),
    "expv_mi_host_symtridiag_expcol": (_i, [_i, _pd, _pd, _d, _d, _pd]),
    "expv_mi_host_symtridiag_exp_last": (_i, [_i, _pd, _pd, _d, _d, _pd]),
    "expv_mi_host_phiv_dense": (_i, [_i, _i, _i, _vp, _i, _vp, _vp]),
}

_lib = None


class ExpvMIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code
        self.kind = STATUS_NAMES.get(code, str(code))


def load():
    """Load libexpv_mi.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    try:
        # torch bundles its own libamdhip64.so.7; when tensors are shared with this library the
        # runtime torch was built against must be the one in the process, so it has to load first.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)           # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, ctx=None):
    if code != 0:
        msg = load().expv_mi_last_error(ctx)
        raise ExpvMIError(code, msg.decode() if msg else "")
