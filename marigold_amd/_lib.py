"""ctypes binding of libmarigold_hip.so (C ABI in include/marigold_hip.h).

The product path has NO fallback: if the HIP library is missing or does not export the ABI
this module raises, loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``make -C marigold_amd/csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARIGOLD_HIP_LIB") or os.path.join(_HERE, "libmarigold_hip.so")   # (override: same-box A/B of two builds)
ABI_VERSION = 4

# enum mg_op_kind
OP_IGEMM, OP_GN_STATS, OP_GN_FINALIZE, OP_GN_APPLY = 1, 2, 3, 4
OP_FLASH_ATTN64, OP_SOFTMAX_ROWS = 6, 7
OP_GN_SLAB = 9
OP_ROWGEMM = 10
OP_FLASH_ATTN512 = 11
RG_BF16, RG_GEGLU, RG_QKV, RG_XATTN = 0, 1, 2, 3
OP_SCHED_STEP = 12
OP_LINEAR_SMALL_M, OP_LATENT_1X1, OP_POST_NCHW, OP_IM2COL_SMALL = 13, 14, 15, 16
OP_CONV3X3 = 17
OP_CONV3X3_HEAD = 18
OP_ENS_DEPTH_STATS, OP_ENS_DEPTH_MEDIAN, OP_ENS_DEPTH_NORM, OP_ENS_NORMALS = 20, 21, 22, 23
OP_RESIZE = 24
OP_COLORIZE = 25
OP_MEMSET, OP_COPY = 30, 31
EPI_BF16, EPI_GEGLU, EPI_F32, EPI_SOFTMAX2, EPI_XATTN2 = 0, 1, 2, 3, 4
POST_NONE, POST_DEPTH, POST_NORMALS, POST_UNIT, POST_SCHED = 0, 1, 2, 3, 4

OP_NAMES = {v: k[3:].lower() for k, v in list(globals().items()) if k.startswith("OP_")}

EXPORTS = [
    "mg_abi_version", "mg_operand_bits", "mg_last_error", "mg_init", "mg_geglu_interleave", "mg_device_info", "mg_launch",
    "mg_program_create", "mg_program_num_ops", "mg_program_run", "mg_program_validate", "mg_program_run_range",
    "mg_program_capture", "mg_program_profile", "mg_program_destroy", "mg_conv2d_igemm", "mg_conv3x3", "mg_conv3x3_gn_slots", "mg_flash4w_plan_test",
    "mg_sched_step", "mg_ensemble_normals", "mg_ens_align_cost_grad", "mg_bfgs_minimize", "mg_ens_align_minimize", "mg_event_create", "mg_event_record",
    "mg_event_elapsed_ms", "mg_event_destroy", "mg_clock_probe", "mg_debug_read_workspace",
    "mg_model_load", "mg_model_destroy", "mg_model_info", "mg_model_device_bytes", "mg_model_validate", "mg_model_vae_encode",
    "mg_model_denoise", "mg_model_vae_decode", "mg_ensemble_depth",
]


class MgOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("i", ctypes.c_int32 * 40), ("f", ctypes.c_float * 8),
                ("p", ctypes.c_void_p * 16), ("l", ctypes.c_int64 * 4)]


class MarigoldHipError(RuntimeError):
    pass


# The fp16-operand twin (csrc/Makefile: OPERAND_F16=1, same sources, same ABI): what the engine modules load for
# torch_dtype=torch.float16 (the reference's --fp16, script/depth/run.py:203-211).  Its own globals, its own mg_init.
LIB_PATH_F16 = os.environ.get("MARIGOLD_HIP_LIB_F16") or os.path.join(_HERE, "libmarigold_hip_f16.so")

_libs = {}


def load(f16=False):
    """Load the shared library (no GPU needed) and check that every ABI symbol is exported.  ``f16``: the fp16-operand build."""
    key = bool(f16)
    if key in _libs:
        return _libs[key]
    path = LIB_PATH_F16 if key else LIB_PATH
    if not os.path.exists(path):
        raise MarigoldHipError(
            f"{path} not found: the HIP engine is not built. Run __graft_entry__.build() "
            f"(make -C marigold_amd/csrc). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise MarigoldHipError(f"{path} lacks ABI symbols: {missing}")
    lib.mg_last_error.restype = ctypes.c_char_p
    lib.mg_launch.argtypes = [ctypes.POINTER(MgOp), ctypes.c_void_p]
    lib.mg_geglu_interleave.restype = ctypes.c_int
    lib.mg_conv2d_igemm.argtypes = [ctypes.POINTER(MgOp), ctypes.c_void_p]
    lib.mg_conv3x3.argtypes = [ctypes.POINTER(MgOp), ctypes.c_void_p]
    lib.mg_conv3x3_gn_slots.argtypes = [ctypes.POINTER(MgOp)]
    lib.mg_conv3x3_gn_slots.restype = ctypes.c_int
    lib.mg_flash4w_plan_test.argtypes = [ctypes.c_int] * 4 + [ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)]
    lib.mg_program_create.restype = ctypes.c_void_p
    lib.mg_program_create.argtypes = [ctypes.POINTER(MgOp), ctypes.c_int]
    lib.mg_program_num_ops.argtypes = [ctypes.c_void_p]
    lib.mg_ens_align_cost_grad.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7
    lib.mg_bfgs_minimize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mg_ens_align_minimize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mg_program_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.mg_program_validate.argtypes = [ctypes.c_void_p]
    lib.mg_program_run_range.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.mg_program_capture.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.mg_program_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.mg_program_destroy.argtypes = [ctypes.c_void_p]
    lib.mg_program_destroy.restype = None
    lib.mg_sched_step.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_float] * 3 + [ctypes.c_void_p]
    lib.mg_ensemble_normals.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    lib.mg_device_info.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, ctypes.c_int]
    lib.mg_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.mg_debug_read_workspace.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.mg_model_load.restype = ctypes.c_void_p
    lib.mg_model_load.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.mg_model_destroy.argtypes = [ctypes.c_void_p]
    lib.mg_model_destroy.restype = None
    lib.mg_model_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.mg_model_device_bytes.argtypes = [ctypes.c_void_p]
    lib.mg_model_device_bytes.restype = ctypes.c_longlong
    lib.mg_model_validate.argtypes = [ctypes.c_void_p]
    lib.mg_model_vae_encode.argtypes = [ctypes.c_void_p] * 4
    lib.mg_model_denoise.argtypes = [ctypes.c_void_p] * 5
    lib.mg_model_vae_decode.argtypes = [ctypes.c_void_p] * 4
    lib.mg_ensemble_depth.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mg_event_create.restype = ctypes.c_void_p
    lib.mg_event_record.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.mg_event_elapsed_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.mg_event_destroy.argtypes = [ctypes.c_void_p]
    lib.mg_event_destroy.restype = None
    lib.mg_operand_bits.restype = ctypes.c_int
    if lib.mg_abi_version() != ABI_VERSION:
        raise MarigoldHipError(f"ABI version mismatch: library {lib.mg_abi_version()}, binding {ABI_VERSION}")
    if bool(lib.mg_operand_bits() & 1) != key:
        raise MarigoldHipError(f"{path} is the {'fp16' if lib.mg_operand_bits() & 1 else 'bf16'}-operand build")
    lib._mg_f16 = key
    _libs[key] = lib
    return lib


def check(rc, what="libmarigold_hip", lib=None):
    if rc != 0:
        msgs = [(lib or l_).mg_last_error().decode(errors="replace") for l_ in ([lib] if lib is not None else list(_libs.values()) or [load()])]
        raise MarigoldHipError(f"{what}: {' | '.join(m for m in msgs if m) or 'error'}")


_inited = set()


def init(device_index=0, f16=False):
    """Bind the library to a GPU (one process drives one GPU)."""
    lib = load(f16)
    if (device_index, bool(f16)) not in _inited:
        check(lib.mg_init(int(device_index)), "mg_init", lib)
        _inited.add((device_index, bool(f16)))
    return lib
