"""ctypes binding of libnsd_hip.so (C ABI: include/nsd.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If the shared object is
missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("NSD_LIB", "libnsd_hip.so"))   # NSD_LIB=libnsd_hip_prof.so: diagnostic build
CSRC = os.path.join(_HERE, "csrc")

NSD_FLAG_RESIDUAL = 1
NSD_FLAG_TRAIN = 2
NSD_FLAG_BF16 = 4
NSD_FLAG_BIDIR = 8
# honoured by the DIAGNOSTIC build only (csrc/nsd_diag.h, libnsd_hip_diag.so); the product library rejects them
NSD_DIAG_FLAG_NO_L2_EXCHANGE = 16
NSD_DIAG_FLAG_SPREAD_GROUPS = 32
NSD_DIAG_FLAG_NO_FUSED_LAYERS = 64
NSD_DIAG_FLAG_LOSE_MEMBER = 128


class Rng(C.Structure):
    """nsd_rng of include/nsd.h"""
    _fields_ = [("seed", C.c_uint64), ("base_stream", C.c_uint32), ("p_lstm", C.c_float), ("p_head", C.c_float)]


class NsdError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "T", "C", "H", "L", "K", "F")]


class WsLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("hseq", "cseq", "gact", "inseq", "top", "alpha", "pooled", "fc0_pre",
                                          "dscore", "dpooled", "loss", "adpack", "slabs", "n_slabs", "hslabs", "da_seq", "din", "total")]


# every symbol include/nsd.h declares: name -> (restype, argtypes)
_fp, _vp, _ip = C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers
_dp = C.POINTER(Dims)
SYMBOLS = {
    "nsd_version": (C.c_int, []),
    "nsd_last_error": (C.c_char_p, []),
    "nsd_param_count": (C.c_int64, [C.c_int32] * 5),
    "nsd_param_layout": (C.c_int, [C.c_int32] * 5 + [C.POINTER(C.c_int64)]),
    "nsd_workspace_bytes": (C.c_int64, [_dp, C.POINTER(WsLayout)]),
    "nsd_fast_path": (C.c_int, [_dp]),
    "nsd_zscore_fwd": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "nsd_infer_scratch_bytes": (C.c_int64, [_dp]),
    "nsd_infer": (C.c_int, [_dp, _fp, _fp, C.c_uint32, _fp, _fp, _vp, _vp]),
    "nsd_lstm_fwd": (C.c_int, [_dp, _fp, _fp, _fp, C.c_uint32, _fp, C.c_int64, _vp]),
    "nsd_head_fwd": (C.c_int, [_dp, _fp, _fp, _fp, _fp, C.c_int64, _fp, _fp, _vp]),
    "nsd_head_bwd": (C.c_int, [_dp, _fp, _fp, _fp, _fp, _fp, _ip, C.c_float, _fp, C.c_int64, _vp]),
    "nsd_head_train": (C.c_int, [_dp, _fp, _fp, _fp, _ip, C.c_float, _fp, C.c_int64, _fp, _vp]),
    "nsd_lstm_head_train": (C.c_int, [_dp, _fp, _fp, _fp, _fp, _fp, _ip, C.c_float, C.c_uint32, _fp, C.c_int64, _fp, _vp]),
    "nsd_rng_path": (C.c_int, [_dp]),
    "nsd_lstm_head_train_rng": (C.c_int, [_dp, _fp, _fp, _vp, _ip, C.c_float, C.c_uint32, _fp, C.c_int64, _fp, _vp]),
    "nsd_lstm_bwd_rng": (C.c_int, [_dp, _fp, _fp, _vp, C.c_uint32, _fp, C.c_int64, _vp]),
    "nsd_lstm_bwd": (C.c_int, [_dp, _fp, _fp, _fp, C.c_uint32, _fp, C.c_int64, _fp, _vp]),
    "nsd_grad_reduce": (C.c_int, [_dp, _fp, C.c_int64, _fp, C.c_int32, _vp]),
    "nsd_grad_reduce_adam": (C.c_int, [_dp, _fp, C.c_int64, _fp, _fp, _fp, _fp] + [C.c_float] * 6 + [C.c_int32, _vp]),
    "nsd_loss_sum": (C.c_int, [_dp, _fp, C.c_int64, _fp, _vp]),
    "nsd_adam_step": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp] + [C.c_float] * 6 + [C.c_int32, _vp]),
    "nsd_adam_step_guarded": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp] + [C.c_float] * 6 + [C.c_int32, _fp, _vp]),
    "nsd_dropout_mask": (C.c_int, [C.c_uint64, C.c_uint32, C.c_float, C.c_int64, _fp, _vp]),
    "nsd_rrelu_noise": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int64, _fp, _vp]),
    "nsd_step_counter_inc": (C.c_int, [_vp, _vp]),
    "nsd_train_masks_dev": (C.c_int, [C.c_uint64, _vp, C.c_float, C.c_float, C.c_int64, _fp, C.c_int64, _fp, _fp, _vp]),
    "nsd_adam_step_dev": (C.c_int, [C.c_int64, _fp, _fp, _fp, _fp] + [C.c_float] * 6 + [_vp, _vp]),
    "nsd_gemm_bf16": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int64, C.c_int32, C.c_int64, _vp, C.c_int64, C.c_int32, _fp,
                                C.c_int32, C.c_int32, C.c_int64, C.c_int32, _vp]),
    "nsd_seq_param_count": (C.c_int64, [C.c_int32] * 6),
    "nsd_seq_param_layout": (C.c_int, [C.c_int32] * 6 + [C.POINTER(C.c_int64)]),
    "nsd_seq_supported": (C.c_int, [_dp, C.c_uint32]),
    "nsd_seq_workspace_bytes": (C.c_int64, [_dp, C.c_uint32]),
    "nsd_seq_infer": (C.c_int, [_dp, _fp, _fp, C.c_uint32, _fp, _fp, _vp, C.c_int64, _vp]),
    "nsd_seq_train_fwd": (C.c_int, [_dp, _fp, _fp, _vp, _ip, C.c_float, C.c_uint32, _vp, C.c_int64, _fp, _vp]),
    "nsd_seq_train_bwd": (C.c_int, [_dp, _fp, _vp, C.c_uint32, _vp, C.c_int64, _fp, _vp]),
    "nsd_seq_loss_sum": (C.c_int, [_dp, C.c_uint32, _vp, C.c_int64, _fp, _vp]),
    "nsd_seq_workspace_init": (C.c_int, [_vp, C.c_int64, _vp]),
    "nsd_seq_status": (C.c_int, [_vp, C.POINTER(C.c_int32), _vp]),
    "nsd_seq_guard": (C.c_int, [_vp, _fp, _vp]),
    "nsd_train_masks": (C.c_int, [C.c_uint64, C.c_uint32, C.c_float, C.c_float, C.c_int64, _fp, C.c_int64, _fp, _fp, _vp]),
}


# entry points of the diagnostic build only (csrc/nsd_diag.h)
DIAG_SYMBOLS = {
    "nsd_seq_profile": (C.c_int, [C.c_int32]),
    "nsd_seq_profile_read": (C.c_int, [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "nsd_diag_force_fwd48": (C.c_int, [C.c_int32]),
    "nsd_diag_force_bwd48": (C.c_int, [C.c_int32]),
}
DIAG_LIB_PATH = os.path.join(_HERE, "libnsd_hip_diag.so")


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with hipcc into libnsd_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "nsd.h"))
    outs = [LIB_PATH] + ([DIAG_LIB_PATH] if os.path.basename(LIB_PATH) == "libnsd_hip.so" else [])
    stale = any((not os.path.exists(o)) or any(os.path.getmtime(s) > os.path.getmtime(o) for s in srcs) for o in outs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load libnsd_hip.so; raise loudly when it is absent (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NsdError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        diag = os.path.basename(LIB_PATH) != "libnsd_hip.so"     # NSD_LIB: profiling / A-B builds, possibly of an older ABI
        for name, (res, args) in SYMBOLS.items():
            if diag and not hasattr(L, name):
                continue
            fn = getattr(L, name)        # AttributeError if the ABI and the header ever diverge
            fn.restype, fn.argtypes = res, args
        if diag and hasattr(L, "nsd_debug_profile_buffer"):      # diagnostic build only; not part of include/nsd.h
            L.nsd_debug_profile_buffer.restype, L.nsd_debug_profile_buffer.argtypes = C.c_int, [_vp]
        if L.nsd_version() < 300 and not diag:
            raise NsdError(f"{LIB_PATH} is ABI v{L.nsd_version()}, this binding needs >= 300: rebuild it")
        _lib = L
    return _lib


_diag = None
_diag_active = False


def diag_lib() -> C.CDLL:
    """libnsd_hip_diag.so: the same kernels, with the diagnostic flag bits and the per-kernel timing entry points of
    csrc/nsd_diag.h.  Test / bench / tools infrastructure -- the product modules never ask for it."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise NsdError(f"{DIAG_LIB_PATH} is missing: `make -C {CSRC}` builds it next to the product library")
        L = C.CDLL(DIAG_LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(DIAG_SYMBOLS.items()):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _diag = L
    return _diag


class diagnostic_library:
    """`with diagnostic_library():` -- every C-ABI call of the block goes to libnsd_hip_diag.so instead of the product
    library (same sources; both can be loaded at once).  Not re-entrant, not for product code."""

    def __enter__(self):
        global _lib, _diag_active
        if _diag_active:
            raise NsdError("diagnostic_library() is not re-entrant")
        lib()
        self._saved, _lib, _diag_active = _lib, diag_lib(), True
        return _lib

    def __exit__(self, *exc):
        global _lib, _diag_active
        _lib, _diag_active = self._saved, False
        return False


def diag_active() -> bool:
    return _diag_active


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NsdError(f"{what} failed (rc={rc}): {lib().nsd_last_error().decode(errors='replace')}")
