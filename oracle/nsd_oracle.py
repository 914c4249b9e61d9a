"""ctypes binding of oracle/nsd_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (neural-speech-decoding_amd/) never does.

The oracle restates the reference's EEG_LSTM path
(Neuro-Alpha-App/Utilities/lstm_eeg_model.py:14-39,97 and
Neuro-Alpha-App/Frontend/app.py:166-170) on the CPU in plain C; it is pinned to
the reference by tests/golden/*.npz (see tools/make_goldens.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnsd_oracle.so")

PARAM_ORDER_TAIL = ["ln.weight", "ln.bias", "attn.weight", "attn.bias",
                    "fc.0.weight", "fc.0.bias", "fc.3.weight", "fc.3.bias"]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "nsd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libnsd_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.nsd_oracle_param_count.restype = C.c_long
        L.nsd_oracle_param_count.argtypes = [C.c_int] * 5
        L.nsd_oracle_layout.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_long)]
        L.nsd_oracle_rrelu_eval_slope.restype = C.c_float
        L.nsd_oracle_forward.argtypes = [C.c_int] * 7 + [fp] * 5 + [C.c_int] + [fp] * 10
        L.nsd_oracle_ce_loss.argtypes = [C.c_int, C.c_int, fp, ip, C.c_float, fp, fp]
        L.nsd_oracle_backward.argtypes = [C.c_int] * 7 + [fp] * 5 + [C.c_int] + [fp] * 9
        L.nsd_oracle_zscore.argtypes = [C.c_int] * 3 + [fp, fp]
        L.nsd_oracle_adam.argtypes = [C.c_long, fp, fp, fp, fp] + [C.c_float] * 5 + [C.c_int]
        L.nsd_oracle_rand_u32.restype = C.c_uint32
        L.nsd_oracle_rand_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        L.nsd_oracle_dropout_mask.argtypes = [C.c_uint64, C.c_uint32, C.c_float, C.c_long, fp]
        L.nsd_oracle_rrelu_noise.argtypes = [C.c_uint64, C.c_uint32, C.c_long, fp]
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(C.POINTER(C.c_float))


@dataclass(frozen=True)
class Dims:
    C: int = 8
    H: int = 48
    L: int = 2
    K: int = 3
    F: int = 32

    @property
    def tup(self):
        return (self.C, self.H, self.L, self.K, self.F)


def param_count(d: Dims) -> int:
    return int(lib().nsd_oracle_param_count(*d.tup))


def param_names(d: Dims):
    names = []
    for l in range(d.L):
        names += [f"lstm.weight_ih_l{l}", f"lstm.weight_hh_l{l}", f"lstm.bias_ih_l{l}", f"lstm.bias_hh_l{l}"]
    return names + PARAM_ORDER_TAIL


def param_shapes(d: Dims) -> Dict[str, tuple]:
    shp = {}
    for l in range(d.L):
        I = d.C if l == 0 else d.H
        shp[f"lstm.weight_ih_l{l}"] = (4 * d.H, I)
        shp[f"lstm.weight_hh_l{l}"] = (4 * d.H, d.H)
        shp[f"lstm.bias_ih_l{l}"] = (4 * d.H,)
        shp[f"lstm.bias_hh_l{l}"] = (4 * d.H,)
    shp.update({"ln.weight": (d.H,), "ln.bias": (d.H,), "attn.weight": (1, d.H), "attn.bias": (1,),
                "fc.0.weight": (d.F, d.H), "fc.0.bias": (d.F,), "fc.3.weight": (d.K, d.F), "fc.3.bias": (d.K,)})
    return shp


def layout(d: Dims) -> Dict[str, int]:
    n = 4 * d.L + 8
    offs = (C.c_long * n)()
    rc = lib().nsd_oracle_layout(*d.tup, offs)
    assert rc == 0
    return dict(zip(param_names(d), [int(o) for o in offs]))


def flatten_state(state: Dict[str, np.ndarray], d: Dims) -> np.ndarray:
    """state_dict (name -> array) -> flat fp32 vector in the canonical order."""
    flat = np.zeros(param_count(d), np.float32)
    lo, shp = layout(d), param_shapes(d)
    for k in param_names(d):
        a = np.asarray(state[k], np.float32)
        assert tuple(a.shape) == shp[k], (k, a.shape, shp[k])
        flat[lo[k]:lo[k] + a.size] = a.ravel()
    return flat


def unflatten(flat: np.ndarray, d: Dims) -> Dict[str, np.ndarray]:
    lo, shp = layout(d), param_shapes(d)
    return {k: flat[lo[k]:lo[k] + int(np.prod(shp[k]))].reshape(shp[k]).copy() for k in param_names(d)}


def forward(params: np.ndarray, x: np.ndarray, d: Dims, *, drop_lstm=None, rrelu_slope=None,
            drop_head=None, residual=False, saves=False) -> Dict[str, np.ndarray]:
    x = np.ascontiguousarray(x, np.float32)
    B, T, Cc = x.shape
    assert Cc == d.C and params.size == param_count(d)
    out = {"logits": np.zeros((B, d.K), np.float32), "probs": np.zeros((B, d.K), np.float32)}
    if saves:
        out.update(hseq=np.zeros((d.L, B, T, d.H), np.float32), cseq=np.zeros((d.L, B, T, d.H), np.float32),
                   gates=np.zeros((d.L, B, T, 4, d.H), np.float32), alpha=np.zeros((B, T), np.float32),
                   pooled=np.zeros((B, d.H), np.float32), ln_out=np.zeros((B, d.H), np.float32),
                   fc0_pre=np.zeros((B, d.F), np.float32), fc0_act=np.zeros((B, d.F), np.float32))
    g = out.get
    rc = lib().nsd_oracle_forward(B, T, *d.tup, _p(np.ascontiguousarray(params, np.float32)), _p(x),
                                  _p(drop_lstm), _p(rrelu_slope), _p(drop_head), int(residual),
                                  _p(out["logits"]), _p(out["probs"]), _p(g("hseq")), _p(g("cseq")),
                                  _p(g("gates")), _p(g("alpha")), _p(g("pooled")), _p(g("ln_out")),
                                  _p(g("fc0_pre")), _p(g("fc0_act")))
    if rc != 0:
        raise RuntimeError(f"nsd_oracle_forward rc={rc}")
    return out


def ce_loss(logits: np.ndarray, labels: np.ndarray, scale: Optional[float] = None):
    """returns (mean loss, dlogits) with dlogits = (softmax-onehot)*scale, scale default 1/B."""
    B, K = logits.shape
    scale = 1.0 / B if scale is None else scale
    lab = np.ascontiguousarray(labels, np.int32)
    loss = C.c_float(0)
    dl = np.zeros((B, K), np.float32)
    rc = lib().nsd_oracle_ce_loss(B, K, _p(np.ascontiguousarray(logits, np.float32)),
                                  lab.ctypes.data_as(C.POINTER(C.c_int32)), scale, C.byref(loss), _p(dl))
    if rc != 0:
        raise RuntimeError("label out of range")
    return loss.value / B, dl


def backward(params, x, d: Dims, fw: Dict[str, np.ndarray], dlogits, *, drop_lstm=None, rrelu_slope=None,
             drop_head=None, residual=False, want_dx=False):
    x = np.ascontiguousarray(x, np.float32)
    B, T, _ = x.shape
    grads = np.zeros(param_count(d), np.float32)
    dx = np.zeros_like(x) if want_dx else None
    rc = lib().nsd_oracle_backward(B, T, *d.tup, _p(np.ascontiguousarray(params, np.float32)), _p(x),
                                   _p(drop_lstm), _p(rrelu_slope), _p(drop_head), int(residual),
                                   _p(fw["hseq"]), _p(fw["cseq"]), _p(fw["gates"]), _p(fw["alpha"]),
                                   _p(fw["pooled"]), _p(fw["fc0_pre"]),
                                   _p(np.ascontiguousarray(dlogits, np.float32)), _p(grads), _p(dx))
    if rc != 0:
        raise RuntimeError(f"nsd_oracle_backward rc={rc}")
    return (grads, dx) if want_dx else grads


def loss_and_grads(params, x, labels, d: Dims, *, scale=None, **kw):
    """One CE training evaluation: returns (mean loss, flat grads, forward dict)."""
    fw = forward(params, x, d, saves=True, **kw)
    loss, dl = ce_loss(fw["logits"], labels, scale)
    return loss, backward(params, x, d, fw, dl, **kw), fw


def zscore(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    squeeze = x.ndim == 2
    x3 = x[None] if squeeze else x
    y = np.zeros_like(x3)
    lib().nsd_oracle_zscore(*x3.shape, _p(x3), _p(y))
    return y[0] if squeeze else y


def adam(p, g, m, v, *, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1):
    """in-place on p, m, v (fp32 contiguous)."""
    lib().nsd_oracle_adam(p.size, _p(p), _p(np.ascontiguousarray(g, np.float32)), _p(m), _p(v),
                          lr, beta1, beta2, eps, weight_decay, step)


def dropout_mask(seed: int, stream: int, p: float, shape) -> np.ndarray:
    m = np.zeros(shape, np.float32)
    lib().nsd_oracle_dropout_mask(seed, stream, p, m.size, _p(m))
    return m


def rrelu_noise(seed: int, stream: int, shape) -> np.ndarray:
    m = np.zeros(shape, np.float32)
    lib().nsd_oracle_rrelu_noise(seed, stream, m.size, _p(m))
    return m


def rrelu_eval_slope() -> float:
    return float(lib().nsd_oracle_rrelu_eval_slope())
