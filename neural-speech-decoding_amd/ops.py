"""Tensor-level wrappers over the C ABI (include/nsd.h).  PyTorch is only the plumbing here: it owns the
device buffers and the HIP stream; every number is produced by the kernels in csrc/.

All functions require CUDA(HIP) fp32 tensors and raise otherwise -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import Dims, NsdError, WsLayout, check

FC_HIDDEN = 32   # width of fc.0 in the reference (lstm_eeg_model.py:26)


@dataclass(frozen=True)
class ModelSpec:
    """Static shape of an EEG_LSTM (reference ctor kwargs, lstm_eeg_model.py:14)."""
    C: int = 8
    H: int = 48
    L: int = 2
    K: int = 3
    F: int = FC_HIDDEN
    D: int = 1          # directions: 2 = bidirectional (torch's nn.LSTM(bidirectional=True)); sequence-batched path only
    residual: bool = False   # sequence-batched path: the residual extension (NSD_FLAG_RESIDUAL) as part of the model's shape

    def dims(self, B: int, T: int) -> Dims:
        return Dims(B, T, self.C, self.H, self.L, self.K, self.F)

    @property
    def param_count(self) -> int:
        if self.D == 1:
            n = _lib.lib().nsd_param_count(self.C, self.H, self.L, self.K, self.F)
        else:
            n = _lib.lib().nsd_seq_param_count(self.C, self.H, self.L, self.K, self.F, self.D)
        if n < 0:
            raise NsdError(f"bad model dims {self}")
        return int(n)

    def names(self) -> List[str]:
        out = []
        for l in range(self.L):
            for sfx in (("",) if self.D == 1 else ("", "_reverse")):      # torch's state_dict order
                out += [f"lstm.weight_ih_l{l}{sfx}", f"lstm.weight_hh_l{l}{sfx}", f"lstm.bias_ih_l{l}{sfx}", f"lstm.bias_hh_l{l}{sfx}"]
        return out + ["ln.weight", "ln.bias", "attn.weight", "attn.bias",
                      "fc.0.weight", "fc.0.bias", "fc.3.weight", "fc.3.bias"]

    def shapes(self) -> Dict[str, Tuple[int, ...]]:
        s = {}
        DH = self.D * self.H
        for l in range(self.L):
            I = self.C if l == 0 else DH
            for sfx in (("",) if self.D == 1 else ("", "_reverse")):
                s[f"lstm.weight_ih_l{l}{sfx}"] = (4 * self.H, I)
                s[f"lstm.weight_hh_l{l}{sfx}"] = (4 * self.H, self.H)
                s[f"lstm.bias_ih_l{l}{sfx}"] = (4 * self.H,)
                s[f"lstm.bias_hh_l{l}{sfx}"] = (4 * self.H,)
        s.update({"ln.weight": (DH,), "ln.bias": (DH,), "attn.weight": (1, DH), "attn.bias": (1,),
                  "fc.0.weight": (self.F, DH), "fc.0.bias": (self.F,),
                  "fc.3.weight": (self.K, self.F), "fc.3.bias": (self.K,)})
        return s

    def offsets(self) -> Dict[str, int]:
        n = 4 * self.L * self.D + 8
        offs = (C.c_int64 * n)()
        if self.D == 1:
            _call("nsd_param_layout", None, self.C, self.H, self.L, self.K, self.F, offs)
        else:
            _call("nsd_seq_param_layout", None, self.C, self.H, self.L, self.K, self.F, self.D, offs)
        return dict(zip(self.names(), [int(o) for o in offs]))

    @property
    def seq_flags(self) -> int:
        return (_lib.NSD_FLAG_BIDIR if self.D == 2 else 0) | (_lib.NSD_FLAG_RESIDUAL if self.residual else 0) | _seq_extra_flags

    def seq_path(self, B: int = 32, T: int = 1) -> bool:
        """True where the sequence-batched bf16 path (nsd_seq_*) covers this model (H in 64/128/256/512, F, K <= 64)."""
        d = self.dims(B, T)
        return bool(_lib.lib().nsd_seq_supported(C.byref(d), self.seq_flags))

    def fast_path(self) -> bool:
        d = self.dims(1, 1)
        return bool(_lib.lib().nsd_fast_path(C.byref(d)))


def _dev_f32(t: Optional[torch.Tensor], name: str, shape=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise NsdError(f"{name}: expected a tensor on the MI355X (cuda/hip device), got device={t.device}; "
                       "the HIP path has no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise NsdError(f"{name}: expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise NsdError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.data_ptr()


class _StreamOf:
    """Placeholder argument: replaced by the current HIP stream of the launch device inside _call's device guard."""


STREAM = _StreamOf()


_launch_hook = None


_extra_flags = 0
_seq_extra_flags = 0


def set_seq_diag_flags(l2_exchange: bool = True, spread_groups: bool = False, fused_layers: bool = True,
                       lose_member: bool = False) -> None:
    """Diagnostic build only (inside `with _lib.diagnostic_library():`; the product library rejects these bits):
    l2_exchange=False -> scan groups always use the write-through exchange; spread_groups=True -> every group is spread over all
    XCDs; fused_layers=False -> two unidirectional layers run as two scans + GEMMs instead of one skewed launch; lose_member=True
    -> every scan launch misses its last workgroup (that group must time out and report it)."""
    global _seq_extra_flags
    _seq_extra_flags = ((0 if l2_exchange else _lib.NSD_DIAG_FLAG_NO_L2_EXCHANGE) | (_lib.NSD_DIAG_FLAG_SPREAD_GROUPS if spread_groups else 0)
                        | (0 if fused_layers else _lib.NSD_DIAG_FLAG_NO_FUSED_LAYERS) | (_lib.NSD_DIAG_FLAG_LOSE_MEMBER if lose_member else 0))


def force_fwd48(nb: int) -> None:
    """Diagnostic build only (inside `with _lib.diagnostic_library():`): pin the H = 48 forward instantiation of the fp32 fast path
    to 1 / 2 / 4 trials per workgroup (4 = the matrix-pipe kernel where it applies), 8 = the experimental one-wave-per-layer kernel
    (unfused training launches only), 0 = the product's own choice.  Process-wide
    state of the DIAGNOSTIC library: reset it to 0 before leaving the block."""
    if not _lib.diag_active():
        raise NsdError("force_fwd48: the instantiation can be pinned in the diagnostic build only: use `with _lib.diagnostic_library():`")
    _lib.check(_lib.lib().nsd_diag_force_fwd48(int(nb)), "nsd_diag_force_fwd48")


def force_bwd48(nb: int) -> None:
    """Diagnostic build only: pin the H = 48 backward kernel -- 2 = the one- / two-trial kernel, 4 = the four-trial matrix-pipe kernel
    where it applies, 0 = the product's own choice.  Reset it to 0 before leaving the block."""
    if not _lib.diag_active():
        raise NsdError("force_bwd48: the kernel can be pinned in the diagnostic build only: use `with _lib.diagnostic_library():`")
    _lib.check(_lib.lib().nsd_diag_force_bwd48(int(nb)), "nsd_diag_force_bwd48")


def set_gemm_bf16(on: bool) -> None:
    """Large-H batched path only (NSD_FLAG_BF16): GEMM operands rounded to bf16 (fp32 accumulate / storage).  Off by default."""
    global _extra_flags
    _extra_flags = _lib.NSD_FLAG_BF16 if on else 0


def set_launch_hook(hook) -> None:
    """bench.py: hook(name) -> context manager entered around each C-ABI launch (HIP-event timing)."""
    global _launch_hook
    _launch_hook = hook


def _call(name: str, dev, *args) -> None:
    """Launch `name` on device `dev` (the device of the tensors whose pointers are in `args`): the C ABI enqueues on the
    CURRENT device, so the call is wrapped in a device guard and its stream is that device's current stream -- tensors
    on cuda:1 while cuda:0 is current would otherwise launch on the wrong GPU."""
    fn = getattr(_lib.lib(), name)
    if dev is None:                                   # host-only entry points (layouts, counts)
        check(fn(*args), name)
        return
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        args = tuple(st if a is STREAM else a for a in args)
        if _launch_hook is None:
            check(fn(*args), name)
        else:
            with _launch_hook(name):
                check(fn(*args), name)


def _nbytes(t: torch.Tensor) -> int:
    return int(t.numel()) * t.element_size()


def workspace_layout(spec: ModelSpec, B: int, T: int) -> Tuple[int, WsLayout]:
    d, w = spec.dims(B, T), WsLayout()
    n = _lib.lib().nsd_workspace_bytes(C.byref(d), C.byref(w))
    if n < 0:
        check(int(n), "nsd_workspace_bytes")
    return int(n), w


def new_workspace(spec: ModelSpec, B: int, T: int, device) -> torch.Tensor:
    nbytes, _ = workspace_layout(spec, B, T)
    return torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=device)


def ws_view(ws: torch.Tensor, spec: ModelSpec, B: int, T: int, region: str) -> torch.Tensor:
    """A shaped view of one workspace region (used by tests to inspect intermediates)."""
    _, w = workspace_layout(spec, B, T)
    H, L, F = spec.H, spec.L, spec.F
    shapes = {"hseq": (L, B, T, H), "cseq": (L, B, T, H), "gact": (L, B, T, H, 4), "inseq": (max(L - 1, 0), B, T, H),
              "top": (B, T, H), "alpha": (B, T), "pooled": (B, H), "fc0_pre": (B, F), "dscore": (B, T),
              "dpooled": (B, H), "loss": (B,), "adpack": (B, T, 4)}
    shp = shapes[region]
    n = 1
    for v in shp:
        n *= v
    off = getattr(w, region)
    return ws[off:off + n].view(shp)


def zscore(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(x - mean_T) / (std_T + 1e-6) per trial and channel; x [B,T,C] or [T,C].  (app.py:166-170)"""
    squeeze = x.dim() == 2
    x3 = x.unsqueeze(0) if squeeze else x
    if x3.dim() != 3:
        raise ValueError(f"Expected [B,T,C] or [T,C], got {tuple(x.shape)}")
    x3 = x3.contiguous()
    y = torch.empty_like(x3) if out is None else out
    B, T, Cc = x3.shape
    _call("nsd_zscore_fwd", x3.device, _dev_f32(x3, "x"), _dev_f32(y, "y", x3.shape), B, T, Cc, STREAM)
    return y[0] if squeeze else y


def infer(spec: ModelSpec, flat: torch.Tensor, x: torch.Tensor, *, residual: bool = False,
          want_probs: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Eval-mode forward: logits [B,K] (+ softmax probabilities)."""
    B, T, Cc = x.shape
    if Cc != spec.C:
        raise NsdError(f"x has {Cc} channels, model expects {spec.C}")
    d = spec.dims(B, T)
    logits = torch.empty((B, spec.K), dtype=torch.float32, device=x.device)
    probs = torch.empty_like(logits) if want_probs else None
    if B == 0:                      # empty batch: nothing to launch (empty tensors have no device pointer)
        _dev_f32(flat, "params", (spec.param_count,)); _dev_f32(x, "x")
        return logits, probs
    nscr = _lib.lib().nsd_infer_scratch_bytes(C.byref(d))
    scratch = torch.empty(max(int(nscr) // 4, 1), dtype=torch.float32, device=x.device)
    _call("nsd_infer", x.device, C.byref(d), _dev_f32(flat, "params", (spec.param_count,)), _dev_f32(x, "x"),
          (_lib.NSD_FLAG_RESIDUAL if residual else 0) | _extra_flags, _dev_f32(logits, "logits"),
          _dev_f32(probs, "probs"), scratch.data_ptr(), STREAM)
    return logits, probs


def train_forward(spec: ModelSpec, flat: torch.Tensor, x: torch.Tensor, ws: torch.Tensor, *,
                  drop_lstm: Optional[torch.Tensor] = None, rrelu_slope: Optional[torch.Tensor] = None,
                  drop_head: Optional[torch.Tensor] = None, residual: bool = False,
                  want_probs: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Train-mode forward keeping activations in `ws` (from new_workspace)."""
    B, T, Cc = x.shape
    d = spec.dims(B, T)
    flags = _lib.NSD_FLAG_TRAIN | (_lib.NSD_FLAG_RESIDUAL if residual else 0) | _extra_flags
    L = _lib.lib()
    pp = _dev_f32(flat, "params", (spec.param_count,))
    _call("nsd_lstm_fwd", x.device, C.byref(d), pp, _dev_f32(x, "x", (B, T, spec.C)),
          _dev_f32(drop_lstm, "drop_lstm", (spec.L - 1, B, T, spec.H)), flags, _dev_f32(ws, "workspace"), _nbytes(ws), STREAM)
    logits = torch.empty((B, spec.K), dtype=torch.float32, device=x.device)
    probs = torch.empty_like(logits) if want_probs else None
    _call("nsd_head_fwd", x.device, C.byref(d), pp, _dev_f32(rrelu_slope, "rrelu_slope", (B, spec.F)),
          _dev_f32(drop_head, "drop_head", (B, spec.F)), ws.data_ptr(), _nbytes(ws), logits.data_ptr(),
          _dev_f32(probs, "probs"), STREAM)
    return logits, probs


def train_backward(spec: ModelSpec, flat: torch.Tensor, x: torch.Tensor, ws: torch.Tensor, logits: torch.Tensor, *,
                   dlogits: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                   scale: Optional[float] = None, drop_lstm=None, rrelu_slope=None, drop_head=None,
                   residual: bool = False, grads: Optional[torch.Tensor] = None,
                   accumulate: bool = False, dx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward through head + LSTM; returns the flat gradient vector (same layout as the parameters).
    Give either dlogits [B,K], or int32 labels [B] (+ scale, default 1/B) for fused mean cross-entropy.
    `dx` [B,T,C] (optional output): the gradient w.r.t. the EEG window (H = 48 and the generic path; it consumes layer 0's saved gates
    on the H = 48 path: no second backward on the same forward then)."""
    B, T, _ = x.shape
    d = spec.dims(B, T)
    L = _lib.lib()
    pp = _dev_f32(flat, "params", (spec.param_count,))
    if dlogits is None:
        if labels is None:
            raise NsdError("train_backward needs dlogits or labels")
        if labels.dtype != torch.int32 or not labels.is_cuda or not labels.is_contiguous():
            raise NsdError("labels must be a contiguous int32 tensor on the device")
        lab_ptr = labels.data_ptr()
    else:
        lab_ptr = None
    scale = (1.0 / max(B, 1)) if scale is None else float(scale)
    flags = _lib.NSD_FLAG_TRAIN | (_lib.NSD_FLAG_RESIDUAL if residual else 0) | _extra_flags
    wsp, wsn = _dev_f32(ws, "workspace"), _nbytes(ws)
    _call("nsd_head_bwd", x.device, C.byref(d), pp, _dev_f32(rrelu_slope, "rrelu_slope"), _dev_f32(drop_head, "drop_head"),
          _dev_f32(logits, "logits", (B, spec.K)), _dev_f32(dlogits, "dlogits", (B, spec.K)), lab_ptr, scale, wsp, wsn, STREAM)
    _call("nsd_lstm_bwd", x.device, C.byref(d), pp, _dev_f32(x, "x"), _dev_f32(drop_lstm, "drop_lstm"), flags, wsp, wsn,
          _dev_f32(dx, "dx", tuple(x.shape)) if dx is not None else None, STREAM)
    if grads is None:
        grads = torch.empty(spec.param_count, dtype=torch.float32, device=x.device)
        accumulate = False
    _call("nsd_grad_reduce", x.device, C.byref(d), wsp, wsn, _dev_f32(grads, "grads", (spec.param_count,)),
          1 if accumulate else 0, STREAM)
    return grads


def train_step_grads(spec: ModelSpec, flat: torch.Tensor, x: torch.Tensor, ws: torch.Tensor, labels: torch.Tensor,
                     logits: torch.Tensor, grads: torch.Tensor, *, scale: Optional[float] = None, drop_lstm=None,
                     rrelu_slope=None, drop_head=None, residual: bool = False, adam: Optional[dict] = None,
                     fused_head: bool = True, rng: Optional[dict] = None) -> None:
    """The launches of one training evaluation: lstm fwd + head (fwd, mean CE, bwd) in one launch where the shape allows
    (nsd_lstm_head_train; fused_head=False forces the two separate launches), lstm bwd, slab reduce -> `grads` (flat,
    overwritten).  `logits` [B,K] is an output buffer.

    rng=dict(seed=, base_stream=, p_lstm=, p_head=): dropout multipliers and RReLU slopes are generated inside the kernels
    (bit-identical to passing the tensors of nsd_train_masks with the same seed / stream ids); needs rng_path(spec, B, T).

    adam=dict(m=, v=, step=, lr=, beta1=, beta2=, eps=, weight_decay=): single-rank training -- the optimizer update of
    `flat` rides in the reduction launch (nsd_grad_reduce_adam); `grads` is still written."""
    B, T, _ = x.shape
    d = spec.dims(B, T)
    flags = _lib.NSD_FLAG_TRAIN | (_lib.NSD_FLAG_RESIDUAL if residual else 0) | _extra_flags
    pp = _dev_f32(flat, "params", (spec.param_count,))
    if labels.dtype != torch.int32 or not labels.is_cuda or not labels.is_contiguous():
        raise NsdError("labels must be a contiguous int32 tensor on the device")
    scale = (1.0 / max(B, 1)) if scale is None else float(scale)
    xp, wsp, wsn, st, dev = _dev_f32(x, "x", (B, T, spec.C)), _dev_f32(ws, "workspace"), _nbytes(ws), STREAM, x.device
    dl, sl, dh = _dev_f32(drop_lstm, "drop_lstm"), _dev_f32(rrelu_slope, "rrelu_slope"), _dev_f32(drop_head, "drop_head")
    lp = _dev_f32(logits, "logits", (B, spec.K))
    if rng is not None:
        # the three random streams of the step are generated inside the kernels: no mask tensors
        if drop_lstm is not None or rrelu_slope is not None or drop_head is not None:
            raise NsdError("train_step_grads: pass either rng= or explicit mask tensors, not both")
        r = _lib.Rng(int(rng["seed"]) & 0xFFFFFFFFFFFFFFFF, int(rng["base_stream"]) & 0xFFFFFFFF, float(rng["p_lstm"]), float(rng["p_head"]))
        _call("nsd_lstm_head_train_rng", dev, C.byref(d), pp, xp, C.byref(r), labels.data_ptr(), scale, flags, wsp, wsn, lp, st)
        _call("nsd_lstm_bwd_rng", dev, C.byref(d), pp, xp, C.byref(r), flags, wsp, wsn, st)
    else:
        if fused_head:
            _call("nsd_lstm_head_train", dev, C.byref(d), pp, xp, dl, sl, dh, labels.data_ptr(), scale, flags, wsp, wsn, lp, st)
        else:
            _call("nsd_lstm_fwd", dev, C.byref(d), pp, xp, dl, flags, wsp, wsn, st)
            _call("nsd_head_train", dev, C.byref(d), pp, sl, dh, labels.data_ptr(), scale, wsp, wsn, lp, st)
        _call("nsd_lstm_bwd", dev, C.byref(d), pp, xp, dl, flags, wsp, wsn, None, st)
    gp = _dev_f32(grads, "grads", (spec.param_count,))
    if adam is None:
        _call("nsd_grad_reduce", dev, C.byref(d), wsp, wsn, gp, 0, st)
    else:
        _call("nsd_grad_reduce_adam", dev, C.byref(d), wsp, wsn, gp, pp, _dev_f32(adam["m"], "m", flat.shape), _dev_f32(adam["v"], "v", flat.shape),
              adam.get("lr", 1e-3), adam.get("beta1", 0.9), adam.get("beta2", 0.999), adam.get("eps", 1e-8),
              adam.get("weight_decay", 0.0), 1.0, int(adam["step"]), st)


def rng_path(spec: ModelSpec, B: int, T: int) -> bool:
    """True where the kernels can generate the train-mode random streams themselves (nsd_rng_path)."""
    d = spec.dims(B, T)
    return bool(_lib.lib().nsd_rng_path(C.byref(d)))


def loss_sum(spec: ModelSpec, ws: torch.Tensor, B: int, T: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sum of the per-trial CE losses written by the labels form of train_backward (device scalar)."""
    d = spec.dims(B, T)
    out = torch.empty(1, dtype=torch.float32, device=ws.device) if out is None else out
    _call("nsd_loss_sum", ws.device, C.byref(d), _dev_f32(ws, "workspace"), _nbytes(ws), out.data_ptr(), STREAM)
    return out


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, step: int, lr: float = 1e-3,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0,
              grad_scale: float = 1.0, skip: Optional[torch.Tensor] = None) -> None:
    """torch.optim.Adam update of the flat vector.  skip: device fp32 flag (ops.seq_guard); non-zero -> nothing is updated."""
    n = p.numel()
    if skip is None:
        _call("nsd_adam_step", p.device, n, _dev_f32(p, "p"), _dev_f32(g, "g", p.shape), _dev_f32(m, "m", p.shape),
              _dev_f32(v, "v", p.shape), lr, beta1, beta2, eps, weight_decay, grad_scale, step, STREAM)
    else:
        _call("nsd_adam_step_guarded", p.device, n, _dev_f32(p, "p"), _dev_f32(g, "g", p.shape), _dev_f32(m, "m", p.shape),
              _dev_f32(v, "v", p.shape), lr, beta1, beta2, eps, weight_decay, grad_scale, step, _dev_f32(skip, "skip"), STREAM)


def dropout_mask(seed: int, stream_id: int, p: float, shape, device) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _call("nsd_dropout_mask", out.device, seed & 0xFFFFFFFFFFFFFFFF, stream_id, p, out.numel(), _dev_f32(out, "out"), STREAM)
    return out


def rrelu_noise(seed: int, stream_id: int, shape, device) -> torch.Tensor:
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _call("nsd_rrelu_noise", out.device, seed & 0xFFFFFFFFFFFFFFFF, stream_id, out.numel(), _dev_f32(out, "out"), STREAM)
    return out


# ---- sequence-batched path (large hidden sizes): building blocks ---------------------------------------------------------
def gemm_bf16(a: torch.Tensor, b: torch.Tensor, *, a_kmajor: bool = False, b_kmajor: bool = False, b_shift: int = 0,
              epilogue: int = 0, bias: Optional[torch.Tensor] = None, splits: int = 1) -> torch.Tensor:
    """C[M,N] = A . B on the matrix pipe (nsd_gemm_bf16): bf16 device tensors, fp32 accumulate.
    a: [M,K] (or [K,M] when a_kmajor), b: [N,K] (or [K,N] when b_kmajor).  epilogue 0 -> fp32 [splits,M,N] summed here when
    splits > 1; 1 -> bf16 [M,N]; 2 -> bf16 accumulator tiles [N/32, M/32, 64, 16] (+ bias[m]); 3 -> the same tiles, register group
    first: [N/32, M/32, 4, 64, 4]."""
    for t, n in ((a, "a"), (b, "b")):
        if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise NsdError(f"gemm_bf16: {n} must be a contiguous bf16 tensor on the MI355X")
    K, M = (a.shape[0], a.shape[1]) if a_kmajor else (a.shape[1], a.shape[0])
    Kb, N = (b.shape[0], b.shape[1]) if b_kmajor else (b.shape[1], b.shape[0])
    if K != Kb:
        raise NsdError(f"gemm_bf16: K mismatch {K} vs {Kb}")
    dev = a.device
    if epilogue == 0:
        c = torch.empty((max(splits, 1), M, N), dtype=torch.float32, device=dev)
    elif epilogue == 1:
        c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    elif epilogue == 2:
        c = torch.empty((N // 32, M // 32, 64, 16), dtype=torch.bfloat16, device=dev)
    else:
        c = torch.empty((N // 32, M // 32, 4, 64, 4), dtype=torch.bfloat16, device=dev)
    _call("nsd_gemm_bf16", dev, a.data_ptr(), a.shape[1], int(a_kmajor), b.data_ptr(), b.shape[1], int(b_kmajor), int(b_shift),
          c.data_ptr(), N, int(epilogue), _dev_f32(bias, "bias", (M,)), M, N, K, int(splits), STREAM)
    if epilogue == 0:
        return c[0] if splits <= 1 else c.sum(0)
    return c


# ---- sequence-batched path (nsd_seq_*): large hidden sizes, optional bidirectional, bf16 operands -------------------------
def seq_workspace(spec: ModelSpec, B: int, T: int, device) -> torch.Tensor:
    d = spec.dims(B, T)
    n = _lib.lib().nsd_seq_workspace_bytes(C.byref(d), spec.seq_flags)
    if n < 0:
        check(int(n), "nsd_seq_workspace_bytes")
    ws = torch.empty(int(n), dtype=torch.uint8, device=device)
    _call("nsd_seq_workspace_init", ws.device, ws.data_ptr(), _nbytes(ws), STREAM)     # persistent header: sticky status = 0
    return ws


def _seq_rng(rng: Optional[dict]):
    if rng is None:
        return None
    return C.byref(_lib.Rng(int(rng["seed"]) & 0xFFFFFFFFFFFFFFFF, int(rng["base_stream"]) & 0xFFFFFFFF, float(rng["p_lstm"]),
                            float(rng["p_head"])))


SEQ_ST_TIMEOUT_MASK, SEQ_ST_NONFINITE = 3, 4


def seq_status(ws: torch.Tensor, detail: bool = False):
    """0 = ok; bit 0 / bit 1 = a forward / backward scan group timed out, in the last evaluation or (sticky) in any evaluation
    since the workspace was created: results invalid; bit 2 (SEQ_ST_NONFINITE, last evaluation only) = a forward scan met a NaN / Inf
    hidden state: the logits of the affected trials are NaN, as the reference's are.  detail=True: (status, groups that ran on one XCD, groups spread over
    several XCDs) counted over the scan launches since the last forward.  Synchronises."""
    out = (C.c_int32 * 4)(-1, 0, 0, 0)
    _call("nsd_seq_status", ws.device, ws.data_ptr(), out, STREAM)
    return (int(out[0]), int(out[2]), int(out[3])) if detail else int(out[0])


def seq_guard(ws: torch.Tensor, flag: torch.Tensor) -> None:
    """flag[0] (device fp32) = 1 if `ws` reports a scan time-out (last evaluation or sticky) or non-finite activations in the last
    evaluation, else 0.  Enqueued, no sync."""
    _call("nsd_seq_guard", ws.device, ws.data_ptr(), _dev_f32(flag, "flag"), STREAM)


def seq_raise_on_timeout(ws: torch.Tensor, what: str, nonfinite: bool = False) -> None:
    """Synchronises; raises NsdError when `ws` reports a scan time-out, or (nonfinite=True: what a training loop wants to know)
    non-finite activations in the last evaluation."""
    st = seq_status(ws)
    if nonfinite and (st & SEQ_ST_NONFINITE) and not (st & SEQ_ST_TIMEOUT_MASK):
        raise NsdError(f"{what}: non-finite activations (NaN / Inf hidden state) in the last evaluation of the sequence-batched path "
                       f"(status {st}): a NaN / Inf window or diverged / NaN weights.  The affected trials' logits and the gradients "
                       "are NaN as in the reference; the guarded Adam update was skipped")
    if st & SEQ_ST_TIMEOUT_MASK:
        stage = " and ".join(n for bit, n in ((1, "forward"), (2, "backward")) if st & bit) or f"code {st}"
        raise NsdError(f"{what}: a {stage} scan group of the sequence-batched path timed out (status {st}): its workgroups were not "
                       "all resident at once (another process on the GPU, a CU mask or a partition mode?).  Results since then "
                       "are invalid (NaN logits / loss, the guarded Adam update was skipped); allocate a fresh workspace to go on")


def seq_infer(spec: ModelSpec, flat: torch.Tensor, x: torch.Tensor, ws: Optional[torch.Tensor] = None, *,
              want_probs: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    B, T, Cc = x.shape
    if Cc != spec.C:
        raise NsdError(f"x has {Cc} channels, model expects {spec.C}")
    d = spec.dims(B, T)
    logits = torch.empty((B, spec.K), dtype=torch.float32, device=x.device)
    probs = torch.empty_like(logits) if want_probs else None
    if B == 0:
        return logits, probs
    ws = seq_workspace(spec, B, T, x.device) if ws is None else ws
    _call("nsd_seq_infer", x.device, C.byref(d), _dev_f32(flat, "params", (spec.param_count,)), _dev_f32(x, "x"), spec.seq_flags,
          _dev_f32(logits, "logits"), _dev_f32(probs, "probs"), ws.data_ptr(), _nbytes(ws), STREAM)
    return logits, probs


def seq_train_fwd(spec: ModelSpec, flat: torch.Tensor, x: torch.Tensor, labels: torch.Tensor, ws: torch.Tensor, *,
                  rng: Optional[dict] = None, scale: Optional[float] = None, logits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Forward + head + mean CE + head backward of one training evaluation; activations stay in `ws` for seq_train_bwd."""
    B, T, _ = x.shape
    d = spec.dims(B, T)
    if labels.dtype != torch.int32 or not labels.is_cuda or not labels.is_contiguous():
        raise NsdError("labels must be a contiguous int32 tensor on the device")
    logits = torch.empty((B, spec.K), dtype=torch.float32, device=x.device) if logits is None else logits
    scale = (1.0 / max(B, 1)) if scale is None else float(scale)
    _call("nsd_seq_train_fwd", x.device, C.byref(d), _dev_f32(flat, "params", (spec.param_count,)), _dev_f32(x, "x", (B, T, spec.C)),
          _seq_rng(rng), labels.data_ptr(), scale, spec.seq_flags, ws.data_ptr(), _nbytes(ws), _dev_f32(logits, "logits", (B, spec.K)), STREAM)
    return logits


def seq_train_bwd(spec: ModelSpec, flat: torch.Tensor, ws: torch.Tensor, B: int, T: int, *, rng: Optional[dict] = None,
                  grads: Optional[torch.Tensor] = None) -> torch.Tensor:
    """BPTT + every parameter gradient of the evaluation seq_train_fwd left in `ws` -> flat gradient vector (overwritten)."""
    d = spec.dims(B, T)
    grads = torch.empty(spec.param_count, dtype=torch.float32, device=flat.device) if grads is None else grads
    _call("nsd_seq_train_bwd", flat.device, C.byref(d), _dev_f32(flat, "params", (spec.param_count,)), _seq_rng(rng), spec.seq_flags,
          ws.data_ptr(), _nbytes(ws), _dev_f32(grads, "grads", (spec.param_count,)), STREAM)
    return grads


def seq_loss_sum(spec: ModelSpec, ws: torch.Tensor, B: int, T: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = spec.dims(B, T)
    out = torch.empty(1, dtype=torch.float32, device=ws.device) if out is None else out
    _call("nsd_seq_loss_sum", ws.device, C.byref(d), spec.seq_flags, ws.data_ptr(), _nbytes(ws), out.data_ptr(), STREAM)
    return out


SEQ_PROFILE_KINDS = ("scan_fwd", "scan_bwd", "gemm_xproj", "gemm_dw", "gemm_din", "head", "head_grads", "prep")


def seq_profile(enable: bool) -> None:
    """Diagnostic build only (`with _lib.diagnostic_library():`): start / stop the HIP-event timing of the sequence-batched
    path's kernels (nsd_seq_profile, csrc/nsd_diag.h)."""
    if not _lib.diag_active():
        raise NsdError("seq_profile: per-kernel timing lives in the diagnostic build: use `with _lib.diagnostic_library():`")
    _call("nsd_seq_profile", None, 1 if enable else 0)


def seq_profile_read() -> Dict[str, Tuple[float, int]]:
    """kind -> (total ms, launches) recorded since seq_profile(True) / the last read.  Synchronises."""
    out = {}
    for i, name in enumerate(SEQ_PROFILE_KINDS):
        ms, n = C.c_float(0), C.c_int32(0)
        _call("nsd_seq_profile_read", None, i, C.byref(ms), C.byref(n))
        out[name] = (float(ms.value), int(n.value))
    return out
