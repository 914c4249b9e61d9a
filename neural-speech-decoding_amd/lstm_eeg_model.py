"""Drop-in counterpart of the reference's Neuro-Alpha-App/Utilities/lstm_eeg_model.py, running on MI355X.

Same public names, constructor signatures, defaults and state_dict keys as the reference
(`EEG_LSTM` lstm_eeg_model.py:13-39, `SimplePredictor` :42-101, `CLASS_NAMES` :11), so the reference's
checkpoint loads with strict=True and `tester.run_trials` / the Streamlit app can import this module
instead.  The arithmetic is NOT torch's: forward / backward call the hand-written HIP kernels of
libnsd_hip.so through the C ABI (include/nsd.h).  The nn.Linear / nn.LayerNorm sub-modules below only
hold parameters under the reference's names; their own forward() is never used.

There is no CPU implementation: parameters and inputs must be on the GPU, otherwise forward raises.
"""
from __future__ import annotations

import importlib
import importlib.util
import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import NsdError

CLASS_NAMES = ["Food", "Water", "BG-Noise"]   # verbatim from lstm_eeg_model.py:11


class _StackedLSTMParams(nn.Module):
    """Parameter container with torch.nn.LSTM's parameter names, shapes and default init."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int, bidirectional: bool = False):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.bidirectional = bool(bidirectional)
        k = 1.0 / math.sqrt(hidden_size)
        D = 2 if bidirectional else 1
        for l in range(num_layers):
            I = input_size if l == 0 else D * hidden_size
            for sfx in (("", "_reverse") if bidirectional else ("",)):         # torch.nn.LSTM's registration order
                for name, shape in ((f"weight_ih_l{l}{sfx}", (4 * hidden_size, I)), (f"weight_hh_l{l}{sfx}", (4 * hidden_size, hidden_size)),
                                    (f"bias_ih_l{l}{sfx}", (4 * hidden_size,)), (f"bias_hh_l{l}{sfx}", (4 * hidden_size,))):
                    self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def extra_repr(self) -> str:
        return (f"{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, batch_first=True"
                + (", bidirectional=True" if self.bidirectional else ""))


class _EEGFunction(torch.autograd.Function):
    """Whole-model autograd node: x, 16 parameters -> logits.  Backward returns the parameter gradients as
    views of one flat vector produced by the HIP backward kernels."""

    @staticmethod
    def forward(ctx, module: "EEG_LSTM", x: torch.Tensor, masks, *params):
        spec, flat = module.spec, module._flat
        B, T, _ = x.shape
        ws = ops.new_workspace(spec, B, T, x.device)
        drop_lstm, rrelu_slope, drop_head = masks
        logits, _ = ops.train_forward(spec, flat, x, ws, drop_lstm=drop_lstm, rrelu_slope=rrelu_slope,
                                      drop_head=drop_head, residual=module.residual)
        ctx.module, ctx.ws, ctx.masks = module, ws, masks
        ctx.save_for_backward(x, logits)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module = ctx.module
        spec = module.spec
        x, logits = ctx.saved_tensors
        drop_lstm, rrelu_slope, drop_head = ctx.masks
        # the gradient w.r.t. the EEG window, where somebody asked for it (x.requires_grad): what autograd through self.lstm(x)
        # (lstm_eeg_model.py:34) gives the reference's users; the kernels form it for H = 48 and on the generic path
        if getattr(ctx, "gates_consumed", False):
            raise NsdError("backward a second time after one that returned the input gradient: the H = 48 kernel forms dx in place of "
                           "layer 0's saved gates -- run the forward pass again")
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        ctx.gates_consumed = dx is not None and spec.H == 48
        g = ops.train_backward(spec, module._flat, x, ctx.ws, logits, dlogits=dlogits.contiguous().float(),
                               drop_lstm=drop_lstm, rrelu_slope=rrelu_slope, drop_head=drop_head,
                               residual=module.residual, dx=dx)
        offs, shapes = spec.offsets(), spec.shapes()
        grads = tuple(g[offs[n]:offs[n] + math.prod(shapes[n])].view(shapes[n]) for n in spec.names())
        return (None, dx, None) + grads            # (ctx.ws lives as long as the graph does: retain_graph may come back)


class _EEGSeqFunction(torch.autograd.Function):
    """Whole-model autograd node on the sequence-batched bf16 path (nsd_seq_*): x, parameters -> logits; backward needs
    d loss / d logits of a mean cross-entropy, which that path fuses into the forward -- so the node takes the labels
    and returns (logits, loss) with loss the mean CE the kernels computed, and only `loss.backward()` (times a scalar) is
    supported.  The Trainer calls the ops directly; this node exists so that the nn.Module surface works too."""

    @staticmethod
    def forward(ctx, module: "EEG_LSTM", x: torch.Tensor, labels: torch.Tensor, rng, *params):
        spec, flat = module.spec, module._flat
        B, T, _ = x.shape
        ws = ops.seq_workspace(spec, B, T, x.device)
        logits = ops.seq_train_fwd(spec, flat, x, labels, ws, rng=rng)
        loss = ops.seq_loss_sum(spec, ws, B, T) / B
        ctx.module, ctx.ws, ctx.rng, ctx.shape = module, ws, rng, (B, T)
        ctx.mark_non_differentiable(logits)
        return logits, loss.reshape(())

    @staticmethod
    def backward(ctx, _dlogits, dloss):
        module = ctx.module
        spec = module.spec
        B, T = ctx.shape
        g = ops.seq_train_bwd(spec, module._flat, ctx.ws, B, T, rng=ctx.rng) * dloss
        offs, shapes = spec.offsets(), spec.shapes()
        grads = tuple(g[offs[n]:offs[n] + math.prod(shapes[n])].view(shapes[n]) for n in spec.names())
        return (None, None, None, None) + grads    # (the workspace stays with the graph: a second backward re-runs the scans on it)


class EEG_LSTM(nn.Module):
    """2-layer LSTM -> attention pooling over time -> LayerNorm -> Linear/RReLU/Dropout/Linear.

    Signature and defaults of the reference (lstm_eeg_model.py:14).  Keyword-only extensions, all
    default OFF so that reference checkpoints reproduce reference logits:
      residual       add the layer input to the output of every LSTM layer l>=1 (README's "residual stack")
      normalize      per-channel z-score of the window before the LSTM (Frontend/app.py:166-170 semantics)
      bidirectional  torch.nn.LSTM(bidirectional=True) where lstm_eeg_model.py:16-22 builds the LSTM: `_reverse`
                     parameters in torch's order, the attention pooling / LayerNorm / fc.0 act on 2H columns (BASELINE cfg5)
      precision      "fp32" (default) or "bf16": the sequence-batched path for hidden sizes 64/128/256/512 (BASELINE cfg3 /
                     cfg5) -- bf16 GEMM operands and saved activations, fp32 accumulation and cell state; logits differ
                     from an fp32 run at the 1e-2 level.  bidirectional needs "bf16".
    """

    def __init__(self, input_size=8, hidden_size=48, num_layers=2, num_classes=3, dropout=0.60, *,
                 residual: bool = False, normalize: bool = False, bidirectional: bool = False, precision: str = "fp32"):
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision={precision!r}: expected 'fp32' or 'bf16'")
        self.spec = ops.ModelSpec(C=input_size, H=hidden_size, L=num_layers, K=num_classes, F=ops.FC_HIDDEN,
                                  D=2 if bidirectional else 1, residual=bool(residual) and precision == "bf16")
        self.precision = precision
        if bidirectional and precision != "bf16":
            raise ValueError("bidirectional=True is built on the sequence-batched path only: pass precision='bf16'")
        self.dropout_p = float(dropout) if num_layers > 1 else 0.0   # nn.LSTM drops only between layers (:21)
        self.head_dropout_p = float(dropout)
        self.residual, self.normalize = bool(residual), bool(normalize)
        self.lstm = _StackedLSTMParams(input_size, hidden_size, num_layers, bidirectional)
        DH = self.spec.D * hidden_size
        self.ln = nn.LayerNorm(DH)
        self.attn = nn.Linear(DH, 1)
        self.fc = nn.Sequential(
            nn.Linear(DH, ops.FC_HIDDEN),
            nn.RReLU(),
            nn.Dropout(dropout),
            nn.Linear(ops.FC_HIDDEN, num_classes),
        )
        self._flat: Optional[torch.Tensor] = None
        self._seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self._step = 0
        self._mask_override = None    # tests inject explicit (drop_lstm, rrelu_slope, drop_head)

    # ---- flat parameter storage: all 16 tensors are views into one contiguous vector -----------------
    def _named_in_order(self):
        byname = dict(self.named_parameters())
        return [(n, byname[n]) for n in self.spec.names()]

    def flatten_parameters(self) -> torch.Tensor:
        """(Re)pack the parameters into one flat fp32 vector in state_dict order; the kernels read it
        directly and the optimizer / gradient all-reduce work on it as a single buffer."""
        named = self._named_in_order()
        dev = named[0][1].device
        flat = torch.empty(self.spec.param_count, dtype=torch.float32, device=dev)
        offs = self.spec.offsets()
        with torch.no_grad():
            for n, p in named:
                seg = flat[offs[n]:offs[n] + p.numel()].view(p.shape)
                seg.copy_(p.detach().to(torch.float32))
                p.data = seg
        self._flat = flat
        return flat

    def _flat_ok(self) -> bool:
        f = self._flat
        if f is None:
            return False
        base, offs = f.data_ptr(), self.spec.offsets()
        return all(p.data_ptr() == base + 4 * offs[n] and p.dtype == torch.float32 for n, p in self._named_in_order())

    def flat_parameters(self) -> torch.Tensor:
        if not self._flat_ok():
            self.flatten_parameters()
        return self._flat

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._flat = None        # .to()/.cuda() re-created the storages
        return out

    # ---- forward --------------------------------------------------------------------------------------
    def _train_masks(self, B: int, T: int, device):
        if self._mask_override is not None:
            return self._mask_override
        self._step += 1
        base = self._step * 4
        sp = self.spec
        drop_lstm = (ops.dropout_mask(self._seed, base, self.dropout_p, (sp.L - 1, B, T, sp.H), device)
                     if self.dropout_p > 0 and sp.L > 1 else None)
        rrelu = ops.rrelu_noise(self._seed, base + 1, (B, sp.F), device)
        drop_head = (ops.dropout_mask(self._seed, base + 2, self.head_dropout_p, (B, sp.F), device)
                     if self.head_dropout_p > 0 else None)
        return drop_lstm, rrelu, drop_head

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # x: [B, T, C] -> logits [B, num_classes]  (no softmax, as lstm_eeg_model.py:32-39)
        if x.dim() != 3 or x.shape[-1] != self.spec.C:
            raise ValueError(f"Expected x of shape [B, T, {self.spec.C}], got {tuple(x.shape)}")
        flat = self.flat_parameters()
        if not flat.is_cuda or not x.is_cuda:
            raise NsdError("EEG_LSTM runs only on the MI355X HIP path: move the module and the input to the GPU "
                           f"(module on {flat.device}, input on {x.device}); there is no CPU fallback")
        x = x.contiguous().float()     # predict() hands over a transposed view (preprocessor.py:34)
        if self.normalize:
            x = ops.zscore(x)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.precision == "bf16":
            if not self.spec.seq_path(max(x.shape[0], 1), x.shape[1]):
                raise NsdError(f"precision='bf16': model shape {self.spec} is not covered by the sequence-batched path "
                               "(hidden size 64/128/256/512, fc width and classes <= 64)")
            if self.training or need_grad:
                raise NsdError("precision='bf16': training goes through EEG_LSTM.loss(x, labels) or nsd_amd.trainer.Trainer "
                               "(the path fuses the cross-entropy into the forward); forward() serves eval mode")
            logits, _ = ops.seq_infer(self.spec, flat, x, want_probs=False)
            return logits
        if not self.training and not need_grad:
            logits, _ = ops.infer(self.spec, flat, x, residual=self.residual, want_probs=False)
            return logits
        masks = self._train_masks(x.shape[0], x.shape[1], x.device) if self.training else (None, None, None)
        params = [p for _, p in self._named_in_order()]
        return _EEGFunction.apply(self, x, masks, *params)

    def loss(self, x: torch.Tensor, labels: torch.Tensor):
        """precision='bf16' training surface: (logits, mean cross-entropy) with the loss differentiable w.r.t. the
        parameters -- `model.loss(x, y)[1].backward()` fills `.grad` of all tensors.  Train mode draws the dropout / RReLU
        streams inside the kernels."""
        if self.precision != "bf16":
            raise NsdError("EEG_LSTM.loss is the bf16 path's training surface; with precision='fp32' use "
                           "torch.nn.functional.cross_entropy(model(x), y)")
        flat = self.flat_parameters()
        x = x.contiguous().float()
        if self.normalize:
            x = ops.zscore(x)
        rng = None
        if self.training:
            self._step += 1
            rng = dict(seed=self._seed, base_stream=4 * self._step, p_lstm=self.dropout_p, p_head=self.head_dropout_p)
        params = [p for _, p in self._named_in_order()]
        return _EEGSeqFunction.apply(self, x, labels.to(torch.int32).contiguous(), rng, *params)

    @torch.no_grad()
    def predict_proba(self, x: torch.Tensor) -> torch.Tensor:
        """Eval-mode class probabilities with the softmax fused into the head kernel (lstm_eeg_model.py:97)."""
        x = x.contiguous().float()
        if self.normalize:
            x = ops.zscore(x)
        if self.precision == "bf16":
            # (the workspace of the evaluation is kept: its status word says whether NaN probabilities are a result or a failure)
            self._last_seq_ws = ops.seq_workspace(self.spec, x.shape[0], x.shape[1], x.device) if x.shape[0] > 0 else None
            return ops.seq_infer(self.spec, self.flat_parameters(), x, ws=self._last_seq_ws, want_probs=True)[1]
        _, probs = ops.infer(self.spec, self.flat_parameters(), x, residual=self.residual, want_probs=True)
        return probs


# Module names under which the reference's PreProcessor can live, in the reference's own resolution order
# (lstm_eeg_model.py:7-10): package import first (Frontend/app.py:22-28 puts Neuro-Alpha-App/ on sys.path and imports
# Utilities.tester -> Utilities.lstm_eeg_model -> `.preprocessor`), then the script form with Utilities/ itself on the path.
_REFERENCE_PREPROCESSOR_MODULES = ("Utilities.preprocessor", "preprocessor")


def _find_module(name: str) -> bool:
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):          # parent package missing / half-initialised entry in sys.modules
        return False


def resolve_reference_preprocessor(package: Optional[str] = None):
    """The reference's `PreProcessor` class (preprocessor.py:7-36, the MindsAI filter) as the reference would import
    it at lstm_eeg_model.py:7-10, or None when no such module is importable.  `package`: name of the package the
    re-exporting stub lives in (its `__package__`); tried first as `<package>.preprocessor`.

    Only the *lookup* is guarded: once the module is found, errors raised while importing it (e.g. its own MindsAI
    dependency missing, preprocessor.py:3-6,18-19) propagate, exactly as they would from the reference."""
    names = ([f"{package}.preprocessor"] if package else []) + list(_REFERENCE_PREPROCESSOR_MODULES)
    for name in names:
        if name.split(".")[0] == __name__.split(".")[0]:
            continue                              # never resolve to this package itself
        if _find_module(name):
            return getattr(importlib.import_module(name), "PreProcessor")
    return None


def _default_preprocessor(sr: int, tailoring_lambda: float, package: Optional[str] = None):
    """The reference always filters the window with the third-party MindsAI filter before the model
    (lstm_eeg_model.py:66,91).  The filter is out of scope here (non-commercial licence, host-side numpy) and is never
    restated: the facade uses the reference's own class when it is importable in either of the reference's import
    modes, and otherwise REFUSES to guess -- skipping the filter changes the probabilities (max |delta| ~ 38 uV on a
    real trial), so running without it must be asked for explicitly with `preprocess="identity"`."""
    cls = resolve_reference_preprocessor(package)
    if cls is None:
        raise NsdError(
            "SimplePredictor: the reference's PreProcessor (Utilities/preprocessor.py, MindsAI filter) is not importable "
            f"as any of {list(_REFERENCE_PREPROCESSOR_MODULES)}; the reference applies it to every window "
            "(lstm_eeg_model.py:91).  Put Neuro-Alpha-App/ (or Neuro-Alpha-App/Utilities/) on sys.path, pass "
            "preprocess=<object with .transform([T,C]) -> [T,C]>, or pass preprocess=\"identity\" to run the model "
            "on unfiltered windows on purpose.")
    return cls(sr=sr, tailoring_lambda=tailoring_lambda)


def _raise_on_poison(model: "EEG_LSTM", probs: np.ndarray, what: str) -> None:
    """NaN probabilities are a RESULT wherever the reference produces them (a NaN / Inf window, NaN or diverged weights:
    lstm_eeg_model.py:96-99 returns NaN probabilities and label index 0, and so does this facade on both precisions).  The one case
    that is a failed evaluation instead is the bf16 path's scan time-out (include/nsd.h, 'Failure reporting': the head then poisons
    every output with NaN): decided from the status word of the evaluation's workspace, never inferred from the NaN itself."""
    if model.precision != "bf16" or not np.isnan(probs).any():
        return
    ws = getattr(model, "_last_seq_ws", None)
    if ws is not None:
        ops.seq_raise_on_timeout(ws, what)


class IdentityPreProcessor:
    """Explicit opt-out of the MindsAI filter: same shape / dtype / ValueError contract as the reference's
    PreProcessor.transform (preprocessor.py:21-36), no filtering.  Selected with `preprocess="identity"`."""

    def __init__(self, sr: int = 125, tailoring_lambda: float = 1.25e-29):
        self.sr, self.tailoring_lambda = sr, tailoring_lambda

    def transform(self, chunk_samples_by_channels: np.ndarray) -> np.ndarray:
        x = np.asarray(chunk_samples_by_channels)
        if x.ndim != 2:
            raise ValueError(f"Expected 2D array [samples, channels], got {x.shape}")
        return x.astype(np.float32, copy=False)


class SimplePredictor:
    """Mirror of the reference's SimplePredictor (lstm_eeg_model.py:42-101): preprocess a [T,C] window,
    run the model, softmax, return (probs float32[K], label).

    `device` keeps the reference's keyword and default ("cpu", what tester.py:83 passes) but only names
    where the caller's arrays live: numpy in, numpy out.  The model itself always runs on the GPU
    (`gpu` argument, default "cuda"); if none is available construction fails loudly.

    `preprocess`: None (default) = the reference's own PreProcessor, resolved the way lstm_eeg_model.py:7-10 does
    (`preprocess_package`: the stub's `__package__`); construction FAILS when it cannot be found.
    "identity" = no filtering, on purpose.  Any object with `.transform([T,C]) -> [T,C]` is used as given.
    """

    def __init__(self, pth_path: str, sr: int, channel_order=None, input_size: int = 8, hidden_size: int = 48,
                 num_layers: int = 2, num_classes: int = 3, dropout: float = 0.60, device: str = "cpu",
                 tailoring_lambda: float = 1.25e-29, class_names=None, *, preprocess=None,
                 preprocess_package: Optional[str] = None, gpu: str = "cuda", residual: bool = False,
                 normalize: bool = False, bidirectional: bool = False, precision: str = "fp32"):
        self.device = torch.device(device)
        self.class_names = class_names or CLASS_NAMES
        if preprocess is None:
            self.pre = _default_preprocessor(sr, tailoring_lambda, preprocess_package)
        elif isinstance(preprocess, str):
            if preprocess != "identity":
                raise ValueError(f"preprocess={preprocess!r}: the only named preprocessor is \"identity\"")
            self.pre = IdentityPreProcessor(sr, tailoring_lambda)
        else:
            self.pre = preprocess
        if not torch.cuda.is_available():
            raise NsdError("SimplePredictor needs an MI355X: torch.cuda.is_available() is False and the HIP path has "
                           "no CPU fallback")
        self.gpu = torch.device(gpu)
        self.model = EEG_LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                              num_classes=num_classes, dropout=dropout, residual=residual, normalize=normalize,
                              bidirectional=bidirectional, precision=precision)
        state = torch.load(pth_path, map_location="cpu", weights_only=True)
        if isinstance(state, dict) and "state_dict" in state:     # both forms, as lstm_eeg_model.py:79-80
            state = state["state_dict"]
        self.model.load_state_dict(state, strict=True)
        self.model.to(self.gpu).eval()
        self.model.flatten_parameters()

    def predict(self, chunk_TxC: np.ndarray):
        """chunk_TxC: [T, C] float32 window -> (probs np.float32[K], label str)."""
        x = self.pre.transform(chunk_TxC)
        x_t = torch.from_numpy(np.ascontiguousarray(x[None, ...], dtype=np.float32)).to(self.gpu, non_blocking=True)
        probs = self.model.predict_proba(x_t)[0].cpu().numpy().astype(np.float32)
        _raise_on_poison(self.model, probs, "SimplePredictor.predict")
        y_idx = int(np.argmax(probs))
        return probs, self.class_names[y_idx]

    def predict_windows(self, recording_NxC: np.ndarray, window: int, hop: Optional[int] = None):
        """Batched / streaming mode: classify every `window`-sample window of a longer recording
        [N, C], `hop` samples apart (default: back to back, the live loop of tester.py:52-96 run over a file), in ONE
        launch.  Each window goes through the same preprocess -> model -> softmax as predict(), so
        predict_windows(r, T)[0][k] == predict(r[k*hop : k*hop + T])[0].  Returns (probs float32 [n, K], labels list[str])."""
        rec = np.asarray(recording_NxC)
        if rec.ndim != 2:
            raise ValueError("recording must be 2-D [N, C]")
        hop = int(hop) if hop is not None else int(window)
        if window < 1 or hop < 1:
            raise ValueError("window and hop must be positive")
        starts = list(range(0, rec.shape[0] - window + 1, hop))
        K = len(self.class_names)
        if not starts:
            return np.zeros((0, K), np.float32), []
        # the preprocessor is per window by contract (the MindsAI filter is not shift-invariant): host side, like predict()
        xs = np.stack([np.ascontiguousarray(self.pre.transform(rec[s0:s0 + window]), dtype=np.float32) for s0 in starts])
        x_t = torch.from_numpy(xs).to(self.gpu, non_blocking=True)
        probs = self.model.predict_proba(x_t).cpu().numpy().astype(np.float32)
        _raise_on_poison(self.model, probs, "SimplePredictor.predict_windows")
        return probs, [self.class_names[int(i)] for i in probs.argmax(1)]

