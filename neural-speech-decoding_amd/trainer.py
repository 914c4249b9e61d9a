"""Training loop for EEG_LSTM on MI355X: fused step through the C ABI + data-parallel gradient all-reduce.

The reference's training notebook (DeepLearning/lstm_trainer.ipynb) is not part of the reference tree
(.MISSING_LARGE_BLOBS:1), so this trainer defines the step itself:
    zero_grad -> CE(model(x), y) -> backward -> Adam(lr=1e-3)
with the reference model's train-mode stochastic parts (inter-layer dropout p, RReLU noise, head dropout p).

Per step and per GPU the stream sees, for the reference model's shape, three launches: LSTM forward + attention
pooling + head + CE + head backward (nsd_lstm_head_train_rng, the dropout / RReLU streams drawn in the kernel), BPTT
(nsd_lstm_bwd_rng), slab reduction + Adam (nsd_grad_reduce_adam); with more than one rank the last becomes
nsd_grad_reduce -> ONE all-reduce of the flat fp32 gradient over RCCL -> nsd_adam_step.  Other shapes run the unfused
equivalents (ops.train_step_grads).  Nothing synchronises the host.

Data parallelism: trials are independent, so the global batch is split contiguously over ranks
(shard_range; shards may differ by one trial and may be EMPTY); every rank scales its CE gradient by 1/B_global, the flat
gradient vector (31 764 floats = 127 KB for the reference model) is summed with ONE all-reduce, and every rank applies
the same Adam update.  Every rank enters the collective in every step -- a rank without trials contributes a zero
gradient -- and the parameters are broadcast from rank 0 once at construction, so identical replicas do not rest on
identical seeding.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import ops
from .lstm_eeg_model import EEG_LSTM


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n trials for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradAllReducer:
    """Sum one flat gradient vector over the data-parallel group with a single collective.
    backend 'nccl' is RCCL over xGMI on ROCm; 'gloo' is used by the CPU tests."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def __call__(self, flat_grad: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return flat_grad


def _on_own_device(method):
    """Run a Trainer method with the trainer's GPU as the current device (the C ABI enqueues on the current device)."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *a, **kw):
        with torch.cuda.device(self.flat.device):
            return method(self, *a, **kw)
    return wrapped


class DataParallelStep:
    """The rank-level control flow of one optimisation step, separated from how the numbers are produced so that the
    CPU tests (gloo, world_size 2 and 3, uneven and empty shards) drive exactly this code with an injected gradient
    function.  `local_grads(x, y, scale)` must leave this rank's share of the gradient of sum_b CE_b * scale in the
    flat buffer `grads`; `zero_grads()` clears it; `apply_update()` consumes it."""

    def __init__(self, grads: torch.Tensor, reducer: FlatGradAllReducer, local_grads, zero_grads, apply_update):
        self.grads, self.reducer = grads, reducer
        self.local_grads, self.zero_grads, self.apply_update = local_grads, zero_grads, apply_update

    def __call__(self, x, y, global_batch: int) -> None:
        if global_batch < 1:
            raise ValueError("global_batch must be >= 1 (the number of trials of the step over all ranks)")
        if int(x.shape[0]) > 0:
            self.local_grads(x, y, 1.0 / float(global_batch))
        else:
            self.zero_grads()              # empty shard: still take part in the collective, with a zero gradient
        self.reducer(self.grads)           # EVERY rank, EVERY step
        self.apply_update()


def broadcast_parameters(flat: torch.Tensor, group=None, src: int = 0) -> None:
    """Make every replica start from rank `src`'s parameters (one collective on the flat vector)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


class Trainer:
    def __init__(self, model: EEG_LSTM, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, seed: int = 1234, stochastic: bool = True, group=None):
        self.model = model
        self.spec = model.spec
        self.flat = model.flat_parameters()
        if not self.flat.is_cuda:
            raise ops.NsdError("Trainer needs the model on the MI355X (model.to('cuda')); there is no CPU training path")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        # the flat gradient with ONE extra element behind it: the bf16 path's failure flag (ops.seq_guard).  The whole buffer is
        # what the ranks all-reduce, so a scan time-out on ANY rank makes EVERY rank skip the update (guarded Adam) -- still
        # one collective per step.  Always 0 on the fp32 path.
        self._grads_ext = torch.zeros(self.flat.numel() + 1, dtype=torch.float32, device=self.flat.device)
        self.grads = self._grads_ext[:self.flat.numel()]
        self._skip = self._grads_ext[self.flat.numel():]
        self.reducer = FlatGradAllReducer(group)
        self.rank = dist.get_rank(group) if self.reducer.world > 1 else 0
        self.world = self.reducer.world
        broadcast_parameters(self.flat, group)     # identical replicas by construction, not by seeding
        self.fused_head = True           # nsd_lstm_head_train (one launch) where the shape allows; False: two launches
        self.in_kernel_rng = True        # dropout / RReLU streams generated inside the kernels where the shape allows
        self.seed = (int(seed) + 0x9E3779B97F4A7C15 * (self.rank + 1)) & 0xFFFFFFFFFFFFFFFF
        self.stochastic = stochastic
        self.step_count = 0
        self._bufs = {}
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        self._last_B = 0
        self._carried_failure = ""       # bf16 path, world > 1: message of a time-out found in a workspace that was replaced
        # hipGraph replay of the step (static-input path): everything that varies per step lives on the device
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=self.flat.device)
        self._graphs = {}

    # buffers that depend on the batch shape are created once and reused every step
    def _buffers(self, B: int, T: int):
        key = (B, T)
        if key not in self._bufs:
            sp, dev = self.spec, self.flat.device
            buf = {"ws": ops.new_workspace(sp, B, T, dev),
                   "logits": torch.empty((B, sp.K), dtype=torch.float32, device=dev)}
            if self.stochastic:
                if self.model.dropout_p > 0 and sp.L > 1:
                    buf["drop_lstm"] = torch.empty((sp.L - 1, B, T, sp.H), dtype=torch.float32, device=dev)
                buf["rrelu"] = torch.empty((B, sp.F), dtype=torch.float32, device=dev)
                if self.model.head_dropout_p > 0:
                    buf["drop_head"] = torch.empty((B, sp.F), dtype=torch.float32, device=dev)
            self._bufs = {key: buf}      # keep only the current shape (workspaces are large)
        return self._bufs[key]

    @_on_own_device
    def step(self, x: torch.Tensor, y: torch.Tensor, global_batch: Optional[int] = None) -> None:
        """One optimisation step on this rank's shard: x [B,T,C] fp32, y [B] int32 (device tensors).  `global_batch`: trials
        of this step over ALL ranks (default B * world, i.e. equal shards); the CE gradient is scaled by 1/global_batch so
        that the all-reduced sum is the global mean whatever the shard sizes.  B may be 0 when world > 1."""
        B = int(x.shape[0])
        if global_batch is None:
            global_batch = B * self.world
        self.step_count += 1
        if self.world == 1:
            if B == 0:
                raise ops.NsdError("Trainer.step: empty batch")
            # no exchange step between reduction and update: one launch does both
            self._local_grads(x, y, 1.0 / float(global_batch), fuse_adam=True)
        else:
            DataParallelStep(self._grads_ext, self.reducer, self._local_grads, self._grads_ext.zero_, self._adam)(x, y, global_batch)
        if B > 0:
            self._last_B, self._last_T = B, int(x.shape[1])

    def _hyper(self) -> dict:
        return dict(step=self.step_count, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                    weight_decay=self.weight_decay)

    def _adam(self) -> None:
        # bf16 path: skipped on the device when any rank's scan timed out (self._skip, summed over ranks by the all-reduce)
        ops.adam_step(self.flat, self.grads, self.m, self.v, skip=self._skip if self.model.precision == "bf16" else None, **self._hyper())

    def _prepare_input(self, x: torch.Tensor) -> torch.Tensor:
        """What EEG_LSTM.forward does to a window before the LSTM (lstm_eeg_model.py facade): contiguous fp32 and, for
        normalize=True, the per-channel z-score -- the model must be trained on what it is evaluated on."""
        x = x.contiguous().float()
        return ops.zscore(x) if self.model.normalize and x.shape[0] > 0 else x

    def _local_grads(self, x: torch.Tensor, y: torch.Tensor, scale: float, fuse_adam: bool = False) -> None:
        """Launches that leave this shard's gradient (scaled) in self.grads; fuse_adam: the update rides in the last one."""
        from . import _lib
        sp = self.spec
        B, T, _ = x.shape
        x = self._prepare_input(x)
        if self.model.precision == "bf16":
            # sequence-batched path: forward (+ head, CE, head backward), backward (+ all parameter gradients), Adam
            key = ("seq", B, T)
            if key not in self._bufs:
                # a reported time-out must not vanish with the buffer.  One rank: raise here.  Several ranks: raising on this rank
                # alone would leave the others waiting in the step's all-reduce, so the old status is carried into the failure
                # flag that rides that all-reduce (below) and every rank raises in check().
                if self.world == 1:
                    self._check_old_workspaces("Trainer.step")
                else:
                    try:
                        self._check_old_workspaces("Trainer.step")
                    except ops.NsdError as e:
                        self._carried_failure = str(e)
                self._bufs = {key: {"ws": ops.seq_workspace(sp, B, T, self.flat.device),
                                    "logits": torch.empty((B, sp.K), dtype=torch.float32, device=self.flat.device)}}
            buf = self._bufs[key]
            rng = None
            if self.stochastic:
                rng = dict(seed=self.seed, base_stream=(self.step_count & 0x3FFFFFFF) * 4, p_lstm=self.model.dropout_p,
                           p_head=self.model.head_dropout_p)
            ops.seq_train_fwd(sp, self.flat, x, y, buf["ws"], rng=rng, scale=scale, logits=buf["logits"])
            ops.seq_train_bwd(sp, self.flat, buf["ws"], B, T, rng=rng, grads=self.grads)
            ops.seq_guard(buf["ws"], self._skip)
            if self._carried_failure:                          # (seq_guard overwrites the flag with this workspace's status)
                self._skip.add_(1.0)
            if fuse_adam:
                self._adam()
            return
        buf = self._buffers(B, T)
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        sid = (self.step_count & 0x3FFFFFFF) * 4
        dl = buf.get("drop_lstm"); sl = buf.get("rrelu"); dh = buf.get("drop_head")
        rng = None
        if "rng_ok" not in buf:
            buf["rng_ok"] = ops.rng_path(sp, B, T)
        if self.in_kernel_rng and dl is not None and sl is not None and dh is not None and buf["rng_ok"]:
            # the three streams of the step are generated inside the LSTM kernels (same values as nsd_train_masks)
            rng = dict(seed=self.seed, base_stream=sid, p_lstm=self.model.dropout_p, p_head=self.model.head_dropout_p)
            dl = sl = dh = None
        elif dl is not None and sl is not None and dh is not None:
            _lib.check(L.nsd_train_masks(self.seed, sid, self.model.dropout_p, self.model.head_dropout_p, dl.numel(),
                                         dl.data_ptr(), sl.numel(), sl.data_ptr(), dh.data_ptr(), st), "train_masks")
        else:
            if dl is not None:
                _lib.check(L.nsd_dropout_mask(self.seed, sid, self.model.dropout_p, dl.numel(), dl.data_ptr(), st), "dropout_mask")
            if sl is not None:
                _lib.check(L.nsd_rrelu_noise(self.seed, sid + 1, sl.numel(), sl.data_ptr(), st), "rrelu_noise")
            if dh is not None:
                _lib.check(L.nsd_dropout_mask(self.seed, sid + 2, self.model.head_dropout_p, dh.numel(), dh.data_ptr(), st), "dropout_mask")
        ops.train_step_grads(sp, self.flat, x, buf["ws"], y, buf["logits"], self.grads, scale=scale, drop_lstm=dl,
                             rrelu_slope=sl, drop_head=dh, residual=self.model.residual, fused_head=self.fused_head, rng=rng,
                             adam=dict(m=self.m, v=self.v, **self._hyper()) if fuse_adam else None)

    # ---- hipGraph path ------------------------------------------------------------------------------------
    def static_inputs(self, B: int, T: int):
        """(x [B,T,C] fp32, y [B] int32) device buffers owned by the trainer.  Fill them in place (copy_, index_select
        with out=...) and call step_static(B, T): the whole step is then ONE hipGraph replay per segment instead of
        seven launches, with no host-side argument that changes from step to step."""
        if self.model.precision == "bf16":
            raise ops.NsdError("Trainer.static_inputs / step_static (hipGraph replay) exist for the fp32 path only; "
                               "precision='bf16' trains through Trainer.step")
        buf = self._buffers(B, T)
        if "x" not in buf:
            dev = self.flat.device
            buf["x"] = torch.zeros((B, T, self.spec.C), dtype=torch.float32, device=dev)
            buf["y"] = torch.zeros((B,), dtype=torch.int32, device=dev)
        return buf["x"], buf["y"]

    def _issue_segment_a(self, B: int, T: int) -> None:
        """step counter, random streams, lstm fwd, fused head, lstm bwd, slab reduce -> self.grads"""
        from . import _lib
        L = _lib.lib()
        buf = self._buffers(B, T)
        st = torch.cuda.current_stream().cuda_stream
        dl = buf.get("drop_lstm"); sl = buf.get("rrelu"); dh = buf.get("drop_head")
        _lib.check(L.nsd_step_counter_inc(self._step_dev.data_ptr(), st), "step_counter_inc")
        if dl is not None and sl is not None and dh is not None:
            _lib.check(L.nsd_train_masks_dev(self.seed, self._step_dev.data_ptr(), self.model.dropout_p, self.model.head_dropout_p,
                                             dl.numel(), dl.data_ptr(), sl.numel(), sl.data_ptr(), dh.data_ptr(), st), "train_masks_dev")
        elif self.stochastic:
            raise ops.NsdError("graph step needs all three random streams (dropout > 0, num_layers > 1) or stochastic=False")
        xin = buf["x"]
        if self.model.normalize:                               # a static buffer of its own: nothing is allocated inside the capture
            if "xn" not in buf:
                buf["xn"] = torch.empty_like(buf["x"])
            xin = ops.zscore(buf["x"], out=buf["xn"])
        ops.train_step_grads(self.spec, self.flat, xin, buf["ws"], buf["y"], buf["logits"], self.grads,
                             scale=1.0 / (B * self.world), drop_lstm=dl, rrelu_slope=sl, drop_head=dh, residual=self.model.residual)

    def _issue_segment_b(self) -> None:
        from . import _lib
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().nsd_adam_step_dev(self.flat.numel(), self.flat.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                                self.v.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                                                self.weight_decay, 1.0, self._step_dev.data_ptr(), st), "adam_step_dev")

    @_on_own_device
    def step_static(self, B: int, T: int) -> None:
        """One optimisation step on the trainer's static input buffers, replayed from captured hipGraphs.
        world == 1: one graph.  world > 1: graph A, the eager RCCL all-reduce of the flat gradient, graph B (Adam)."""
        key = (B, T)
        if self.model.precision == "bf16":
            raise ops.NsdError("Trainer.step_static (hipGraph replay) exists for the fp32 path only; precision='bf16' trains "
                               "through Trainer.step")
        if key not in self._graphs:
            self.static_inputs(B, T)
            self._step_dev.fill_(self.step_count)
            # warm up eagerly on a side stream (lazy module loading, hipFuncSetAttribute), then capture
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                snap = (self.flat.clone(), self.m.clone(), self.v.clone())
                self._issue_segment_a(B, T)
                self._issue_segment_b()
                self.flat.copy_(snap[0]); self.m.copy_(snap[1]); self.v.copy_(snap[2])     # the warm-up step does not count
                self._step_dev.fill_(self.step_count)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                self._issue_segment_a(B, T)
                if self.world == 1:
                    self._issue_segment_b()
            gb = None
            if self.world > 1:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb):
                    self._issue_segment_b()
            self._graphs = {key: (ga, gb)}
        ga, gb = self._graphs[key]
        self.step_count += 1
        ga.replay()
        if gb is not None:
            self.reducer(self.grads)
            gb.replay()
        self._last_B, self._last_T = B, T

    @_on_own_device
    def scan_status(self) -> int:
        """precision='bf16' only: 0 unless a scan group timed out in any step on the current workspace (the status is sticky:
        nsd_seq_status).  Synchronises."""
        for key, buf in self._bufs.items():
            if key[0] == "seq":
                return ops.seq_status(buf["ws"])
        return 0

    def _check_old_workspaces(self, what: str) -> None:
        for key, buf in self._bufs.items():
            if key[0] == "seq":
                ops.seq_raise_on_timeout(buf["ws"], what, nonfinite=True)

    @_on_own_device
    def check(self) -> None:
        """Raise NsdError if a scan group of the bf16 path timed out on ANY rank since the workspaces were created
        (synchronises this rank's stream; no collective: the flag read here was summed over the ranks by the step's own
        all-reduce, so every rank raises in the same step).  nsd_amd.train calls it on every rank at every log interval."""
        if self.model.precision != "bf16":
            return
        if float(self._skip.item()) != 0.0:
            if self._carried_failure:
                raise ops.NsdError(self._carried_failure)
            self._check_old_workspaces("Trainer")              # this rank's own status, with the stage in the message
            raise ops.NsdError("Trainer: a scan group of the sequence-batched path timed out, or met non-finite activations, on another "
                               "rank; every rank has skipped the guarded Adam update of that step (and, after a time-out, of every "
                               "step since).  Results are invalid")
        self._check_old_workspaces("Trainer")

    @_on_own_device
    def last_loss(self) -> float:
        """Mean CE loss of this rank's shard in the most recent step (synchronises)."""
        if not self._last_B:
            return float("nan")
        if self.model.precision == "bf16":
            ws = self._bufs[("seq", self._last_B, self._last_T)]["ws"]
            ops.seq_loss_sum(self.spec, ws, self._last_B, self._last_T, out=self._loss)
            loss = float(self._loss.item()) / self._last_B
            if loss != loss:                                   # NaN: the head poisons its outputs when a scan timed out
                ops.seq_raise_on_timeout(ws, "Trainer.last_loss")
            return loss
        ws = self._buffers(self._last_B, self._last_T)["ws"]
        ops.loss_sum(self.spec, ws, self._last_B, self._last_T, out=self._loss)
        return float(self._loss.item()) / self._last_B

    def state_dict(self):
        return {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                "adam_m": self.m.cpu(), "adam_v": self.v.cpu(), "step": self.step_count}

    def load_state_dict(self, sd):
        self.model.load_state_dict(sd["model"], strict=True)
        self.flat = self.model.flat_parameters()
        self.m.copy_(sd["adam_m"]); self.v.copy_(sd["adam_v"]); self.step_count = int(sd["step"])


def save_reference_checkpoint(model: EEG_LSTM, path: str) -> None:
    """Write a .pth the reference's SimplePredictor loads unchanged (raw state_dict with the reference's
    key names on CPU; lstm_eeg_model.py:77-81)."""
    torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the process group described by torchrun's environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # NSD_DIST_BACKEND=gloo: rehearsal of the multi-rank code path where RCCL cannot run (several ranks on one GPU)
        backend = backend or os.environ.get("NSD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world
