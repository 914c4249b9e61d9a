"""PyTorch-CPU stand-in for the reference model -- TEST INFRASTRUCTURE ONLY.

The reference module (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:13-39) cannot
travel to the GPU box, so the CPU baseline that bench.py times there, and the
"extension" oracles the reference class cannot express (explicit dropout masks,
RReLU noise, residual stack), are assembled here from stock torch.nn
primitives.  Each layer is its own single-layer nn.LSTM so masks can be applied
between layers; with masks absent this runs the same ATen/oneDNN LSTM kernels
as the reference's stacked nn.LSTM.  tools/make_goldens.py checks it against the
imported reference class before any fixture is written.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

RRELU_LOWER, RRELU_UPPER = 1.0 / 8.0, 1.0 / 3.0


class TorchRefEEG(nn.Module):
    """Stacked LSTM -> additive attention pooling over time -> LayerNorm -> MLP head."""

    def __init__(self, C=8, H=48, L=2, K=3, F=32, p_drop=0.6, residual=False, bidirectional=False):
        """bidirectional (extension, BASELINE cfg5): every layer is a stock one-layer nn.LSTM(bidirectional=True); explicit
        masks then have shape [L-1, B, T, 2H] -- the element order of the product's counter streams."""
        super().__init__()
        self.dims = (C, H, L, K, F)
        self.p_drop, self.residual, self.bidirectional = p_drop, residual, bidirectional
        D = 2 if bidirectional else 1
        self.cells = nn.ModuleList([nn.LSTM(C if l == 0 else D * H, H, 1, batch_first=True, bidirectional=bidirectional) for l in range(L)])
        self.norm = nn.LayerNorm(D * H)
        self.score = nn.Linear(D * H, 1)
        self.dense_a = nn.Linear(D * H, F)
        self.dense_b = nn.Linear(F, K)

    # -- parameter exchange with the reference's state_dict naming ------------------
    def load_reference_state(self, state: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            for l, cell in enumerate(self.cells):
                for sfx in (("", "_reverse") if self.bidirectional else ("",)):
                    for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                        getattr(cell, f"{nm}_l0{sfx}").copy_(torch.as_tensor(state[f"lstm.{nm}_l{l}{sfx}"]))
            self.norm.weight.copy_(torch.as_tensor(state["ln.weight"]))
            self.norm.bias.copy_(torch.as_tensor(state["ln.bias"]))
            self.score.weight.copy_(torch.as_tensor(state["attn.weight"]))
            self.score.bias.copy_(torch.as_tensor(state["attn.bias"]))
            self.dense_a.weight.copy_(torch.as_tensor(state["fc.0.weight"]))
            self.dense_a.bias.copy_(torch.as_tensor(state["fc.0.bias"]))
            self.dense_b.weight.copy_(torch.as_tensor(state["fc.3.weight"]))
            self.dense_b.bias.copy_(torch.as_tensor(state["fc.3.bias"]))

    def reference_named_grads(self) -> Dict[str, torch.Tensor]:
        g = {}
        for l, cell in enumerate(self.cells):
            for sfx in (("", "_reverse") if self.bidirectional else ("",)):
                for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    g[f"lstm.{nm}_l{l}{sfx}"] = getattr(cell, f"{nm}_l0{sfx}").grad
        g.update({"ln.weight": self.norm.weight.grad, "ln.bias": self.norm.bias.grad,
                  "attn.weight": self.score.weight.grad, "attn.bias": self.score.bias.grad,
                  "fc.0.weight": self.dense_a.weight.grad, "fc.0.bias": self.dense_a.bias.grad,
                  "fc.3.weight": self.dense_b.weight.grad, "fc.3.bias": self.dense_b.bias.grad})
        return g

    # -- forward -------------------------------------------------------------------
    def forward(self, x, drop_lstm: Optional[torch.Tensor] = None, rrelu_slope: Optional[torch.Tensor] = None,
                drop_head: Optional[torch.Tensor] = None, stochastic: bool = False, want: Optional[dict] = None):
        """drop_lstm [L-1,B,T,H], rrelu_slope [B,F], drop_head [B,F] are explicit multiplier
        masks / slopes.  stochastic=True draws them from torch's RNG instead (throughput
        baseline only: same amount of work as the reference's train() step)."""
        C, H, L, K, F = self.dims
        seq = x
        for l, cell in enumerate(self.cells):
            y, _ = cell(seq)
            if want is not None:
                want[f"h{l}"] = y
            if self.residual and l >= 1:
                y = y + seq
            if l < L - 1:
                if drop_lstm is not None:
                    y = y * drop_lstm[l]
                elif stochastic:
                    y = torch.nn.functional.dropout(y, self.p_drop, True)
            seq = y
        w = torch.softmax(self.score(seq).squeeze(-1), dim=1)
        pooled = torch.einsum("bt,bth->bh", w, seq)
        z = self.dense_a(self.norm(pooled))
        if rrelu_slope is not None:
            z = torch.where(z >= 0, z, z * rrelu_slope)
        elif stochastic:
            z = torch.nn.functional.rrelu(z, RRELU_LOWER, RRELU_UPPER, True)
        else:
            z = torch.nn.functional.rrelu(z, RRELU_LOWER, RRELU_UPPER, False)
        if drop_head is not None:
            z = z * drop_head
        elif stochastic:
            z = torch.nn.functional.dropout(z, self.p_drop, True)
        if want is not None:
            want.update(alpha=w, pooled=pooled)
        return self.dense_b(z)


def host_cores() -> int:
    """CPU cores this process may really use: the cgroup quota if there is one (a GPU box exposes all
    host CPUs in the affinity mask but grants a share of them), else the affinity mask."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, math.ceil(float(quota) / period)))
            break
        except Exception:
            continue
    env = os.environ.get("NSD_CPU_THREADS")
    if env:
        n = int(env)
    elif n > 32:
        n = 16      # no quota visible: gpurun's documented CPU share for a one-GPU box
    return max(1, n)


class StackedTorchEEG(nn.Module):
    """The reference module's own structure for CPU timing: ONE stacked torch.nn.LSTM with its built-in inter-layer dropout
    (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22), attention pooling, LayerNorm, fc (:23-39) -- re-declared from stock
    PyTorch primitives because the reference tree is not present on the GPU box.  `bidirectional` is where :16-22 would take
    the kwarg (BASELINE cfg5).  Same oneDNN kernels as the reference class; verified equal to it in
    tests/test_oracle_golden.py (state_dict keys are the reference's)."""

    def __init__(self, C=8, H=48, L=2, K=3, dropout=0.60, bidirectional=False):
        super().__init__()
        D = 2 if bidirectional else 1
        self.lstm = nn.LSTM(input_size=C, hidden_size=H, num_layers=L, batch_first=True,
                            dropout=dropout if L > 1 else 0.0, bidirectional=bidirectional)
        self.ln = nn.LayerNorm(D * H)
        self.attn = nn.Linear(D * H, 1)
        self.fc = nn.Sequential(nn.Linear(D * H, 32), nn.RReLU(), nn.Dropout(dropout), nn.Linear(32, K))

    def forward(self, x):
        out, _ = self.lstm(x)
        w = torch.softmax(self.attn(out).squeeze(-1), dim=1)
        return self.fc(self.ln((out * w.unsqueeze(-1)).sum(dim=1)))


def time_cpu_train(B=256, T=250, C=8, H=48, L=2, K=3, threads: Optional[int] = None,
                   budget_s: float = 15.0, min_steps: int = 3, seed: int = 1234, stacked: bool = False,
                   bidirectional: bool = False):
    """CPU baseline: CE train step (zero_grad, fwd with dropout+RReLU noise, bwd, Adam lr=1e-3)
    on synthetic x = 2.7*randn.  Runs whole steps until ~budget_s of CPU time is spent.
    Returns dict(trials_per_s, ms_per_step, steps, threads)."""
    import time
    threads = threads or host_cores()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    x = 2.7 * torch.randn(B, T, C, generator=g)
    y = torch.randint(0, K, (B,), generator=g)
    m = (StackedTorchEEG(C, H, L, K, bidirectional=bidirectional) if (stacked or bidirectional) else TorchRefEEG(C, H, L, K)).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    fwd = (lambda: m(x)) if (stacked or bidirectional) else (lambda: m(x, stochastic=True))

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(fwd(), y)
        loss.backward()
        opt.step()

    step()  # warm-up (oneDNN primitive creation)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < min_steps or time.perf_counter() < t_end:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
        if len(times) >= 200:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"trials_per_s": B / med, "ms_per_step": med * 1e3, "steps": len(times), "threads": threads}


def time_cpu_infer(B=256, T=250, C=8, H=48, L=2, K=3, threads: Optional[int] = None, budget_s: float = 5.0, seed: int = 1234,
                   bidirectional: bool = False):
    """CPU baseline of eval-mode inference (the reference's predict() arithmetic, lstm_eeg_model.py:95-98, on a batch)."""
    import time
    threads = threads or host_cores()
    torch.set_num_threads(threads)
    x = 2.7 * torch.randn(B, T, C, generator=torch.Generator().manual_seed(seed))
    m = StackedTorchEEG(C, H, L, K, bidirectional=bidirectional).eval()
    times = []
    with torch.inference_mode():
        m(x)
        t_end = time.perf_counter() + budget_s
        while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 200):
            t0 = time.perf_counter()
            torch.softmax(m(x), dim=-1)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"trials_per_s": B / med, "ms_per_call": med * 1e3, "calls": len(times), "threads": threads}
