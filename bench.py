#!/usr/bin/env python3
"""EEG-trials/sec of the train step (fwd + bwd + Adam [+ gradient all-reduce]) on N MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10                      # BASELINE configs[1] ("cfg2"), the default
    python bench.py --config cfg3                                        # K=5, H=256, B=1024, bf16 (sequence-batched path)
    python bench.py --config cfg5 --steps 5 --warmup 2                   # C=64, T=1000, H=512 bidirectional, B=512 per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (synthetic windows x = 2.7*N(0,1), labels uniform; dropout 0.6 + RReLU noise ON: they are part of the reference's
train() step; inputs resident in HBM before the timed region; every rank keeps the per-GPU batch = weak scaling; the flat
gradient is summed with ONE RCCL all-reduce per step):
  cfg2  the reference's 3-class model (C=8, H=48, L=2, K=3, fp32, reference checkpoint weights), 250-step windows, 256 trials/GPU
  cfg4  the same model at 1024 trials/GPU (BASELINE configs[3]'s per-GPU share of a global batch of 8192)
  cfg3  5-class model, H=256, 1024 trials/GPU, bf16 operands / fp32 accumulation (BASELINE configs[2])
  cfg5  64 channels x 1000 steps, 2-layer bidirectional LSTM, H=512, 512 trials/GPU, bf16 (BASELINE configs[4] per GPU)

`--gpus N` with N > 1 outside a torchrun environment starts the N ranks ITSELF: the parent process (which never touches a GPU)
spawns one child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and waits for them; rank 0 prints the line.

ONE JSON line on rank 0.  Besides the driver's contract it carries
  other_configs (default config cfg2, any N) the same measurement -- value, ms_per_step, roofline, kernels_us -- for cfg3, cfg4
                (1024 trials per GPU: global 8192 at N = 8 = BASELINE configs[3]) and cfg5 (512 per GPU = configs[4]), a few steps
                each, so that every BASELINE config is driver-timed; a failure there is recorded as {"error": ...} in its place
  ranks_seen    SUM all-reduce of 1 over the process group: the number of ranks the collective library (RCCL) really joined
  preheat_steps untimed steps run before the W warm-up steps (GPU clock ramp)
  step_ms_events    median / min of the per-step duration measured with HIP events on the launch stream (second loop)
  roofline      dominant kernel (longest average launch): algorithmic FLOP per launch / its launch time measured live with
                HIP events on the launch stream, against the peak of the arithmetic it runs on; HBM view beside it;
                `traffic` from the committed rocprofv3 PMC passes IF they were taken on the kernel sources of this tree
  cpu_baseline  the same train step on PyTorch-CPU (the reference module's structure on stock nn.LSTM / oneDNN), all host
                cores, bounded; `one_thread` row beside it
  inference / T625   (cfg2, one GPU) inference trials/s at B=256,T=250, single-window latency at the real 625-sample shape,
                and the train step at T=625
"""
from __future__ import annotations

import argparse
import contextlib
import glob
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

X4_MIN_B = 513                                 # csrc/nsd_lstm2.hip: batch from which the H = 48 path runs its four-trial matrix-pipe kernels
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: fp32 vector == fp32 matrix rate; dense bf16 MFMA
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

CONFIGS = {
    "cfg2": dict(C=8, H=48, L=2, K=3, T=250, B=256, precision="fp32", bidirectional=False, dtype="f32",
                 text="cfg2: 3-class EEG_LSTM train step (dropout 0.6 + RReLU noise, CE, Adam), 8ch x 250-step windows, H=48 L=2 fp32"),
    "cfg4": dict(C=8, H=48, L=2, K=3, T=250, B=1024, precision="fp32", bidirectional=False, dtype="f32",
                 text="cfg4 per-GPU share: the cfg2 model at 1024 trials per GPU (global batch 8192 on 8 GPUs)"),
    "cfg3": dict(C=8, H=256, L=2, K=5, T=250, B=1024, precision="bf16", bidirectional=False, dtype="bf16",
                 text="cfg3: 5-class EEG_LSTM train step (dropout 0.6 + RReLU noise, CE, Adam), 8ch x 250-step windows, H=256 L=2, "
                      "bf16 GEMM operands and saved activations, fp32 accumulation / cell state"),
    "cfg5": dict(C=64, H=512, L=2, K=5, T=1000, B=512, precision="bf16", bidirectional=True, dtype="bf16",
                 text="cfg5 per-GPU share: 64ch x 1000-step windows, 2-layer bidirectional LSTM H=512 (global batch 4096 on 8 GPUs), "
                      "bf16 GEMM operands and saved activations, fp32 accumulation / cell state"),
}
# sources whose hash ties a recorded PMC traffic file to the kernels it was measured on
KERNEL_SOURCES = {
    "fp32": ["nsd_lstm2_fwd48.hip", "nsd_lstm2_bwd48.hip", "nsd_lstm2_fwd48x4.hip", "nsd_lstm2_bwd48x4.hip", "nsd_lstm2.hip", "nsd_common.h", "nsd_args.h",
             "nsd_prof.h", "nsd_bf16.h"],
    "bf16": ["nsd_scan.hip", "nsd_scan2.hip", "nsd_scan_common.h", "nsd_gemm_bf16.hip", "nsd_head_tm.hip", "nsd_seq.hip", "nsd_seq.h", "nsd_bf16.h",
             "nsd_common.h"],
}


def kernel_source_hash(precision: str) -> str:
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[precision]:
        h.update(open(os.path.join(ROOT, "neural-speech-decoding_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def recorded_traffic(config: str, precision: str, kernel: str, B: int, T: int):
    """HBM bytes per launch of `kernel` from the newest profiles/*_hbm_traffic.json taken for this config, batch and on
    exactly these kernel sources; (None, reason) otherwise -- a stale file is never quoted."""
    want = kernel_source_hash(precision)
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            j = json.load(open(p))
        except Exception:
            continue
        if j.get("config") != config or j.get("B") != B or j.get("T") != T:
            continue
        if j.get("source_sha256") != want:
            best = best or (None, f"{os.path.basename(p)} was measured on other kernel sources: not quoted", None)
            continue
        e = j.get("kernels", {}).get(kernel)
        if e:
            best = (e["hbm_bytes_per_launch_corrected"], f"profiles/{os.path.basename(p)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                                                         f"separate passes; kernel sources sha256 {want[:12]})",
                    e.get("rocprof_avg_us"))
    return best or (None, "no recorded PMC pass for this config / batch", None)


def algorithmic(cfg, B, T):
    """SURVEY 8(d) contract figures, per launch of the forward / backward recurrence kernel and per train step."""
    C, H, L, K = cfg["C"], cfg["H"], cfg["L"], cfg["K"]
    D = 2 if cfg["bidirectional"] else 1
    s_a = 2 if cfg["precision"] == "bf16" else 4
    macs_step = sum(D * 4 * H * ((C if l == 0 else D * H) + H) for l in range(L))
    fwd_flop = 2 * T * macs_step + 2 * T * D * H * 2
    x_bytes = T * C * 4
    hc = L * D * T * 2 * H * s_a                              # h and c per layer-direction-step
    out = {"flop_train": 3 * fwd_flop * B, "bytes_train": (2 * x_bytes + 2 * hc + K * 4) * B}
    if cfg["precision"] == "fp32":                          # one launch = both layers, all steps
        out.update(fwd_flop=fwd_flop * B, bwd_flop=2 * fwd_flop * B, fwd_bytes=(x_bytes + hc) * B, bwd_bytes=(x_bytes + hc) * B)
    elif fused_scans(cfg):                                  # one scan launch = BOTH layers of a unidirectional 2-layer stack:
        rec = 2 * T * 4 * H * H                             #   W_hh0, W_ih1, W_hh1 products (forward) / their transposes (backward)
        out.update(fwd_flop=3 * rec * B, bwd_flop=3 * rec * B, fwd_bytes=2 * T * 2 * H * s_a * B, bwd_bytes=2 * T * 2 * H * s_a * B)
    else:                                                   # one scan launch = one layer (all directions): the recurrent product
        rec = 2 * T * D * 4 * H * H
        out.update(fwd_flop=rec * B, bwd_flop=rec * B, fwd_bytes=D * T * 2 * H * s_a * B, bwd_bytes=D * T * 2 * H * s_a * B)
    return out


def fused_scans(cfg) -> bool:
    """The sequence-batched path runs a unidirectional two-layer stack as ONE forward and ONE backward launch (nsd_scan2.hip)."""
    return cfg["precision"] == "bf16" and not cfg["bidirectional"] and cfg["L"] == 2 and cfg["H"] in (64, 128, 256)


def scan_kernel_names(cfg):
    return ("scan2_fwd_kernel", "scan2_bwd_kernel") if fused_scans(cfg) else ("scan_fwd_kernel", "scan_bwd_kernel")


class KernelTimer:
    """HIP events (torch.cuda.Event on the stream the kernels are launched on) around selected C-ABI launches."""

    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in names}

    @contextlib.contextmanager
    def __call__(self, name):
        if name not in self.names:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.events[name].append((a, b))

    def mean_us(self):
        return {n: (1e3 * sum(a.elapsed_time(b) for a, b in ev) / len(ev)) if ev else None for n, ev in self.events.items()}


def event_times_ms(fn, n):
    """Per-call durations of n calls of fn measured with HIP events on the current stream (calls issued back to back)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


class _StubTrainer:
    """NSD_BENCH_STUB=1 (CPU test of the launcher / rendezvous / timing / reporting code, tests/test_ddp_gloo_cpu.py): the step
    is one small all-reduce over the process group and nothing else.  The line it produces says "data": "stub"."""

    def __init__(self, world, name=""):
        self.t = torch.zeros(1024)
        self.world = world
        self.fail = os.environ.get("NSD_BENCH_STUB_FAIL") == name        # test hook of the stub only: this config's step raises

    def step(self, x, y):
        if self.fail:
            raise MemoryError("stub: forced failure of this config")
        if self.world > 1:
            dist.all_reduce(self.t)
        time.sleep(0.001)

    def scan_status(self):
        return 0

    def last_loss(self):
        return 0.0


def _sync(stub):
    if not stub:
        torch.cuda.synchronize()


class ScanFailure(RuntimeError):
    """A scan group of the sequence-batched path timed out on SOME rank.  Raised on EVERY rank in the same place (the ranks agree
    on the status with one MAX all-reduce first), so the callers may treat it as a collective event."""


def run_config(name, args, rank, local, world, dev, steps, warmup, preheat, extras, cpu_baseline, stub=False):
    """Measure one config; returns the output dict on rank 0 (None elsewhere).  Raises ScanFailure on EVERY rank when a scan
    group timed out on any rank."""
    cfg = CONFIGS[name]
    small = cfg["precision"] == "fp32"
    import nsd_amd
    from nsd_amd import _lib, ops
    from nsd_amd.trainer import Trainer

    B, T = args.batch_per_gpu or cfg["B"], args.T or cfg["T"]
    C, H, L, K = cfg["C"], cfg["H"], cfg["L"], cfg["K"]
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    weights = "seeded default init (torch.manual_seed(4321))"
    if stub:
        model, trainer, x, y = None, _StubTrainer(world, name), None, None
    else:
        torch.manual_seed(4321)                   # identical initial weights on every rank (the Trainer broadcasts rank 0's anyway)
        model = nsd_amd.EEG_LSTM(C, H, L, K, dropout=0.60, precision=cfg["precision"], bidirectional=cfg["bidirectional"])
        wpath = os.path.join(ROOT, "tests", "golden", "weights_3class.npz")
        if name in ("cfg2", "cfg4") and os.path.exists(wpath):
            w = np.load(wpath)
            model.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=True)
            weights = "reference checkpoint"
        model.to(dev).train()
        trainer = Trainer(model, lr=1e-3, seed=1234, stochastic=True)
        x = (2.7 * torch.randn(B, T, C, generator=g)).to(dev)
        y = torch.randint(0, K, (B,), generator=g).to(torch.int32).to(dev)

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def check_status(where):
        """A scan time-out on ANY rank ends the run on EVERY rank (the ranks agree on the status before anyone leaves)."""
        st = int(trainer.scan_status())
        if world > 1:
            t = torch.tensor([st], dtype=torch.int32, device=dev if not stub else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            st = int(t.item())
        if st != 0:
            raise ScanFailure(f"{name}: scan status {st} {where} (a scan group timed out on some rank): results invalid")

    do_step = lambda: trainer.step(x, y)
    # clock ramp: the GPU reaches its sustained clock only after some tens of milliseconds of load; run the same step untimed
    # first (library load, allocator and RCCL set-up happen here too), then the W warm-up steps.  A FIXED number of steps, not
    # a time budget: with more than one rank every step contains a collective, so all ranks must run the same number of them.
    for _ in range(max(preheat, 0)):
        do_step()
    _sync(stub)
    note(f"{name}: warm-up {warmup} steps of B={B}/GPU T={T} on {world} GPU(s)")
    for _ in range(max(warmup, 1)):
        do_step()
    _sync(stub)
    check_status("after the warm-up")
    note("timing")

    def barrier():
        if world > 1:
            dist.barrier()
        _sync(stub)

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        do_step()
    barrier()
    dt = time.perf_counter() - t0
    check_status("after the timed loop")
    loss = trainer.last_loss()

    # second loop of the same K steps: per-step HIP events + per-kernel HIP events on the launch stream (kept out of the
    # timed region so that recording does not perturb `value`); profiles/ holds the rocprofv3 summary of the same command
    step_ms, kern_us = None, {}
    if not args.no_kernel_timing and not stub:
        step_ms = event_times_ms(do_step, steps)
        if small:
            timer = KernelTimer(["nsd_lstm_head_train_rng", "nsd_lstm_head_train", "nsd_lstm_fwd", "nsd_lstm_bwd_rng", "nsd_lstm_bwd",
                                 "nsd_head_train", "nsd_grad_reduce", "nsd_grad_reduce_adam", "nsd_adam_step"])
            ops.set_launch_hook(timer)
            for _ in range(steps):
                do_step()
            torch.cuda.synchronize()
            ops.set_launch_hook(None)
            kern_us = {k: v for k, v in timer.mean_us().items() if v is not None}
        else:
            # the sequence-batched path issues many kernels per C call: their per-launch HIP events live in the DIAGNOSTIC twin of
            # the library (csrc/nsd_diag.h: same objects, orchestration compiled with -DNSD_DIAG=1) -- this loop, outside the
            # timed region, runs through it; profiles/ holds rocprofv3's durations of the PRODUCT library for the same kernels
            with _lib.diagnostic_library():
                do_step()                                       # (code-object load of the twin)
                torch.cuda.synchronize()
                ops.seq_profile(True)
                for _ in range(steps):
                    do_step()
                torch.cuda.synchronize()
                kern_us = {k: 1e3 * ms / n for k, (ms, n) in ops.seq_profile_read().items() if n}
                ops.seq_profile(False)
            check_status("after the per-kernel timing loop")

    ranks_seen = 1
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if not stub else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ones = torch.ones(1, dtype=torch.int32, device=dev if not stub else "cpu")      # what the collective library itself saw
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())

    out = None
    if rank == 0:
        alg = algorithmic(cfg, B, T)
        peak = PEAK_TFLOPS[cfg["dtype"]]
        ms_per_step = 1e3 * dt / steps
        value = world * B * steps / dt
        out = {
            "metric": "EEG-trials/sec (train fwd+bwd)" if not stub else "STUB: launcher / rendezvous rehearsal, not a measurement",
            "value": round(value, 1), "unit": "trials/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "stub" if stub else "synthetic",
            "preheat_steps": preheat, "ranks_seen": ranks_seen,
            "config": {"workload": f"{cfg['text']}, batch {B}/GPU", "name": name,
                       "batch_per_gpu": B, "global_batch": B * world, "T": T, "C": C, "H": H, "L": L, "K": K,
                       "bidirectional": cfg["bidirectional"], "weights": weights, "parallelism": f"dp{world}",
                       "loss_last_step": round(loss, 5), "launch": "eager"},
        }
        if step_ms:
            out["step_ms_events"] = {"median": round(statistics.median(step_ms), 4), "min": round(min(step_ms), 4), "n": len(step_ms),
                                     "trials_per_s_at_median": round(world * B / (statistics.median(step_ms) * 1e-3), 1)}
        if kern_us:
            if small:
                fwd_key = next(k for k in ("nsd_lstm_head_train_rng", "nsd_lstm_head_train", "nsd_lstm_fwd") if kern_us.get(k))
                bwd_key = next(k for k in ("nsd_lstm_bwd_rng", "nsd_lstm_bwd") if kern_us.get(k))
                x4 = B >= X4_MIN_B                                # (csrc/nsd_lstm2.hip: four trials per workgroup on the matrix pipe)
                names = {fwd_key: "lstm2_fwd48x4_kernel" if x4 else "lstm2_fwd48_kernel", bwd_key: "lstm2_bwd48x4_kernel" if x4 else "lstm2_bwd48_kernel"}
                bound = "mfma" if x4 else "fp32-valu"
                common = ("fp32 path: arithmetic intensity ~110 FLOP/B >> ridge (~20) so the compute roof binds, not HBM; peak = 157.3 "
                          "TFLOP/s fp32 (packed-FMA vector rate == f32 MFMA rate on gfx950), held against the kernel's ALGORITHMIC fp32 "
                          "FLOP; north_star's 40 % of HBM is out of reach for H=48 by construction (at the fp32 peak the step would "
                          "still take 71 us per 256 trials = 18 % of 8 TB/s for its algorithmic bytes): see `hbm` for the measured HBM "
                          "view; in the backward kernels the weight-gradient sums over time (47 % of their FLOP) run as split-bf16 "
                          "products -- x = hi + lo, hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, every "
                          "product within 3 x 2^-18 of the fp32 one -- while recurrences, hand-offs and the whole forward pass are exact "
                          "fp32; ")
                why = common + ("four trials per workgroup: every recurrent product is a v_mfma_f32_4x4x1_16B_f32 (one per 8 cycles and SIMD "
                                "= the fp32 rate); a step is the recurrences' serial tail (LDS hand-off, cell, barrier) plus the matrix "
                                "pipe's share" if x4 else
                                "one trial per workgroup: v_pk_fma_f32 (VALU) recurrences whose per-step instruction streams are the step; "
                                "the backward pass's layer hand-off runs as v_mfma_f32_4x4x1 over four steps at a time")
            else:
                fwd_key, bwd_key = "scan_fwd", "scan_bwd"
                names = dict(zip((fwd_key, bwd_key), scan_kernel_names(cfg)))
                bound = "mfma"
                why = SEQ_ROOFLINE_NOTE
            dom = max((fwd_key, bwd_key), key=lambda n: kern_us[n])
            t_s = kern_us[dom] * 1e-6
            fl = alg["fwd_flop"] if dom == fwd_key else alg["bwd_flop"]
            by = alg["fwd_bytes"] if dom == fwd_key else alg["bwd_bytes"]
            tf, gbs = fl / t_s / 1e12, by / t_s / 1e9
            traffic, tsrc, rocprof_us = recorded_traffic(name, cfg["precision"], names[dom], B, T)
            out["roofline"] = {
                "kernel": names[dom], "bound": bound, "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(tf / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                "avg_launch_us": round(kern_us[dom], 2), "algorithmic_flop_per_launch": fl,
                "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": by},
                "note": why,
                # which library the HIP events were taken on: the fp32 path's C calls are one kernel each and are timed on the
                # PRODUCT library; the bf16 path's per-kernel events exist only in the diagnostic twin (same kernel objects)
                "timed_on": "libnsd_hip.so (HIP events around the one-kernel C call)" if small else
                            "libnsd_hip_diag.so (the product's kernel objects; per-kernel HIP events compiled into the orchestration)",
            }
            if rocprof_us:                                  # rocprofv3's average of the PRODUCT library's kernel, same sources
                tfr = fl / (rocprof_us * 1e-6) / 1e12
                out["roofline"]["rocprof_product"] = {"avg_launch_us": rocprof_us, "achieved": round(tfr, 3), "frac": round(tfr / peak, 4),
                                                      "source": tsrc.split(" ")[0].replace("hbm_traffic.json", "kernel_stats.csv")}
            if traffic:
                out["roofline"]["traffic_over_algorithmic"] = round(traffic / by, 2)
                out["roofline"]["hbm"]["measured_traffic_GBps"] = round(traffic / t_s / 1e9, 1)
            out["kernels_us"] = {k: round(v, 2) for k, v in kern_us.items()}
            out["step_frac_of_peak"] = round(alg["flop_train"] / (ms_per_step * 1e-3) / 1e12 / peak, 4)
            out["step_hbm_frac"] = round(alg["bytes_train"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        note(f"{name} GPU: {value:.0f} trials/s, {ms_per_step:.3f} ms/step")

        # ---- further points of SURVEY 8(d) (one GPU only; bounded) -------------------------------------------------------
        if world == 1 and extras and not args.no_kernel_timing and not stub:
            model.eval()
            with torch.no_grad():
                for _ in range(5):
                    model(x)
                ms = event_times_ms(lambda: model.predict_proba(x), 50 if small else 10)
                out["inference"] = {"batch": {"B": B, "T": T, "median_ms": round(statistics.median(ms), 4),
                                              "trials_per_s": round(B / (statistics.median(ms) * 1e-3), 1)}}
                if small:
                    x1 = (2.7 * torch.randn(1, 625, C, generator=g)).to(dev)
                    for _ in range(5):
                        model.predict_proba(x1)
                    ms1 = event_times_ms(lambda: model.predict_proba(x1), 100)
                    out["inference"]["single_window_T625"] = {"median_ms": round(statistics.median(ms1), 4), "min_ms": round(min(ms1), 4)}
            model.train()
            if small:
                x6 = (2.7 * torch.randn(B, 625, C, generator=g)).to(dev)
                for _ in range(20):
                    trainer.step(x6, y)
                ms6 = event_times_ms(lambda: trainer.step(x6, y), 50)
                out["T625"] = {"B": B, "T": 625, "median_ms_per_step": round(statistics.median(ms6), 4),
                               "trials_per_s": round(B / (statistics.median(ms6) * 1e-3), 1)}

        if world == 1 and cpu_baseline and not stub:
            note("timing the PyTorch-CPU baseline")
            from oracle.torch_ref import host_cores, time_cpu_train
            # a BOUNDED sample of the same workload: the full batch where a step takes a fraction of a second, a slice of
            # the batch for the large models (the CPU rate per trial does not depend on the batch beyond oneDNN's blocking)
            Bs = B if small else (64 if name == "cfg3" else 4)
            kw = dict(B=Bs, T=T, C=C, H=H, L=L, K=K, stacked=True, bidirectional=cfg["bidirectional"])
            cpu = time_cpu_train(budget_s=args.cpu_budget_s, **kw)
            cpu1 = time_cpu_train(threads=1, budget_s=args.cpu_budget_s / 2, min_steps=2, **kw)
            torch.set_num_threads(host_cores())
            out["cpu_baseline"] = {"value": round(cpu["trials_per_s"], 1), "unit": "trials/s", "cores": cpu["threads"],
                                   "kind": "port",
                                   "sample": f"{cpu['steps']} train steps of {Bs} trials of the same workload (T={T}) on PyTorch-CPU "
                                             f"{torch.__version__}: the reference module's structure on stock stacked nn.LSTM "
                                             f"(dropout 0.6) / oneDNN, Adam, median {cpu['ms_per_step']:.1f} ms/step",
                                   "one_thread": {"value": round(cpu1["trials_per_s"], 1), "cores": 1,
                                                  "sample": f"{cpu1['steps']} steps, median {cpu1['ms_per_step']:.1f} ms/step"}}
            out["speedup_vs_cpu"] = round(value / cpu["trials_per_s"], 1)
    # release this config's buffers (the workspaces are large: 22.7 GB at cfg5) before the next one is measured
    del trainer, model, x, y
    if not stub:
        torch.cuda.empty_cache()
    return out


SEQ_ROOFLINE_NOTE = ("bf16 path: peak = 2.5 PFLOP/s dense bf16 MFMA; the persistent scan kernels hold the recurrent weights in registers; a "
                     "launch is T + 1 serial steps (flag poll -> exchange loads through the L2 -> cell -> LDS -> 48-64 MFMAs -> exchange "
                     "stores -> drain -> flag), so the achieved fraction is set by the length of one step's chain and by the fabric "
                     "traffic of the exchange (`traffic_over_algorithmic`), not by the matrix pipe")


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a torchrun environment: this parent process, which has not touched and never touches a
    GPU, starts one child per GPU with the rendezvous environment torchrun would set, and waits.  Rank 0 prints the JSON line on
    the inherited stdout.  Returns the exit code (the first non-zero child code; the remaining children are then ended)."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:                                  # a rank failed: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch-per-gpu", type=int, default=None)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--preheat-steps", type=int, default=None, help="untimed steps before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP events and the extra points")
    ap.add_argument("--no-extras", action="store_true", help="skip the inference / T=625 points")
    ap.add_argument("--no-other-configs", action="store_true", help="default run only: skip the cfg3 / cfg4 / cfg5 measurements")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # (nothing above has initialised a GPU: importing torch does not)
    stub = bool(os.environ.get("NSD_BENCH_STUB"))
    cfg = CONFIGS[args.config]
    small = cfg["precision"] == "fp32"
    steps = args.steps if args.steps is not None else (200 if small else 20)
    warmup = args.warmup if args.warmup is not None else (50 if small else 5)
    preheat = args.preheat_steps if args.preheat_steps is not None else (500 if small else 10)

    import nsd_amd
    from nsd_amd.trainer import init_distributed

    rank, local, world = init_distributed()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    dev = None
    if not stub:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        if os.environ.get("NSD_BENCH_ONE_GPU"):      # rehearsal only: all ranks on GPU 0 (with NSD_DIST_BACKEND=gloo)
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        nsd_amd.load_library()

    def leave(code):
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(code)

    try:
        out = run_config(args.config, args, rank, local, world, dev, steps, warmup, preheat, extras=not args.no_extras,
                         cpu_baseline=not args.no_cpu_baseline, stub=stub)
    except ScanFailure as e:                        # raised on every rank in the same place: leave together
        if rank == 0:
            print(f"bench.py: {e}", file=sys.stderr)
        leave(3)

    # the driver runs `bench.py --gpus N --steps K --warmup W` = cfg2; the other BASELINE configs ride in the same line so that
    # their numbers are driver-timed too (a few steps each, no CPU leg: ~15 s in all) -- at EVERY N: with N ranks cfg4's line is
    # BASELINE configs[3] as stated (1024 trials per GPU = global 8192 at N = 8) and cfg5's is configs[4] (512 per GPU).  A failure
    # in one of them must not cost the headline: it is recorded in its place and the cfg2 line is printed regardless.
    if args.config == "cfg2" and not args.no_other_configs and args.batch_per_gpu is None and args.T is None and not args.no_kernel_timing:
        others = {}
        plan = (("cfg3", (20, 5, 10)), ("cfg4", (100, 20, 200)), ("cfg5", (8, 2, 3)))
        if stub:
            plan = tuple((n, (2, 1, 1)) for n, _ in plan)
        for name, (k, w, ph) in plan:
            try:
                o = run_config(name, args, rank, local, world, dev, k, w, ph, extras=False, cpu_baseline=False, stub=stub)
                if rank == 0:
                    others[name] = {key: o[key] for key in ("value", "unit", "ms_per_step", "steps", "warmup", "n_gpus", "ranks_seen", "dtype",
                                                            "config", "step_ms_events", "roofline", "kernels_us", "step_frac_of_peak",
                                                            "step_hbm_frac") if key in o}
            except ScanFailure as e:                # collective: every rank is here, the next config can run
                others[name] = {"error": str(e)}
            except BaseException as e:              # noqa: BLE001 -- incl. SystemExit / KeyboardInterrupt / out-of-memory
                others[name] = {"error": f"{type(e).__name__}: {e}"[:500]}
                if not stub:
                    with contextlib.suppress(Exception):
                        torch.cuda.empty_cache()
                if world > 1:
                    # a rank-local failure: the other ranks may be inside a collective of the step this rank left, so no further
                    # config can be measured.  Rank 0 still prints what it has; everyone leaves without another collective.
                    if rank == 0:
                        out["other_configs"] = others
                        print(json.dumps(out), flush=True)
                    os._exit(4)
        if rank == 0:
            out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
