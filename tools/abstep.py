#!/usr/bin/env python3
"""A/B timing of the whole train step in ONE process (box-to-box variation is ~1 %, larger than most single changes).

    python tools/abstep.py [--B 256] [--T 250] [--steps 200] [--rounds 3]

Variants: the shipped Trainer.step, the same with the optimizer in its own launch (the multi-rank sequence without the
all-reduce), and the GPU-only floor (sum of kernel durations from back-to-back launches without Python in between is
approximated by a long queue of steps: the host runs ahead, so the wall time per step IS the GPU time per step)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--T", type=int, default=250)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import nsd_amd
    from nsd_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_3class.npz"))
    g = torch.Generator().manual_seed(0)
    x = (2.7 * torch.randn(args.B, args.T, 8, generator=g)).to(dev)
    y = torch.randint(0, 3, (args.B,), generator=g).to(torch.int32).to(dev)

    def make(split_adam, fused_head=True, in_kernel_rng=True):
        m = nsd_amd.EEG_LSTM()
        m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files})
        m.to(dev).train()
        tr = Trainer(m, lr=1e-3, seed=1)
        if split_adam:
            tr.world = 2                      # takes the multi-rank launch sequence; the reducer itself is a no-op at world 1
        tr.fused_head = fused_head
        tr.in_kernel_rng = in_kernel_rng
        return tr

    variants = {"step (shipped)": make(False), "step (separate adam)": make(True),
                "step (mask tensors, fused head)": make(False, in_kernel_rng=False),
                "step (mask tensors, separate head launch)": make(False, fused_head=False, in_kernel_rng=False)}
    host = {}
    for name, tr in variants.items():
        for _ in range(20):
            tr.step(x, y)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for name, tr in variants.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tr.step(x, y)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host[name] = (t1 - t0) / args.steps * 1e6
            print(f"round {r}  {name:42s} {1e6 * (t2 - t0) / args.steps:8.1f} us/step   (host issue {host[name]:6.1f} us/step)", flush=True)


if __name__ == "__main__":
    main()
