#!/usr/bin/env python3
"""Race hunt for the role-split H = 48 kernels: the same train step (forward + head + backward + gradient reduction) repeated N times on
one workspace must give bit-identical gradients and logits every time (every hand-off between waves is ordered by the step barrier or by
program order: a missing wait shows up as a run that differs).

    python tools/soak_determinism.py [--reps 300] [--shapes 256x250,1024x250,300x250,544x250,64x625,7x33]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--shapes", default="256x250,1024x250,300x250,544x250,64x625,7x33")
    args = ap.parse_args()
    import nsd_amd
    from nsd_amd import ops
    dev = torch.device("cuda:0")
    spec = ops.ModelSpec()
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_3class.npz"))
    m = nsd_amd.EEG_LSTM()
    m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files})
    m.to(dev)
    flat = m.flat_parameters()
    bad = 0
    for shp in args.shapes.split(","):
        B, T = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(B * 1000 + T)
        x = (2.7 * torch.randn(B, T, 8, generator=g)).to(dev)
        y = torch.randint(0, 3, (B,), generator=g).to(torch.int32).to(dev)
        ws = ops.new_workspace(spec, B, T, dev)
        rng = dict(seed=0x1234ABCD, base_stream=44, p_lstm=0.6, p_head=0.6)
        ref = None
        diffs = 0
        for r in range(args.reps):
            logits = torch.empty(B, spec.K, device=dev)
            grads = torch.empty_like(flat)
            kw = dict(rng=rng) if ops.rng_path(spec, B, T) else {}
            ops.train_step_grads(spec, flat, x, ws, y, logits, grads, **kw)
            torch.cuda.synchronize()
            cur = (grads.clone(), logits.clone())
            if ref is None:
                ref = cur
                assert torch.isfinite(cur[0]).all() and torch.isfinite(cur[1]).all()
            elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
                diffs += 1
        print(f"B={B:5d} T={T:4d}: {args.reps} repetitions, {diffs} differ from the first", flush=True)
        bad += diffs
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
