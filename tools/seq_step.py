"""Time the sequence-batched path (nsd_seq_*) at a BASELINE shape: python tools/seq_step.py [--cfg cfg3|cfg5] [--B N] [--T N] [--iters N]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nsd_amd
from nsd_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--B", type=int, default=0)
ap.add_argument("--T", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--nodrop", action="store_true")
a = ap.parse_args()
if a.cfg == "cfg3":
    spec, B, T = ops.ModelSpec(C=8, H=256, L=2, K=5), 1024, 250
else:
    spec, B, T = ops.ModelSpec(C=64, H=512, L=2, K=5, D=2), 512, 1000
B, T = a.B or B, a.T or T
dev = torch.device("cuda:0")
torch.manual_seed(0)
P = spec.param_count
k = 1.0 / np.sqrt(spec.H)
flat = (torch.rand(P, device=dev) * 2 - 1) * k
offs = spec.offsets()
flat[offs["ln.weight"]:offs["ln.weight"] + spec.D * spec.H] = 1.0
x = 2.7 * torch.randn(B, T, spec.C, device=dev)
y = torch.randint(0, spec.K, (B,), device=dev, dtype=torch.int32)
ws = ops.seq_workspace(spec, B, T, dev)
print(f"{a.cfg}: B={B} T={T} H={spec.H} D={spec.D} params={P} workspace={ws.numel() / 2**30:.2f} GiB", flush=True)
m, v, g = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
rng = None if a.nodrop else dict(seed=1, base_stream=4, p_lstm=0.6, p_head=0.6)

def step(i):
    r = None if rng is None else dict(rng, base_stream=4 * (i + 1))
    ops.seq_train_fwd(spec, flat, x, y, ws, rng=r)
    ops.seq_train_bwd(spec, flat, ws, B, T, rng=r, grads=g)
    ops.adam_step(flat, g, m, v, step=i + 1)

step(0); torch.cuda.synchronize()
print("status after first step:", ops.seq_status(ws), "loss", float(ops.seq_loss_sum(spec, ws, B, T).item()) / B, flush=True)
for phase in ("infer", "fwd", "step"):
    ts = []
    for i in range(a.iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if phase == "infer":
            ops.seq_infer(spec, flat, x, ws, want_probs=False)
        elif phase == "fwd":
            ops.seq_train_fwd(spec, flat, x, y, ws, rng=rng)
        else:
            step(i + 1)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{phase:6s} median {1e3 * np.median(ts):8.3f} ms  min {1e3 * min(ts):8.3f} ms   -> {B / np.median(ts):10.0f} trials/s", flush=True)
print("status (code, groups on one XCD, groups spread over XCDs):", ops.seq_status(ws, detail=True), " finite grads:", bool(torch.isfinite(g).all()), flush=True)
