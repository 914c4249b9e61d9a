"""Per-phase cycles of a scan step from the diagnostic build (NSD_LIB=libnsd_hip_stamps.so): forward of cfg3, inference and training."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nsd_amd
from nsd_amd import ops
spec, B, T = ops.ModelSpec(C=8, H=256, L=2, K=5), 1024, 250
dev = torch.device("cuda:0")
torch.manual_seed(0)
flat = (torch.rand(spec.param_count, device=dev) * 2 - 1) / 16
x = 2.7 * torch.randn(B, T, 8, device=dev)
y = torch.randint(0, 5, (B,), device=dev, dtype=torch.int32)
ws = ops.seq_workspace(spec, B, T, dev)
names = ["failed looks x 1000", "barrier (wait for the other waves)", "mfma", "cells", "publish (tagged granule stores, no drain)", "saves+rotate", "spin on the tagged granules -> arrival", "ds_write"]
for mode in ("infer", "train"):
    for _ in range(3):
        if mode == "infer":
            ops.seq_infer(spec, flat, x, ws, want_probs=False)
        else:
            ops.seq_train_fwd(spec, flat, x, y, ws, rng=dict(seed=1, base_stream=4, p_lstm=0.6, p_head=0.6))
        torch.cuda.synchronize()
    st = ws[256:384].cpu().numpy().view(np.int32)
    acc = st[4:20].view(np.uint64)
    tot = acc.sum()
    print(f"{mode}: forward scan, cycles per step of workgroup 0 wave 0 (s_memtime = 100 MHz ticks? see total): total/step {tot / (T + 1):.0f}")
    for n, v in zip(names, acc):
        print(f"   {n:28s} {v / (T + 1):9.1f}  ({100.0 * v / max(tot, 1):5.1f} %)")
g = torch.zeros(spec.param_count, device=dev)
for _ in range(3):
    ops.seq_train_bwd(spec, flat, ws, B, T, rng=dict(seed=1, base_stream=4, p_lstm=0.6, p_head=0.6), grads=g)
    torch.cuda.synchronize()
acc = ws[256:384].cpu().numpy().view(np.int32)[4:20].view(np.uint64)
tot = acc.sum()
print(f"backward scan: total/step {tot / (T + 1):.0f}")
for n, v in zip(["poll", "partial-sum loads + add + consume counter", "cells (dh-dependent part)", "own da -> LDS + barrier", "requests (consume counters, da rows, saved set) + MFMA stream (48) with conversions and ring stores", "-", "drain + flag", "cell factors of the step (top of the loop)"], acc):
    print(f"   {n:50s} {v / (T + 1):9.1f}  ({100.0 * v / max(tot, 1):5.1f} %)")

pa = ws[256:384].cpu().numpy().view(np.int32)[20:32].view(np.uint64)
print("   MFMA stream by pass (cycles / step):", " ".join(f"{v / (T + 1):7.1f}" for v in pa))

# ---- single-layer kernels (cfg5 shape: H = 512, bidirectional, 64-trial tiles): backward scan of the LAST launch (layer 0)
if "--cfg5" in sys.argv:
    spec, B, T = ops.ModelSpec(C=64, H=512, L=1 if "--top" in sys.argv else 2, K=5, D=2), 512, 1000     # --top: a one-layer model (the top layer's upstream term: alpha / dscore instead of din)
    flat = (torch.rand(spec.param_count, device=dev) * 2 - 1) / 22
    x = 2.7 * torch.randn(B, T, 64, device=dev)
    y = torch.randint(0, 5, (B,), device=dev, dtype=torch.int32)
    ws = ops.seq_workspace(spec, B, T, dev)
    g = torch.zeros(spec.param_count, device=dev)
    for _ in range(2):
        ops.seq_train_fwd(spec, flat, x, y, ws, rng=dict(seed=1, base_stream=4, p_lstm=0.6, p_head=0.6))
        ops.seq_train_bwd(spec, flat, ws, B, T, rng=dict(seed=1, base_stream=4, p_lstm=0.6, p_head=0.6), grads=g)
        torch.cuda.synchronize()
    acc = ws[256:384].cpu().numpy().view(np.int32)[4:20].view(np.uint64)
    tot = acc.sum()
    print(f"cfg5 backward scan (layer 0, workgroup 0 wave 0): total/step {tot / T:.0f}")
    for n, v in zip(["poll", "partial-sum loads + add", "cell (dh-dependent part)", "own da -> LDS + barrier", "saved-set request + 64 MFMAs + convert + ring stores", "-", "drain + flag", "row-major stores + upstream term + cell factors of the next step"], acc):
        print(f"   {n:66s} {v / T:9.1f}  ({100.0 * v / max(tot, 1):5.1f} %)")
