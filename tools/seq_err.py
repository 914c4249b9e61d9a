"""Relative gradient error of the bf16 sequence path vs the fp32 oracle for a few shapes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nsd_amd
from nsd_amd import ops
from oracle import nsd_oracle as orc
from tests.golden.make_goldens import synth_labels, synth_params, synth_x
dev = torch.device("cuda:0")
for H, L, B, T, res, p in [(128, 3, 33, 10, True, 0.0), (128, 3, 256, 10, True, 0.0), (128, 3, 256, 10, True, 0.4), (128, 3, 256, 40, True, 0.4), (64, 3, 200, 25, True, 0.3)]:
    K, F = 5, 32
    d = orc.Dims(C=8, H=H, L=L, K=K)
    spec = ops.ModelSpec(C=8, H=H, L=L, K=K, residual=res)
    st = synth_params(8, H, L, K, seed=H)
    x, y = synth_x(B, T, seed=12), synth_labels(B, K, seed=12)
    kw = {}
    rng = None
    if p > 0:
        seed, base = 0xBEEF, 12
        kw = dict(drop_lstm=orc.dropout_mask(seed, base, p, (L - 1, B, T, H)), rrelu_slope=orc.rrelu_noise(seed, base + 1, (B, F)),
                  drop_head=orc.dropout_mask(seed, base + 2, p, (B, F)))
        rng = dict(seed=seed, base_stream=base, p_lstm=p, p_head=p)
    flat_np = orc.flatten_state(st, d)
    _, g_ref, fw = orc.loss_and_grads(flat_np, x, y, d, residual=res, **kw)
    flat = torch.from_numpy(flat_np).to(dev)
    ws = ops.seq_workspace(spec, B, T, dev)
    lg = ops.seq_train_fwd(spec, flat, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), ws, rng=rng)
    g = ops.seq_train_bwd(spec, flat, ws, B, T, rng=rng).cpu().numpy()
    gg, rr = orc.unflatten(g, d), orc.unflatten(g_ref, d)
    errs = {k: float(np.abs(gg[k] - rr[k]).max() / max(np.abs(rr[k]).max(), 1e-9)) for k in orc.param_names(d) if k != "attn.bias"}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    pre = fw["fc0_pre"]
    print(f"   min |fc0_pre| over the batch: {np.abs(pre).min():.5f}; entries with |pre| < 0.01: {(np.abs(pre) < 0.01).sum()}")
    print(f"H={H} L={L} B={B} T={T} residual={res} p={p}: logits err {np.abs(lg.cpu().numpy() - fw['logits']).max():.4f}; worst grads:",
          ", ".join(f"{k}={v:.3f}" for k, v in worst), flush=True)
