"""Time the fused two-layer scans of cfg3 (train forward, inference, backward) with the library NSD_LIB names.
    NSD_LIB=libnsd_hip_var_x.so python tools/micro/scan2_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nsd_amd
from nsd_amd import ops
spec, B, T = ops.ModelSpec(C=8, H=256, L=2, K=5), 1024, 250
dev = torch.device("cuda:0")
torch.manual_seed(0)
flat = (torch.rand(spec.param_count, device=dev) * 2 - 1) / 16
x = 2.7 * torch.randn(B, T, 8, device=dev)
y = torch.randint(0, 5, (B,), device=dev, dtype=torch.int32)
ws = ops.seq_workspace(spec, B, T, dev)
g = torch.zeros(spec.param_count, device=dev)
rng = dict(seed=1, base_stream=4, p_lstm=0.6, p_head=0.6)
def timed(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
t_inf = timed(lambda: ops.seq_infer(spec, flat, x, ws, want_probs=False))
t_fwd = timed(lambda: ops.seq_train_fwd(spec, flat, x, y, ws, rng=rng))
t_bwd = timed(lambda: ops.seq_train_bwd(spec, flat, ws, B, T, rng=rng, grads=g))
print(f"{os.environ.get('NSD_LIB', 'product'):34s} infer {t_inf:8.1f} us   train fwd (scan + head) {t_fwd:8.1f} us   bwd (all) {t_bwd:8.1f} us", flush=True)
