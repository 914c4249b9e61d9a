#!/usr/bin/env python3
"""A few train steps at the BASELINE cfg3 shape (H=256, K=5, B=1024, T=250) -- for rocprofv3 --kernel-trace --stats."""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsd_amd
from nsd_amd.trainer import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
m = nsd_amd.EEG_LSTM(8, 256, 2, 5, dropout=0.6).to(dev).train()
tr = Trainer(m, lr=1e-3, seed=1)
g = torch.Generator().manual_seed(0)
x = (2.7 * torch.randn(B, 250, 8, generator=g)).to(dev); y = torch.randint(0, 5, (B,), generator=g).to(torch.int32).to(dev)
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): tr.step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"cfg3 B={B}: {dt*1e3:.2f} ms/step  {B/dt:.0f} trials/s")
from nsd_amd import ops
m.eval()
flat = m.flat_parameters()
for _ in range(2): ops.infer(m.spec, flat, x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ops.infer(m.spec, flat, x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"cfg3 B={B} inference: {dt*1e3:.2f} ms  {B/dt:.0f} windows/s")
ops.set_gemm_bf16(True)
m.train()
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): tr.step(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"cfg3 B={B} bf16 GEMM operands: {dt*1e3:.2f} ms/step  {B/dt:.0f} trials/s")
m.eval()
for _ in range(2): ops.infer(m.spec, flat, x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ops.infer(m.spec, flat, x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"cfg3 B={B} bf16 inference: {dt*1e3:.2f} ms  {B/dt:.0f} windows/s")
