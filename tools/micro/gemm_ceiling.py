"""What the vendor BLAS (through torch.matmul) reaches on the GEMM shapes of the sequence-batched path, next to nsd_gemm_bf16 --
a ceiling estimate for csrc/nsd_gemm_bf16.hip, not a product path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nsd_amd
from nsd_amd import ops
dev = torch.device("cuda:0")

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for name, R, G, N in [("cfg3 dW_hh", 256000, 1024, 256), ("cfg5 dW_hh", 512000, 2048, 512), ("cfg5 dW_ih1", 512000, 2048, 1024)]:
    da = torch.randn(R, G, device=dev, dtype=torch.bfloat16)
    h = torch.randn(R, N, device=dev, dtype=torch.bfloat16)
    ms_t = t(lambda: torch.matmul(da.t(), h))
    fl = 2.0 * R * G * N
    line = f"{name:12s} K={R} M={G} N={N}: torch {ms_t:7.3f} ms = {fl / ms_t / 1e9:7.1f} TFLOP/s | nsd_gemm_bf16"
    for sp in (max(1, min(64, 512 // ((G // 128) * max(1, N // 128)))), max(1, min(64, -(-256 // ((G // 256) * max(1, N // 256)))))):   # 128-tile rule / one 256-tile workgroup per CU
        c = torch.empty((sp, G, N), dtype=torch.float32, device=dev)
        ms_n = t(lambda: ops._call("nsd_gemm_bf16", dev, da.data_ptr(), G, 1, h.data_ptr(), N, 1, 0, c.data_ptr(), N, 0, None, G, N, R, sp, ops.STREAM))
        line += f"  splits {sp}: {ms_n:7.3f} ms = {fl / ms_n / 1e9:7.1f} TFLOP/s"
    print(line, flush=True)
    del da, h
for name, R, K, N in [("cfg5 xproj1", 512000, 1024, 2048), ("cfg5 din", 512000, 4096, 1024)]:
    a = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    ms_t = t(lambda: torch.matmul(a, w.t()))
    fl = 2.0 * R * K * N
    c = torch.empty((R, N), dtype=torch.bfloat16, device=dev)
    ms_n = t(lambda: ops._call("nsd_gemm_bf16", dev, a.data_ptr(), K, 0, w.data_ptr(), K, 0, 0, c.data_ptr(), N, 1, None, R, N, K, 1, ops.STREAM))
    print(f"{name:12s} M={R} K={K} N={N}: torch {ms_t:7.3f} ms = {fl / ms_t / 1e9:7.1f} TFLOP/s | nsd_gemm_bf16 (bf16 out) {ms_n:7.3f} ms = {fl / ms_n / 1e9:7.1f} TFLOP/s", flush=True)
    del a, w
