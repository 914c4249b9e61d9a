"""One GEMM shape of the path, a few launches, for rocprofv3 counter passes:  python3 tools/micro/gemm_one.py [xproj|din|dw]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nsd_amd
from nsd_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "xproj"
if which == "xproj":        # C tiles [N/32][M/32][64][16] = W[M=2048][K=1024] . in[N=512000][K]^T
    M, N, K = 2048, 512000, 1024
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    c = torch.empty((N // 32, M // 32, 64, 16), dtype=torch.bfloat16, device=dev)
    args = (a.data_ptr(), K, 0, b.data_ptr(), K, 0, 0, c.data_ptr(), N, 2, None, M, N, K, 1)
elif which == "din":        # din[M=512000][N=1024] = da[M][K=4096] . wxt[K][N]
    M, N, K = 512000, 1024, 4096
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    c = torch.empty((M, N), dtype=torch.float32, device=dev)
    args = (a.data_ptr(), K, 0, b.data_ptr(), N, 1, 0, c.data_ptr(), N, 0, None, M, N, K, 1)
else:                       # dW[M=2048][N=1024] = da[K=512000][M]^T . in[K][N], 8 splits
    M, N, K = 2048, 1024, 512000
    a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    c = torch.empty((8, M, N), dtype=torch.float32, device=dev)
    args = (a.data_ptr(), M, 1, b.data_ptr(), N, 1, 0, c.data_ptr(), N, 0, None, M, N, K, 8)
for _ in range(4):
    ops._call("nsd_gemm_bf16", dev, *args, ops.STREAM)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    ops._call("nsd_gemm_bf16", dev, *args, ops.STREAM)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 4
print(f"{which}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
