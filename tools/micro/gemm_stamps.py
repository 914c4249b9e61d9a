"""Phase stamps of the LDS-DMA GEMM kernel (ablation build with NSD_GEMM_ABL bit 8): NSD_LIB=libnsd_hip_gabl8.so python tools/micro/gemm_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nsd_amd
from nsd_amd import ops, _lib
dev = torch.device("cuda:0")
for name, R, K, N in [("xproj1", 512000, 1024, 2048), ("din", 512000, 4096, 1024)]:
    a = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    c = torch.empty((R, N), dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops._call("nsd_gemm_bf16", dev, a.data_ptr(), K, 0, w.data_ptr(), K, 0, 0, c.data_ptr(), N, 1, None, R, N, K, 1, ops.STREAM)
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 8)()
    L = _lib.lib()
    L.nsd_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
    L.nsd_debug_gemm_stamps(out)
    nt = max(int(out[4]), 1)
    print(name, "chunks", nt, "ticks per chunk (s_memtime, 100 MHz):", [round(int(out[i]) / nt, 2) for i in range(4)], "= DMA issue | reads + MFMAs | DMA wait | barrier")
