#!/usr/bin/env python3
"""Forward (+ fused head) launch time of the H = 48 training kernels per instantiation (1 / 2 / 4 trials per workgroup) over batch sizes:
where does the four-trial matrix-pipe kernel (nsd_lstm2_fwd48x4.hip) take over?  Runs through the DIAGNOSTIC twin (the product library
cannot pin the instantiation).    python tools/x4_sweep.py [T] [B ...]"""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import nsd_amd
from nsd_amd import _lib, ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
BS = [int(v) for v in sys.argv[2:]] or [64, 128, 256, 320, 384, 512, 768, 1024, 2048]
dev = torch.device("cuda:0")
nsd_amd.load_library()
spec = ops.ModelSpec()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = np.load(os.path.join(root, "tests", "golden", "weights_3class.npz"))
m = nsd_amd.EEG_LSTM().to(dev)
m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=True)
flat = m.flat_parameters()
r = _lib.Rng(1234, 8, 0.6, 0.6)
print(f"T={T}: forward + fused head, us per launch (median of 30 after 10 warm-up launches); columns: 1 / 2 / 4 trials per workgroup")
with _lib.diagnostic_library():
    L = _lib.lib()
    try:
        for B in BS:
            g = torch.Generator().manual_seed(B)
            x = (2.7 * torch.randn(B, T, 8, generator=g)).to(dev)
            y = torch.randint(0, 3, (B,), generator=g).to(torch.int32).to(dev)
            ws = ops.new_workspace(spec, B, T, dev)
            logits = torch.empty((B, 3), device=dev)
            d = spec.dims(B, T)
            st = torch.cuda.current_stream().cuda_stream
            row = []
            for nb in (1, 2, 4):
                ops.force_fwd48(nb)
                evs = []
                for it in range(40):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    rc = L.nsd_lstm_head_train_rng(C.byref(d), flat.data_ptr(), x.data_ptr(), C.byref(r), y.data_ptr(), 1.0 / B, _lib.NSD_FLAG_TRAIN,
                                                   ws.data_ptr(), ws.numel() * ws.element_size(), logits.data_ptr(), st)
                    b.record()
                    assert rc == 0, L.nsd_last_error()
                    evs.append((a, b))
                torch.cuda.synchronize()
                row.append(1e3 * statistics.median(a.elapsed_time(b) for a, b in evs[10:]))
            print(f"B={B:5d}  " + "  ".join(f"{v:8.1f}" for v in row) + f"    trials/s at the best: {B / (min(row) * 1e-6):,.0f}", flush=True)
    finally:
        ops.force_fwd48(0)
