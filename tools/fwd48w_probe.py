#!/usr/bin/env python3
"""The experimental one-wave-per-layer forward (csrc/nsd_lstm2_fwd48w.hip, diagnostic twin, nsd_diag_force_fwd48(8)) against the
product's one-trial forward: saved activations (h, c, gates, in1) of a training launch, then the launch time of nsd_lstm_fwd.

    python tools/fwd48w_probe.py [--B 256] [--T 250]
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
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--T", type=int, default=250)
    args = ap.parse_args()
    import nsd_amd
    from nsd_amd import _lib, ops
    dev = torch.device("cuda:0")
    spec = ops.ModelSpec()
    w = np.load(os.path.join(ROOT, "tests", "golden", "weights_3class.npz"))
    m = nsd_amd.EEG_LSTM()
    m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files})
    m.to(dev)
    flat = m.flat_parameters()
    B, T = args.B, args.T
    g = torch.Generator().manual_seed(0)
    x = (2.7 * torch.randn(B, T, 8, generator=g)).to(dev)
    dl = ops.dropout_mask(1, 0, 0.6, (1, B, T, 48), dev)
    with _lib.diagnostic_library():
        res = {}
        try:
            for nb in (1, 8):
                ops.force_fwd48(nb)
                ws = ops.new_workspace(spec, B, T, dev)
                ws.fill_(float("nan"))
                logits, _ = ops.train_forward(spec, flat, x, ws, drop_lstm=dl)
                torch.cuda.synchronize()
                res[nb] = {r: ops.ws_view(ws, spec, B, T, r).clone() for r in ("hseq", "cseq", "gact", "inseq", "top")}
                res[nb]["logits"] = logits.clone()
                for _ in range(100):
                    ops.train_forward(spec, flat, x, ws, drop_lstm=dl)
                torch.cuda.synchronize()
                import ctypes as C
                L = _lib.lib()
                d = spec.dims(B, T)
                st = torch.cuda.current_stream().cuda_stream
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
                for a, b in ev:
                    a.record()
                    L.nsd_lstm_fwd(C.byref(d), flat.data_ptr(), x.data_ptr(), dl.data_ptr(), _lib.NSD_FLAG_TRAIN, ws.data_ptr(), ws.numel() * 4, st)
                    b.record()
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
                print(f"force_fwd48({nb}): nsd_lstm_fwd {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})  B={B} T={T}", flush=True)
        finally:
            ops.force_fwd48(0)
    for r in res[1]:
        a, b = res[1][r], res[8][r]
        nan_same = bool((torch.isnan(a) == torch.isnan(b)).all())
        err = torch.where(torch.isnan(a), torch.zeros_like(a), (a - b).abs()).max().item()
        print(f"  {r:7s} max |one-trial kernel - one-wave-per-layer kernel| = {err:.3e}   same written region: {nan_same}")


if __name__ == "__main__":
    main()
