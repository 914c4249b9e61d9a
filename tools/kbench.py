#!/usr/bin/env python3
"""Per-kernel timing of the LSTM forward / backward launches (HIP events), optionally under the
NSD_ABLATE timing switches of csrc (results are wrong under ablation; only the clock matters).

    python tools/kbench.py [--B 256] [--T 250] [--iters 20]
    NSD_LIB=libnsd_hip_prof.so python tools/kbench.py --ablate 0,1,64,128 [--prof]     (switches exist in the diagnostic build only)
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ablate", default="0")
    ap.add_argument("--mask", type=int, default=1)
    ap.add_argument("--prof", action="store_true")
    ap.add_argument("--rng", action="store_true", help="the train-mode streams drawn in the kernels (what bench.py's step runs) instead of explicit tensors")
    args = ap.parse_args()
    import nsd_amd
    from nsd_amd import ops
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
    y = torch.randint(0, 3, (B,), generator=g).to(torch.int32).to(dev)
    ws = ops.new_workspace(spec, B, T, dev)
    dl = ops.dropout_mask(1, 0, 0.6, (1, B, T, 48), dev) if args.mask else None

    def timed(fn, n):
        for _ in range(100):       # the GPU clock ramps over the first tens of launches
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return ts[len(ts) // 2], ts[0]

    import ctypes as C
    from nsd_amd import _lib
    L = _lib.lib()
    d = spec.dims(B, T)
    st = torch.cuda.current_stream().cuda_stream
    flags = _lib.NSD_FLAG_TRAIN
    pp, xp, wsp, wsn = flat.data_ptr(), x.data_ptr(), ws.data_ptr(), ws.numel() * 4
    dlp = dl.data_ptr() if dl is not None else None
    for ab in [int(v) for v in args.ablate.split(",")]:
        os.environ["NSD_ABLATE"] = str(ab)
        logits, _ = ops.train_forward(spec, flat, x, ws, drop_lstm=dl)
        ops.train_backward(spec, flat, x, ws, logits, labels=y, drop_lstm=dl)
        rng = _lib.Rng(0x1234ABCD, 44, 0.6, 0.6)
        lg0 = torch.empty(B, 3, device=dev)
        if args.rng:      # leave the workspace as the fused forward of the step leaves it
            L.nsd_lstm_head_train_rng(C.byref(d), pp, xp, C.byref(rng), y.data_ptr(), 1.0 / B, flags, wsp, wsn, lg0.data_ptr(), st)
            bwd = lambda: L.nsd_lstm_bwd_rng(C.byref(d), pp, xp, C.byref(rng), flags, wsp, wsn, st)
        else:
            bwd = lambda: L.nsd_lstm_bwd(C.byref(d), pp, xp, dlp, flags, wsp, wsn, None, st)
        f_med, f_min = timed(lambda: L.nsd_lstm_fwd(C.byref(d), pp, xp, dlp, flags, wsp, wsn, st), args.iters)
        if args.rng:
            L.nsd_lstm_head_train_rng(C.byref(d), pp, xp, C.byref(rng), y.data_ptr(), 1.0 / B, flags, wsp, wsn, lg0.data_ptr(), st)
        b_med, b_min = timed(bwd, args.iters)
        sl = ops.rrelu_noise(1, 1, (B, 32), dev); dh = ops.dropout_mask(1, 2, 0.6, (B, 32), dev)
        lg = torch.empty(B, 3, device=dev)
        if args.rng:
            h_med, h_min = timed(lambda: L.nsd_lstm_head_train_rng(C.byref(d), pp, xp, C.byref(rng), y.data_ptr(), 1.0 / B, flags, wsp, wsn, lg.data_ptr(), st), args.iters)
        elif hasattr(L, "nsd_lstm_head_train"):
            h_med, h_min = timed(lambda: L.nsd_lstm_head_train(C.byref(d), pp, xp, dlp, sl.data_ptr(), dh.data_ptr(), y.data_ptr(),
                                                               1.0 / B, flags, wsp, wsn, lg.data_ptr(), st), args.iters)
        else:
            h_med = h_min = float("nan")
        print(f"ablate={ab:3d}  B={B} T={T}  lstm_fwd {f_med:8.1f} us (min {f_min:.1f})   lstm_bwd {b_med:8.1f} us (min {b_min:.1f})   "
              f"lstm_head_train {h_med:8.1f} us (min {h_min:.1f})", flush=True)
        if args.prof:
            for which in ("fwd", "bwd"):
                dbg = torch.zeros(512, dtype=torch.int64, device=dev)
                L.nsd_debug_profile_buffer(dbg.data_ptr())
                if which == "fwd":
                    L.nsd_lstm_fwd(C.byref(d), pp, xp, dlp, flags, wsp, wsn, st)
                    # role table of lstm2_fwd48_kernel: SIMD g = wave & 3, slot q = wave >> 2
                    roles = ["L1", "L1", "L1", "P", "L0", "L0", "L0", "P", "saver", "spare", "spare", "P"]
                    nst = ((T + 2 + 31) // 32) * 32
                    if B >= 513:                     # four trials per workgroup (nsd_lstm2_fwd48x4.hip): stage / pool helpers, 16-step chunks
                        roles = ["L1", "L1", "L1", "P", "L0", "L0", "L0", "P", "stage", "idle", "idle", "P"]
                        nst = ((T + 2 + 15) // 16) * 16
                else:
                    bwd()
                    roles = ["chain1"] * 3 + ["chain0"] * 3 + ["x1+p1", "x1+p0", "x1", "dW4+cv", "dW2", "dW3", "dW0", "dW1", "dW5+cv", "loader"]   # one trial per workgroup: role map of lstm2_bwd48_kernel<1> (p: prepares a layer's factors, cv: converts a layer's da)
                    nst = 4 * ((((T + 2) // 4 + 1) + 1) & ~1)
                    if B >= 513:                     # four trials per workgroup (nsd_lstm2_bwd48x4.hip): role = f(wave & 3, wave >> 2)
                        roles = ["C1", "C1", "C1", "X1", "C0", "C0", "C0", "X1", "dW1", "dW1", "dW1", "X1", "dW0", "dW0", "rows", "aux"]
                        nst = ((T + 3 + 15) // 16) * 16
                torch.cuda.synchronize()
                L.nsd_debug_profile_buffer(None)
                hw = dbg.cpu().numpy()[256:]
                v = dbg.cpu().numpy()[:256].reshape(32, 8)
                print(f"  {which}: cycles per step (workgroup 0)")
                for wv, role in enumerate(roles):
                    seg = "  ".join(f"{v[wv, 2 + q] / nst:5.0f}" for q in range(6))
                    print(f"    wave {wv:2d} {role:7s} work {v[wv,0]/nst:6.0f}  wait {v[wv,1]/nst:6.0f}   segments {seg}   simd {(int(hw[wv]) >> 4) & 3} slot {int(hw[wv]) & 15} cu {(int(hw[wv]) >> 8) & 15}")
    os.environ["NSD_ABLATE"] = "0"


if __name__ == "__main__":
    main()
