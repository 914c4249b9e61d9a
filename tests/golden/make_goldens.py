#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (the reference tree is not present on the GPU box):

    python tests/golden/make_goldens.py --reference /root/reference

It imports the reference's model file from where it lies (never copied), loads
the reference checkpoint, and writes small .npz fixtures:

  weights_3class.npz   the checkpoint's state_dict re-serialised (16 tensors, 31 764 fp32)
  real_trials.npz      16 recorded windows [625,8] (EEG_data_collection/*.csv, fed RAW, i.e.
                       without the third-party MindsAI filter) -> logits / probs / argmax,
                       plus per-stage intermediates for the first two windows
  synthetic.npz        seeded synthetic batches (inputs re-generated from numpy RandomState
                       in the tests, only outputs stored) for the BASELINE config shapes
  grads_32x250.npz     CE-loss gradients of all 16 tensors at B=32,T=250 (eval-mode RReLU, no
                       dropout) from the reference class; and with explicit dropout masks /
                       RReLU slopes from the torch composition in oracle/torch_ref.py
  extensions.npz       beyond-reference oracles (residual stack, H=256/K=5 cfg3 shape) =
                       stock torch on CPU, labelled as such
  zscore.npz           normalize_eeg semantics (Frontend/app.py:166-170) via numpy

Inputs that are cheap to regenerate are NOT stored: tests call synth_x()/synth_labels()/
synth_params() from this file, which only use numpy's legacy RandomState (bit-stable).
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

SYNTH_SHAPES = [(32, 250), (256, 250), (1, 625), (7, 33)]
CFG3_STRIDE = 97
# bidirectional goldens (extension, oracle = torch): tag -> (C, H, L, K, B, T)
BIDIR_CASES = {"bi64": (8, 64, 2, 3, 6, 20), "bi128c64": (64, 128, 2, 5, 5, 12), "bi64L1": (8, 64, 1, 3, 4, 9)}
X_STD = 2.7  # dataset global std is 2.73


# ----------------------------------------------------------------------------------------
# deterministic input generators (shared with the tests; numpy only)
# ----------------------------------------------------------------------------------------
def synth_x(B, T, C=8, seed=1234):
    return (X_STD * np.random.RandomState(seed).standard_normal((B, T, C))).astype(np.float32)


def synth_labels(B, K=3, seed=1234):
    return np.random.RandomState(seed + 7919).randint(0, K, size=B).astype(np.int32)


def synth_params(C, H, L, K, F=32, seed=99, D=1):
    """torch-default-like init U(-1/sqrt(fan), 1/sqrt(fan)); LayerNorm (1,0).  name -> array, in torch's state_dict order
    (D = 2: bidirectional, `_reverse` tensors after the forward ones of each layer; layers > 0 and the head see D*H columns)."""
    rs = np.random.RandomState(seed)
    st = {}
    k = 1.0 / np.sqrt(H)
    DH = D * H
    for l in range(L):
        I = C if l == 0 else DH
        for sfx in (("",) if D == 1 else ("", "_reverse")):
            st[f"lstm.weight_ih_l{l}{sfx}"] = rs.uniform(-k, k, (4 * H, I)).astype(np.float32)
            st[f"lstm.weight_hh_l{l}{sfx}"] = rs.uniform(-k, k, (4 * H, H)).astype(np.float32)
            st[f"lstm.bias_ih_l{l}{sfx}"] = rs.uniform(-k, k, (4 * H,)).astype(np.float32)
            st[f"lstm.bias_hh_l{l}{sfx}"] = rs.uniform(-k, k, (4 * H,)).astype(np.float32)
    H = DH
    st["ln.weight"] = (1.0 + 0.1 * rs.standard_normal(H)).astype(np.float32)
    st["ln.bias"] = (0.1 * rs.standard_normal(H)).astype(np.float32)
    st["attn.weight"] = rs.uniform(-k, k, (1, H)).astype(np.float32)
    st["attn.bias"] = rs.uniform(-k, k, (1,)).astype(np.float32)
    st["fc.0.weight"] = rs.uniform(-k, k, (F, H)).astype(np.float32)
    st["fc.0.bias"] = rs.uniform(-k, k, (F,)).astype(np.float32)
    kf = 1.0 / np.sqrt(F)
    st["fc.3.weight"] = rs.uniform(-kf, kf, (K, F)).astype(np.float32)
    st["fc.3.bias"] = rs.uniform(-kf, kf, (K,)).astype(np.float32)
    return st


def counter_masks(B, T, H, F, L=2, p=0.6, seed=2024):
    """Explicit dropout multipliers / RReLU slopes from a numpy stream (goldens only)."""
    rs = np.random.RandomState(seed)
    keep = np.float32(1.0 / (1.0 - p))
    drop_lstm = (rs.random_sample((L - 1, B, T, H)) >= p).astype(np.float32) * keep
    slope = rs.uniform(1.0 / 8.0, 1.0 / 3.0, (B, F)).astype(np.float32)
    drop_head = (rs.random_sample((B, F)) >= p).astype(np.float32) * keep
    return drop_lstm, slope, drop_head


REAL_PICK = {"backgroundnoise": 3, "food": 3, "no": 3, "water": 4, "yes": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()

    import torch
    import torch.nn.functional as Fnn
    sys.dont_write_bytecode = True  # the reference tree is read-only
    sys.path.insert(0, os.path.join(args.reference, "Neuro-Alpha-App", "Utilities"))
    sys.path.insert(0, ROOT)
    import lstm_eeg_model as ref  # the reference module, imported where it lies
    from oracle.torch_ref import TorchRefEEG

    torch.manual_seed(0)
    torch.set_num_threads(1)
    ckpt = os.path.join(args.reference, "DeepLearning", "LSTM_Model", "lstm_classifier_Water_Food_Bg_Noise.pth")
    state = torch.load(ckpt, map_location="cpu", weights_only=True)
    state_np = {k: v.detach().numpy().astype(np.float32) for k, v in state.items()}
    model = ref.EEG_LSTM(input_size=8, hidden_size=48, num_layers=2, num_classes=3, dropout=0.60)
    model.load_state_dict(state, strict=True)
    model.eval()
    np.savez(os.path.join(args.out, "weights_3class.npz"), **state_np)

    tref = TorchRefEEG(8, 48, 2, 3)
    tref.load_reference_state(state)
    tref.eval()

    # ---- real recorded windows, fed raw --------------------------------------------
    files, labels = [], []
    for prefix, n in REAL_PICK.items():
        fs = sorted(glob.glob(os.path.join(args.reference, "EEG_data_collection", prefix + "_*.csv")))[:n]
        files += fs
        labels += [prefix] * len(fs)
    xr = np.stack([np.loadtxt(f, delimiter=",", dtype=np.float64).astype(np.float32) for f in files])
    assert xr.shape == (16, 625, 8), xr.shape
    with torch.no_grad():
        xt = torch.from_numpy(xr)
        lg = model(xt)
        pr = Fnn.softmax(lg, dim=-1)
        h1, _ = model.lstm(xt[:2])
        sc = model.attn(h1).squeeze(-1)
        al = torch.softmax(sc, dim=1)
        pooled = (h1 * al.unsqueeze(-1)).sum(dim=1)
        ln_out = model.ln(pooled)
        fc0_pre = model.fc[0](ln_out)
        want = {}
        lg_t = tref(xt[:2], want=want)
        assert torch.allclose(lg_t, lg[:2], atol=2e-5), (lg_t, lg[:2])
        assert torch.allclose(want["h1"], h1, atol=1e-6)
        # batch-of-1 calls, the shape SimplePredictor.predict uses (lstm_eeg_model.py:93-96)
        lg_single = torch.cat([model(xt[i:i + 1]) for i in range(16)])
    np.savez(os.path.join(args.out, "real_trials.npz"),
             x=xr, label_prefix=np.array(labels), file_stem=np.array([os.path.basename(f)[:-4] for f in files]),
             logits=lg.numpy(), probs=pr.numpy(), argmax=lg.argmax(-1).numpy().astype(np.int32),
             logits_single=lg_single.numpy(),
             h0=want["h0"].numpy(), h1=h1.numpy(), alpha=al.numpy(), pooled=pooled.numpy(),
             ln_out=ln_out.numpy(), fc0_pre=fc0_pre.numpy())

    # ---- synthetic batches at the BASELINE shapes -----------------------------------
    syn = {}
    for (B, T) in SYNTH_SHAPES:
        x = synth_x(B, T)
        with torch.no_grad():
            out = model(torch.from_numpy(x))
        syn[f"logits_{B}x{T}"] = out.numpy()
        syn[f"xsum_{B}x{T}"] = np.array([x.astype(np.float64).sum(), x[0, 0, 0], x[-1, -1, -1]])
    np.savez(os.path.join(args.out, "synthetic.npz"), **syn)

    # ---- gradients at (32,250) ---------------------------------------------------------
    B, T = 32, 250
    x = torch.from_numpy(synth_x(B, T))
    y = torch.from_numpy(synth_labels(B).astype(np.int64))
    model.zero_grad()
    loss = Fnn.cross_entropy(model(x), y)  # eval mode: no dropout, deterministic RReLU slope
    loss.backward()
    gd = {"eval.loss": np.array(loss.item(), np.float32)}
    for k, p in model.named_parameters():
        gd["eval." + k] = p.grad.numpy().copy()
    # explicit masks through the torch composition (extension: oracle = torch)
    dl, sl, dh = counter_masks(B, T, 48, 32)
    tref.zero_grad()
    loss_m = Fnn.cross_entropy(tref(x, torch.from_numpy(dl), torch.from_numpy(sl), torch.from_numpy(dh)), y)
    loss_m.backward()
    gd["masked.loss"] = np.array(loss_m.item(), np.float32)
    for k, g in tref.reference_named_grads().items():
        gd["masked." + k] = g.numpy().copy()
    # sanity: composition == reference class in eval mode
    tref.zero_grad()
    Fnn.cross_entropy(tref(x), y).backward()
    for k, g in tref.reference_named_grads().items():
        ref_g = gd["eval." + k]
        assert np.allclose(g.numpy(), ref_g, rtol=2e-4, atol=2e-6), (k, np.abs(g.numpy() - ref_g).max())
    np.savez(os.path.join(args.out, "grads_32x250.npz"), **gd)

    # ---- extensions: residual stack; cfg3 shape (H=256,K=5) ----------------------------
    ext = {}
    B, T = 5, 40
    st = synth_params(8, 48, 2, 3, seed=7)
    mres = TorchRefEEG(8, 48, 2, 3, residual=True)
    mres.load_reference_state(st)
    mres.eval()
    x = torch.from_numpy(synth_x(B, T, seed=5))
    y = torch.from_numpy(synth_labels(B, seed=5).astype(np.int64))
    lgr = mres(x)
    Fnn.cross_entropy(lgr, y).backward()
    ext["residual.logits"] = lgr.detach().numpy()
    for k, g in mres.reference_named_grads().items():
        ext["residual.grad." + k] = g.numpy().copy()
    # cfg3 shape through the REFERENCE class itself (it is shape-generic)
    st3 = synth_params(8, 256, 2, 5, seed=11)
    m3 = ref.EEG_LSTM(input_size=8, hidden_size=256, num_layers=2, num_classes=5, dropout=0.60)
    m3.load_state_dict({k: torch.from_numpy(v) for k, v in st3.items()}, strict=True)
    m3.eval()
    x3 = torch.from_numpy(synth_x(4, 250, seed=3))
    y3 = torch.from_numpy(synth_labels(4, K=5, seed=3).astype(np.int64))
    lg3 = m3(x3)
    Fnn.cross_entropy(lg3, y3).backward()
    ext["cfg3.logits"] = lg3.detach().numpy()
    for k, p in m3.named_parameters():
        g = p.grad.numpy()
        if g.size > 20000:   # keep the fixture small: strided sample + norm of the big matrices
            ext["cfg3.gradsample." + k] = g.ravel()[::CFG3_STRIDE].copy()
            ext["cfg3.gradnorm." + k] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
        else:
            ext["cfg3.grad." + k] = g.copy()
    # the same cfg3 model at B=16 (a whole MFMA batch tile is exercised from B >= 16 on): logits, loss, all gradients of
    # the small tensors, strided samples + norms of the big ones
    x16 = torch.from_numpy(synth_x(16, 250, seed=4))
    y16 = torch.from_numpy(synth_labels(16, K=5, seed=4).astype(np.int64))
    m3.zero_grad()
    lg16 = m3(x16)
    loss16 = Fnn.cross_entropy(lg16, y16)
    loss16.backward()
    ext["cfg3b16.logits"] = lg16.detach().numpy()
    ext["cfg3b16.loss"] = np.array(loss16.item(), np.float32)
    for k, p in m3.named_parameters():
        g = p.grad.numpy()
        if g.size > 20000:
            ext["cfg3b16.gradsample." + k] = g.ravel()[::CFG3_STRIDE].copy()
            ext["cfg3b16.gradnorm." + k] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
        else:
            ext["cfg3b16.grad." + k] = g.copy()
    # bidirectional (BASELINE cfg5's structure at a small size): the reference class cannot express it; oracle = stock torch
    # nn.LSTM(bidirectional=True) + the reference's own head on the 2H-wide sequence ("extension, oracle = torch").
    # Parameters come from synth_params(..., D=2) (numpy, regenerated in the tests): only outputs are stored.
    for tag, (Cb, Hb, Lb, Kb, Bb, Tb) in BIDIR_CASES.items():
        stb = synth_params(Cb, Hb, Lb, Kb, seed=70 + Hb, D=2)
        lstm = torch.nn.LSTM(Cb, Hb, Lb, batch_first=True, bidirectional=True)
        lstm.load_state_dict({k[len("lstm."):]: torch.from_numpy(v) for k, v in stb.items() if k.startswith("lstm.")}, strict=True)
        t = {k: torch.from_numpy(v).requires_grad_(True) for k, v in stb.items() if not k.startswith("lstm.")}
        xb = torch.from_numpy(synth_x(Bb, Tb, C=Cb, seed=60))
        yb = torch.from_numpy(synth_labels(Bb, K=Kb, seed=60).astype(np.int64))
        out, _ = lstm(xb)
        w = torch.softmax((out @ t["attn.weight"].t()).squeeze(-1) + t["attn.bias"], dim=1)
        pooled = (out * w.unsqueeze(-1)).sum(dim=1)
        z = Fnn.layer_norm(pooled, (2 * Hb,), t["ln.weight"], t["ln.bias"], 1e-5) @ t["fc.0.weight"].t() + t["fc.0.bias"]
        slope = (0.125 + 1.0 / 3.0) / 2.0
        lgb = torch.where(z >= 0, z, z * slope) @ t["fc.3.weight"].t() + t["fc.3.bias"]      # eval-mode RReLU, no dropout
        lossb = Fnn.cross_entropy(lgb, yb)
        lossb.backward()
        ext[f"{tag}.logits"] = lgb.detach().numpy()
        ext[f"{tag}.loss"] = np.array(lossb.item(), np.float32)
        ext[f"{tag}.alpha"] = w.detach().numpy()
        ext[f"{tag}.out_t0_tlast"] = out.detach().numpy()[:, [0, Tb - 1], :]
        grads = {"lstm." + k: v.grad.numpy() for k, v in lstm.named_parameters()}
        grads.update({k: v.grad.numpy() for k, v in t.items()})
        assert list(grads) == list(stb), (list(grads), list(stb))             # torch's state_dict order == synth_params order
        for k, g in grads.items():
            if g.size > 5000:
                ext[f"{tag}.gradsample.{k}"] = g.ravel()[::CFG3_STRIDE].copy()
                ext[f"{tag}.gradnorm.{k}"] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
            else:
                ext[f"{tag}.grad.{k}"] = g.copy()
    # one-layer and three-layer variants of the reference class (num_layers is a ctor kwarg)
    for Lx in (1, 3):
        stl = synth_params(8, 48, Lx, 3, seed=20 + Lx)
        ml = ref.EEG_LSTM(input_size=8, hidden_size=48, num_layers=Lx, num_classes=3, dropout=0.60)
        ml.load_state_dict({k: torch.from_numpy(v) for k, v in stl.items()}, strict=True)
        ml.eval()
        with torch.no_grad():
            ext[f"L{Lx}.logits"] = ml(torch.from_numpy(synth_x(3, 50, seed=30 + Lx))).numpy()
    np.savez(os.path.join(args.out, "extensions.npz"), **ext)

    # ---- z-score: Frontend/app.py:166-170 (numpy, on a [T,C] chunk) ----------------------
    chunk = xr[:10].mean(axis=0)  # an averaged chunk like TrialResult.avg_chunk (tester.py:98)
    mu = chunk.mean(axis=0, keepdims=True)
    sigma = chunk.std(axis=0, keepdims=True) + 1e-6
    np.savez(os.path.join(args.out, "zscore.npz"), chunk=chunk, normalized=(chunk - mu) / sigma)

    sizes = {f: os.path.getsize(os.path.join(args.out, f)) for f in sorted(os.listdir(args.out)) if f.endswith(".npz")}
    print("wrote", sizes)


if __name__ == "__main__":
    main()
