"""The CPU oracle (oracle/nsd_oracle.c) pinned to vectors captured from the reference
(tests/golden/*.npz, made by tests/golden/make_goldens.py)."""
import numpy as np
import pytest

from oracle import nsd_oracle as orc
from tests.golden.make_goldens import (CFG3_STRIDE, SYNTH_SHAPES, counter_masks, synth_labels,
                                       synth_params, synth_x)

D = orc.Dims()          # C=8,H=48,L=2,K=3,F=32
LOGIT_TOL = 1e-4        # north_star: class logits within 1e-4 fp32, argmax bit-exact


def test_param_layout_matches_reference_state_dict(ref_state):
    assert orc.param_count(D) == 31764 == sum(v.size for v in ref_state.values())
    assert list(ref_state.keys()) == orc.param_names(D)      # same order as the checkpoint
    shp = orc.param_shapes(D)
    for k, v in ref_state.items():
        assert tuple(v.shape) == shp[k]
    flat = orc.flatten_state(ref_state, D)
    back = orc.unflatten(flat, D)
    for k in ref_state:
        assert np.array_equal(back[k], ref_state[k])


def test_real_trials_logits_probs_argmax(golden, ref_state):
    g = golden("real_trials")
    flat = orc.flatten_state(ref_state, D)
    out = orc.forward(flat, g["x"], D, saves=True)
    assert np.abs(out["logits"] - g["logits"]).max() < LOGIT_TOL
    assert np.abs(out["logits"] - g["logits_single"]).max() < LOGIT_TOL
    assert np.abs(out["probs"] - g["probs"]).max() < 1e-5
    assert np.array_equal(out["logits"].argmax(-1), g["argmax"])
    # per-stage intermediates for the first two windows
    assert np.abs(out["hseq"][0, :2] - g["h0"]).max() < 2e-5
    assert np.abs(out["hseq"][1, :2] - g["h1"]).max() < 2e-5
    assert np.abs(out["alpha"][:2] - g["alpha"]).max() < 1e-6
    assert np.abs(out["pooled"][:2] - g["pooled"]).max() < 1e-5
    assert np.abs(out["ln_out"][:2] - g["ln_out"]).max() < 5e-5
    assert np.abs(out["fc0_pre"][:2] - g["fc0_pre"]).max() < 5e-5


@pytest.mark.parametrize("B,T", SYNTH_SHAPES)
def test_synthetic_shapes(golden, ref_state, B, T):
    g = golden("synthetic")
    x = synth_x(B, T)
    chk = g[f"xsum_{B}x{T}"]
    assert x.astype(np.float64).sum() == chk[0] and x[0, 0, 0] == np.float32(chk[1])  # generator drift guard
    out = orc.forward(orc.flatten_state(ref_state, D), x, D)
    ref = g[f"logits_{B}x{T}"]
    assert np.abs(out["logits"] - ref).max() < LOGIT_TOL
    assert np.array_equal(out["logits"].argmax(-1), ref.argmax(-1))


def _cmp_grads(flat_g, named_ref, d, prefix, rtol=2e-4):
    got = orc.unflatten(flat_g, d)
    for k in orc.param_names(d):
        r = named_ref[prefix + k]
        scale = max(np.abs(r).max(), 1e-6)
        err = np.abs(got[k] - r).max()
        if k == "attn.bias":      # analytically zero (softmax shift invariance): absolute tolerance
            assert err < 1e-6, (k, err)
        else:
            assert err <= rtol * scale + 1e-7, (k, err, scale)


def test_gradients_eval_mode_vs_reference(golden, ref_state):
    g = golden("grads_32x250")
    flat = orc.flatten_state(ref_state, D)
    x, y = synth_x(32, 250), synth_labels(32)
    loss, grads, _ = orc.loss_and_grads(flat, x, y, D)
    assert abs(loss - float(g["eval.loss"])) < 2e-5
    _cmp_grads(grads, g, D, "eval.")


def test_gradients_with_explicit_masks_vs_torch_composition(golden, ref_state):
    g = golden("grads_32x250")
    flat = orc.flatten_state(ref_state, D)
    x, y = synth_x(32, 250), synth_labels(32)
    dl, sl, dh = counter_masks(32, 250, 48, 32)
    loss, grads, _ = orc.loss_and_grads(flat, x, y, D, drop_lstm=dl, rrelu_slope=sl, drop_head=dh)
    assert abs(loss - float(g["masked.loss"])) < 5e-5
    _cmp_grads(grads, g, D, "masked.")


def test_extension_residual(golden):
    e = golden("extensions")
    flat = orc.flatten_state(synth_params(8, 48, 2, 3, seed=7), D)
    x, y = synth_x(5, 40, seed=5), synth_labels(5, seed=5)
    loss, grads, fw = orc.loss_and_grads(flat, x, y, D, residual=True)
    assert np.abs(fw["logits"] - e["residual.logits"]).max() < 2e-5
    _cmp_grads(grads, e, D, "residual.grad.")


def test_extension_cfg3_shape(golden):
    e = golden("extensions")
    d3 = orc.Dims(C=8, H=256, L=2, K=5, F=32)
    assert orc.param_count(d3) == 807878
    flat = orc.flatten_state(synth_params(8, 256, 2, 5, seed=11), d3)
    x, y = synth_x(4, 250, seed=3), synth_labels(4, K=5, seed=3)
    loss, grads, fw = orc.loss_and_grads(flat, x, y, d3)
    assert np.abs(fw["logits"] - e["cfg3.logits"]).max() < LOGIT_TOL
    got = orc.unflatten(grads, d3)
    for k in orc.param_names(d3):
        if "cfg3.grad." + k in e.files:
            r = e["cfg3.grad." + k]
            tol = 1e-6 if k == "attn.bias" else 3e-4 * max(np.abs(r).max(), 1e-6) + 1e-7
            assert np.abs(got[k] - r).max() <= tol, k
        else:
            r = e["cfg3.gradsample." + k]
            s = got[k].ravel()[::CFG3_STRIDE]
            assert np.abs(s - r).max() <= 3e-4 * np.abs(r).max() + 1e-7, k
            nrm = np.sqrt((got[k].astype(np.float64) ** 2).sum())
            assert abs(nrm - float(e["cfg3.gradnorm." + k])) <= 3e-4 * nrm, k


@pytest.mark.parametrize("L", [1, 3])
def test_extension_layer_counts(golden, L):
    e = golden("extensions")
    d = orc.Dims(L=L)
    flat = orc.flatten_state(synth_params(8, 48, L, 3, seed=20 + L), d)
    out = orc.forward(flat, synth_x(3, 50, seed=30 + L), d)
    assert np.abs(out["logits"] - e[f"L{L}.logits"]).max() < 2e-5


def test_zscore_matches_normalize_eeg(golden):
    z = golden("zscore")
    got = orc.zscore(z["chunk"])
    assert np.abs(got - z["normalized"]).max() < 2e-5


def test_adam_matches_torch():
    import torch
    rs = np.random.RandomState(0)
    p0 = rs.standard_normal(1000).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([p], lr=1e-3)
    po, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = rs.standard_normal(1000).astype(np.float32)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        orc.adam(po, g, m, v, lr=1e-3, step=step)
    assert np.abs(po - p.detach().numpy()).max() < 1e-6


def test_counter_rng_properties():
    m = orc.dropout_mask(5, 0, 0.6, (200000,))
    keep_frac = (m > 0).mean()
    assert abs(keep_frac - 0.4) < 0.01 and np.isclose(m.max(), 2.5)
    assert not np.array_equal(m, orc.dropout_mask(5, 1, 0.6, (200000,)))   # streams differ
    assert np.array_equal(m, orc.dropout_mask(5, 0, 0.6, (200000,)))       # deterministic
    s = orc.rrelu_noise(5, 2, (100000,))
    assert s.min() >= 0.125 and s.max() <= 1 / 3 + 1e-7 and abs(s.mean() - (0.125 + 1 / 3) / 2) < 1e-3


def test_cpu_baseline_module_is_the_reference_model(golden, ref_state):
    """bench.py's cpu_baseline times oracle.torch_ref.StackedTorchEEG: the reference module's structure re-declared from stock
    torch (the reference tree does not travel to the GPU box).  It must carry the reference's state_dict keys and give the
    reference's logits on the recorded windows; its bidirectional form must give the torch bidirectional goldens."""
    import torch
    from oracle.torch_ref import StackedTorchEEG
    from tests.golden.make_goldens import BIDIR_CASES, synth_params, synth_x
    m = StackedTorchEEG().eval()
    assert list(m.state_dict().keys()) == list(ref_state.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ref_state.items()}, strict=True)
    g = golden("real_trials")
    with torch.no_grad():
        lg = m(torch.from_numpy(g["x"])).numpy()
    assert np.abs(lg - g["logits"]).max() < 2e-5
    ext = golden("extensions")
    for tag, (C, H, L, K, B, T) in BIDIR_CASES.items():
        st = synth_params(C, H, L, K, seed=70 + H, D=2)
        mb = StackedTorchEEG(C, H, L, K, bidirectional=True).eval()
        assert list(mb.state_dict().keys()) == list(st.keys())
        mb.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        with torch.no_grad():
            lgb = mb(torch.from_numpy(synth_x(B, T, C=C, seed=60))).numpy()
        assert np.abs(lgb - ext[f"{tag}.logits"]).max() < 2e-5, tag


def test_torch_composition_bidirectional_equals_stock_stacked_lstm(golden):
    """oracle.torch_ref.TorchRefEEG(bidirectional=True) -- L one-layer bidirectional nn.LSTMs, explicit masks between them: the
    oracle of the bidirectional train-mode GPU tests -- reproduces the stock stacked nn.LSTM(bidirectional=True) goldens."""
    import torch
    from oracle.torch_ref import TorchRefEEG
    from tests.golden.make_goldens import BIDIR_CASES, synth_labels, synth_params, synth_x
    ext = golden("extensions")
    for tag, (C, H, L, K, B, T) in BIDIR_CASES.items():
        st = synth_params(C, H, L, K, seed=70 + H, D=2)
        m = TorchRefEEG(C, H, L, K, bidirectional=True).eval()
        m.load_reference_state({k: torch.from_numpy(v) for k, v in st.items()})
        x = torch.from_numpy(synth_x(B, T, C=C, seed=60))
        y = torch.from_numpy(synth_labels(B, K=K, seed=60).astype(np.int64))
        lg = m(x)
        torch.nn.functional.cross_entropy(lg, y).backward()
        assert np.abs(lg.detach().numpy() - ext[f"{tag}.logits"]).max() < 2e-5
        for k, g in m.reference_named_grads().items():
            if f"{tag}.grad.{k}" in ext.files:
                ref = ext[f"{tag}.grad.{k}"]
                assert np.abs(g.numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6) + 2e-6, (tag, k)
