"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/nsd.h declares,
layouts agree with the oracle, the Python facade keeps the reference's surface, the producer protocol
works, and the product path refuses to run without the HIP device (no fallback)."""
import os
import re
import time
from multiprocessing import Queue

import numpy as np
import pytest
import torch

import nsd_amd
from nsd_amd import _lib, ops
from oracle import nsd_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nsd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nsd_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    L = nsd_amd.load_library()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in nsd.h but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert L.nsd_version() == 300
    assert not hasattr(L, "nsd_debug_profile_buffer")           # diagnostics are not in the shipped library


def test_diagnostics_live_only_in_the_diagnostic_build():
    """The exchange-mode / forced time-out flag bits and the per-kernel timing entry points (csrc/nsd_diag.h) are not in
    include/nsd.h, not exported by libnsd_hip.so and rejected by it; libnsd_hip_diag.so (same objects, orchestration compiled with
    -DNSD_DIAG=1) has them.  No product module asks for the diagnostic library."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "nsd.h")).read()
    for word in ("nsd_seq_profile", "NO_L2_EXCHANGE", "SPREAD_GROUPS", "NO_FUSED_LAYERS", "LOSE_MEMBER"):
        assert word not in hdr, word
    L = nsd_amd.load_library()
    assert not hasattr(L, "nsd_seq_profile") and not hasattr(L, "nsd_seq_profile_read")
    d = _lib.Dims(64, 10, 8, 256, 2, 5, 32)
    assert L.nsd_seq_supported(C.byref(d), 0) == 1 and L.nsd_seq_supported(C.byref(d), _lib.NSD_FLAG_BIDIR) == 1
    for bit in (_lib.NSD_DIAG_FLAG_NO_L2_EXCHANGE, _lib.NSD_DIAG_FLAG_SPREAD_GROUPS, _lib.NSD_DIAG_FLAG_NO_FUSED_LAYERS,
                _lib.NSD_DIAG_FLAG_LOSE_MEMBER, 1 << 20):
        assert L.nsd_seq_supported(C.byref(d), bit) == 0
        assert L.nsd_seq_workspace_bytes(C.byref(d), bit) == -1 and b"unknown flag" in L.nsd_last_error()
    with _lib.diagnostic_library() as DL:
        assert DL is not L and hasattr(DL, "nsd_seq_profile") and DL.nsd_version() == 300
        assert DL.nsd_seq_supported(C.byref(d), _lib.NSD_DIAG_FLAG_SPREAD_GROUPS) == 1
        assert ops.ModelSpec(H=256, K=5).param_count == 807878        # calls inside the block go to the diagnostic library
        with pytest.raises(nsd_amd.NsdError):
            with _lib.diagnostic_library():
                pass
    assert _lib.lib() is L and not _lib.diag_active()
    with pytest.raises(nsd_amd.NsdError, match="diagnostic build"):
        ops.seq_profile(True)
    pkg = os.path.join(ROOT, "neural-speech-decoding_amd")
    for fn in ("lstm_eeg_model.py", "trainer.py", "train.py", "tester.py", "streaming_process.py", "data.py", "__init__.py"):
        assert "diagnostic_library" not in open(os.path.join(pkg, fn)).read(), fn


def test_seq_failure_reporting_entry_points_without_a_gpu():
    import ctypes as C
    L = nsd_amd.load_library()
    assert L.nsd_seq_workspace_init(None, 1 << 20, None) == -1
    assert L.nsd_seq_workspace_init(4096, 16, None) == -3          # smaller than the persistent header
    assert L.nsd_seq_guard(None, None, None) == -1
    assert L.nsd_adam_step_guarded(8, 4096, 4096, 4096, 4096, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 1, None, None) == -1   # skip flag is mandatory
    # the workspace starts with the 256-byte persistent header (sticky status), the per-evaluation status words follow
    src = open(os.path.join(ROOT, "neural-speech-decoding_amd", "csrc", "nsd_seq.h")).read()
    assert "#define NSD_SEQ_HEADER_BYTES 256" in src


@pytest.mark.parametrize("dims", [orc.Dims(), orc.Dims(H=256, K=5), orc.Dims(C=64, H=512, L=3, K=5), orc.Dims(L=1)])
def test_param_layout_matches_oracle_and_reference_order(dims):
    spec = ops.ModelSpec(C=dims.C, H=dims.H, L=dims.L, K=dims.K, F=dims.F)
    assert spec.param_count == orc.param_count(dims)
    assert spec.offsets() == orc.layout(dims)
    assert spec.names() == orc.param_names(dims) and spec.shapes() == orc.param_shapes(dims)


def test_reference_sizes():
    assert ops.ModelSpec().param_count == 31764
    assert ops.ModelSpec(H=256, K=5).param_count == 807878
    assert ops.ModelSpec().fast_path() and not ops.ModelSpec(H=256).fast_path()
    assert ops.ModelSpec(C=0).dims(1, 1).C == 0
    with pytest.raises(nsd_amd.NsdError):
        _ = ops.ModelSpec(C=0).param_count


def test_workspace_layout_is_disjoint_and_aligned():
    spec = ops.ModelSpec()
    nbytes, w = ops.workspace_layout(spec, 256, 250)
    regs = ["hseq", "cseq", "gact", "inseq", "top", "alpha", "pooled", "fc0_pre", "dscore", "dpooled", "loss", "adpack", "slabs", "hslabs"]
    offs = [getattr(w, r) for r in regs]
    assert offs == sorted(offs) and all(o % 4 == 0 for o in offs) and nbytes == 4 * w.total
    B, T, H = 256, 250, 48
    assert w.cseq - w.hseq >= 2 * B * T * H and w.inseq - w.gact >= 8 * B * T * H
    assert w.total - w.hslabs >= B * (31764 - 29952)
    # algorithmic bytes/trial of the training path: h and c per layer-step, written once + read once
    assert 2 * (2 * T * 2 * H * 4) + 2 * T * 8 * 4 + 12 == 400012


def test_bad_arguments_are_rejected_without_a_gpu():
    L = nsd_amd.load_library()
    import ctypes as C
    d = _lib.Dims(4, 10, 8, 48, 2, 3, 32)
    assert L.nsd_infer(C.byref(d), None, None, 0, None, None, None, None) == -1
    assert b"null" in L.nsd_last_error()
    bad = _lib.Dims(4, 0, 8, 48, 2, 3, 32)
    assert L.nsd_workspace_bytes(C.byref(bad), None) < 0
    assert L.nsd_adam_step(-1, None, None, None, None, 0, 0, 0, 0, 0, 1, 1, None) == -1
    assert L.nsd_param_count(8, 48, 9, 3, 32) < 0                  # more than NSD_MAX_LAYERS
    # a short workspace is refused before anything is launched (NSD_E_WORKSPACE), for every entry point that touches it
    need = L.nsd_workspace_bytes(C.byref(d), None)
    fake = 4096                                                    # never dereferenced: the size check comes first
    E_WS = -3
    assert L.nsd_lstm_fwd(C.byref(d), fake, fake, None, 2, fake, need - 4, None) == E_WS
    assert b"smaller than nsd_workspace_bytes" in L.nsd_last_error()
    assert L.nsd_head_fwd(C.byref(d), fake, None, None, fake, need - 4, fake, None, None) == E_WS
    assert L.nsd_head_bwd(C.byref(d), fake, None, None, fake, fake, None, 1.0, fake, 0, None) == E_WS
    assert L.nsd_head_train(C.byref(d), fake, None, None, fake, 1.0, fake, need - 4, fake, None) == E_WS
    assert L.nsd_lstm_head_train(C.byref(d), fake, fake, None, None, None, fake, 1.0, 2, fake, need - 4, fake, None) == E_WS
    assert L.nsd_lstm_bwd(C.byref(d), fake, fake, None, 2, fake, need - 4, None, None) == E_WS
    assert L.nsd_grad_reduce(C.byref(d), fake, need - 4, fake, 0, None) == E_WS
    assert L.nsd_grad_reduce_adam(C.byref(d), fake, need - 4, fake, fake, fake, fake, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 1, None) == E_WS
    assert L.nsd_loss_sum(C.byref(d), fake, need - 4, fake, None) == E_WS
    rng = _lib.Rng(1, 4, 0.6, 0.6)
    assert L.nsd_lstm_head_train_rng(C.byref(d), fake, fake, C.byref(rng), fake, 1.0, 2, fake, need - 4, fake, None) == E_WS
    assert L.nsd_lstm_bwd_rng(C.byref(d), fake, fake, C.byref(rng), 2, fake, need - 4, None) == E_WS
    # dx is formed for H = 48 and on the generic path; elsewhere a non-NULL dx is refused before anything is launched
    d64 = _lib.Dims(4, 10, 8, 64, 2, 3, 32)
    need64 = L.nsd_workspace_bytes(C.byref(d64), None)
    assert L.nsd_lstm_bwd(C.byref(d64), fake, fake, None, 2, fake, need64, fake, None) == -1
    assert b"dx is available for H = 48" in L.nsd_last_error()


def test_facade_surface_matches_reference(ref_state):
    import inspect
    sig = inspect.signature(nsd_amd.EEG_LSTM.__init__)
    assert [(k, v.default) for k, v in list(sig.parameters.items())[1:6]] == [
        ("input_size", 8), ("hidden_size", 48), ("num_layers", 2), ("num_classes", 3), ("dropout", 0.60)]
    sp = inspect.signature(nsd_amd.SimplePredictor.__init__)
    names = list(sp.parameters)[1:12]
    assert names == ["pth_path", "sr", "channel_order", "input_size", "hidden_size", "num_layers", "num_classes",
                     "dropout", "device", "tailoring_lambda", "class_names"]
    assert sp.parameters["device"].default == "cpu" and sp.parameters["tailoring_lambda"].default == 1.25e-29
    rt = inspect.signature(nsd_amd.run_trials)
    assert [(k, v.default) for k, v in list(rt.parameters.items())[:6]] == [
        ("trials", 10), ("serial_port", nsd_amd.DEFAULT_SERIAL), ("num_channels", 8), ("window_seconds", 5.0),
        ("model_path", nsd_amd.DEFAULT_MODEL), ("verbose", True)]
    assert nsd_amd.CLASS_NAMES == ["Food", "Water", "BG-Noise"]
    r = nsd_amd.TrialResult(trials=0, avg_probs=None)
    assert r.avg_chunk is None

    m = nsd_amd.EEG_LSTM()
    assert list(m.state_dict().keys()) == list(ref_state.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ref_state.items()}, strict=True)
    flat = m.flat_parameters()
    assert np.array_equal(flat.numpy(), orc.flatten_state(ref_state, orc.Dims()))
    with torch.no_grad():
        m.attn.bias.add_(1.0)                                       # parameters are views of the flat vector
    assert flat[ops.ModelSpec().offsets()["attn.bias"]].item() == pytest.approx(float(ref_state["attn.bias"][0]) + 1.0)
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bogus": torch.zeros(1)}, strict=True)


def test_no_cpu_fallback():
    m = nsd_amd.EEG_LSTM()
    with pytest.raises(nsd_amd.NsdError, match="no CPU fallback"):
        m(torch.zeros(2, 10, 8))
    with pytest.raises(nsd_amd.NsdError):
        ops.zscore(torch.zeros(2, 10, 8))
    if not torch.cuda.is_available():
        with pytest.raises(nsd_amd.NsdError, match="MI355X"):
            nsd_amd.SimplePredictor("missing.pth", sr=125, preprocess="identity")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "neural-speech-decoding_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("nsd_oracle.c (nsd_oracle_rand_u32)", ""), fn


def _drain(q, n, timeout=20.0):
    out, t0 = [], time.time()
    while len(out) < n and time.time() - t0 < timeout:
        try:
            out.append(q.get(timeout=0.5))
        except Exception:
            pass
    return out


def test_replay_producer_protocol(tmp_path, golden):
    g = golden("real_trials")
    for i in range(3):
        np.savetxt(tmp_path / f"water_{i}.csv", g["x"][i], fmt="%.7f", delimiter=",")
    q = Queue(maxsize=8)
    p = nsd_amd.StreamingProcess(serial_port=f"replay:{tmp_path}", num_channels=8, window_seconds=5.0, out_queue=q)
    p.start()
    try:
        time.sleep(0.2)
        assert q.empty()                                   # nothing until recording_flag is raised
        p.recording_flag.value = True
        items = _drain(q, 4)
        assert len(items) == 4
        for k, it in enumerate(items):
            assert set(it) == {"sr", "channels", "data", "t_emit"} and it["sr"] == 125
            assert it["data"].dtype == np.float32 and it["data"].shape == (625, 8)
            assert np.allclose(it["data"], g["x"][k % 3], atol=1e-6)   # %.7f round trip, files cycle in order
    finally:
        p.stop(); p.join(timeout=5.0)
        if p.is_alive():
            p.terminate()


def test_synthetic_producer_and_dead_producer():
    q = Queue(maxsize=8)
    p = nsd_amd.StreamingProcess(serial_port="synthetic:3", num_channels=4, window_seconds=1.0, out_queue=q, start_recording=True)
    p.start()
    try:
        it = _drain(q, 1)[0]
        assert it["data"].shape == (125, 4) and np.isfinite(it["data"]).all()
    finally:
        p.stop(); p.join(timeout=5.0)
    with pytest.raises(RuntimeError, match="Producer exited unexpectedly"):
        nsd_amd.run_trials(trials=1, serial_port="/dev/cu.usbserial-FTB6SPL3", model_path="unused", verbose=False,
                           queue_timeout=0.5)


# ---- which preprocessor does the drop-in resolve?  (reference lstm_eeg_model.py:7-10,66,91; Frontend/app.py:22-28) --------
REFERENCE_APP = "/root/reference/Neuro-Alpha-App"
_RESOLVE_SNIPPET = r"""
import sys
sys.dont_write_bytecode = True                      # the reference tree is read-only
sys.path[:0] = [{repo!r}] + {extra!r}
import nsd_amd
from nsd_amd import lstm_eeg_model as M
cls = M.resolve_reference_preprocessor({package!r})
print("RESOLVED", None if cls is None else cls.__module__ + "." + cls.__qualname__)
if cls is not None:
    import numpy as np
    pre = M._default_preprocessor(125, 1.25e-29, {package!r})
    assert type(pre) is cls and pre.sr == 125 and pre.tailoring_lambda == 1.25e-29
    try:
        pre.transform(np.zeros((2, 3, 4), np.float32))
    except ValueError:
        print("VALUEERROR ok")
"""


def _resolve_in_subprocess(extra_path, package=None):
    import subprocess, sys
    code = _RESOLVE_SNIPPET.format(repo=ROOT, extra=list(extra_path), package=package)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


@pytest.mark.skipif(not os.path.isdir(REFERENCE_APP), reason="reference tree not present (GPU box)")
def test_default_preprocessor_is_the_references_in_both_import_modes():
    # package mode: what the Streamlit app does (Neuro-Alpha-App/ on sys.path, `Utilities.*` imports)
    out = _resolve_in_subprocess([REFERENCE_APP])
    assert "RESOLVED Utilities.preprocessor.PreProcessor" in out and "VALUEERROR ok" in out
    # script mode: Utilities/ itself on sys.path (`python tester.py`)
    out = _resolve_in_subprocess([os.path.join(REFERENCE_APP, "Utilities")])
    assert "RESOLVED preprocessor.PreProcessor" in out and "VALUEERROR ok" in out
    # a stub that names its own package gets that package's module first
    out = _resolve_in_subprocess([REFERENCE_APP], package="Utilities")
    assert "RESOLVED Utilities.preprocessor.PreProcessor" in out


def test_missing_reference_preprocessor_is_an_error_not_an_identity():
    out = _resolve_in_subprocess([])
    assert "RESOLVED None" in out
    from nsd_amd import lstm_eeg_model as M
    if M.resolve_reference_preprocessor() is None:
        with pytest.raises(nsd_amd.NsdError, match="identity"):
            M._default_preprocessor(125, 1.25e-29)
    pre = nsd_amd.IdentityPreProcessor(125)
    x = np.arange(12, dtype=np.float64).reshape(4, 3)
    y = pre.transform(x)
    assert y.dtype == np.float32 and np.array_equal(y, x.astype(np.float32))
    with pytest.raises(ValueError):
        pre.transform(np.zeros((2, 3, 4)))


def test_default_model_path_is_checked_before_use(tmp_path, monkeypatch):
    from nsd_amd import tester
    monkeypatch.delenv("NSD_MODEL_PATH", raising=False)
    with pytest.raises(FileNotFoundError, match="does not exist"):
        tester.resolve_model_path(str(tmp_path / "nope.pth"))
    p = tmp_path / "m.pth"
    p.write_bytes(b"x")
    monkeypatch.setenv("NSD_MODEL_PATH", str(p))
    assert tester.resolve_model_path(tester.DEFAULT_MODEL) == str(p)      # the default can be redirected
    assert tester.resolve_model_path(str(p)) == str(p)


def test_bench_accounting_of_the_sequence_batched_configs():
    """The roofline figures bench.py prints are algorithmic FLOP / bytes per launch of the dominant kernel: for cfg3 one launch of
    the fused two-layer scan carries THREE H x 4H products per time step and trial (W_hh0, W_ih1, W_hh1), for cfg5 one launch
    carries one layer's recurrent product in both directions."""
    import bench
    c3, c5 = bench.CONFIGS["cfg3"], bench.CONFIGS["cfg5"]
    assert bench.fused_scans(c3) and bench.scan_kernel_names(c3) == ("scan2_fwd_kernel", "scan2_bwd_kernel")
    assert not bench.fused_scans(c5) and bench.scan_kernel_names(c5) == ("scan_fwd_kernel", "scan_bwd_kernel")
    a3 = bench.algorithmic(c3, c3["B"], c3["T"])
    assert a3["bwd_flop"] == 3 * 2 * 250 * 4 * 256 * 256 * 1024 == 402653184000
    assert a3["bwd_bytes"] == 2 * 250 * 2 * 256 * 2 * 1024          # h and c of both layers, bf16
    a5 = bench.algorithmic(c5, c5["B"], c5["T"])
    assert a5["bwd_flop"] == 2 * 1000 * 2 * 4 * 512 * 512 * 512
    # the kernel-source hash that keys the recorded PMC traffic covers every file of the path
    assert {"nsd_scan.hip", "nsd_scan2.hip", "nsd_scan_common.h", "nsd_gemm_bf16.hip", "nsd_seq.hip"} <= set(bench.KERNEL_SOURCES["bf16"])


def test_bench_knows_where_the_four_trial_kernels_take_over():
    """bench.py names the dominant kernel of the fp32 path by batch size: its threshold must be the launcher's (csrc/nsd_lstm2.hip)."""
    import re
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "neural-speech-decoding_amd", "csrc", "nsd_lstm2.hip")).read()
    m = re.search(r"constexpr int X4_MIN_B = (\d+);", src)
    assert m and int(m.group(1)) == bench.X4_MIN_B
