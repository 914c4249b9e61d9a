import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nsd_amd
from nsd_amd import ops, _lib
dev = torch.device("cuda:0")
spec = ops.ModelSpec()
w = np.load("/root/repo/tests/golden/weights_3class.npz")
m = nsd_amd.EEG_LSTM(); m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}); m.to(dev)
flat = m.flat_parameters()
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts)//2]
for B, T in [(256, 250), (256, 125), (128, 250), (512, 250), (1024, 250), (1, 625)]:
    x = (2.7 * torch.randn(B, T, 8)).to(dev)
    d = spec.dims(B, T)
    ws = ops.new_workspace(spec, B, T, dev)
    logits = torch.empty(B, 3, device=dev); probs = torch.empty(B, 3, device=dev)
    scratch = torch.empty(B*T*48+16, device=dev)
    t_inf = timed(lambda: L.nsd_infer(C.byref(d), flat.data_ptr(), x.data_ptr(), 0, logits.data_ptr(), probs.data_ptr(), scratch.data_ptr(), st))
    t_fwd = timed(lambda: L.nsd_lstm_fwd(C.byref(d), flat.data_ptr(), x.data_ptr(), None, 2, ws.data_ptr(), ws.numel() * 4, st))
    t_hd = timed(lambda: L.nsd_head_fwd(C.byref(d), flat.data_ptr(), None, None, ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(), st))
    print(f"B={B:5d} T={T:4d}: infer(lstm+head) {t_inf:8.1f} us   train lstm_fwd(no mask) {t_fwd:8.1f} us   head_fwd {t_hd:6.1f} us", flush=True)
