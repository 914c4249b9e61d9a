import sys, os, torch, numpy as np, ctypes as C
sys.path.insert(0, "/root/repo")
import nsd_amd
from nsd_amd import ops, _lib
dev = torch.device("cuda:0")
spec = ops.ModelSpec(); B, T = 256, 250
L = _lib.lib(); d = spec.dims(B, T)
ws = ops.new_workspace(spec, B, T, dev); ws.normal_()
g = torch.zeros(spec.param_count, device=dev); p = torch.zeros_like(g); m = torch.zeros_like(g); v = torch.zeros_like(g)
st = torch.cuda.current_stream().cuda_stream
def run():
    L.nsd_grad_reduce_adam(C.byref(d), ws.data_ptr(), ws.numel() * 4, g.data_ptr(), p.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 1, st)
for _ in range(200): run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); run(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(f"{os.environ.get('NSD_LIB','libnsd_hip.so'):22s} reduce+adam {ts[len(ts)//2]:6.2f} us (min {ts[0]:.2f})")
