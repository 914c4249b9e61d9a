import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import nsd_amd
from nsd_amd.trainer import Trainer
dev = torch.device("cuda:0")
w = np.load("/root/repo/tests/golden/weights_3class.npz")
rng = np.random.default_rng(7)
bad = 0
for it in range(40):
    B = int(rng.integers(1, 700)); T = int(rng.integers(1, 300)); res = bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(it)
    x = (2.7 * torch.randn(B, T, 8, generator=g)).to(dev); y = torch.randint(0, 3, (B,), generator=g).to(torch.int32).to(dev)
    outs = []
    for ik in (True, False):
        m = nsd_amd.EEG_LSTM(residual=res)
        m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}); m.to(dev).train()
        tr = Trainer(m, lr=1e-3, seed=3); tr.in_kernel_rng = ik
        tr.step(x, y); tr.step(x, y)
        outs.append((tr.grads.clone(), m.flat_parameters().clone(), tr.last_loss()))
    ok = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.isfinite(outs[0][0]).all().item()
    if not ok:
        bad += 1
        print("MISMATCH", B, T, res, (outs[0][0] - outs[1][0]).abs().max().item())
print("done, mismatches:", bad)
