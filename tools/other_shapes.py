import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import nsd_amd
from nsd_amd.trainer import Trainer
dev = torch.device("cuda:0")
for (C,H,L,K,B,T) in [(8,256,2,5,256,250),(8,256,2,5,1024,250),(8,64,2,3,256,250),(8,32,2,3,256,250)]:
    m = nsd_amd.EEG_LSTM(C,H,L,K, dropout=0.6).to(dev).train()
    tr = Trainer(m, lr=1e-3, seed=1)
    g = torch.Generator().manual_seed(0)
    x = (2.7*torch.randn(B,T,C,generator=g)).to(dev); y = torch.randint(0,K,(B,),generator=g).to(torch.int32).to(dev)
    for _ in range(3): tr.step(x,y)
    torch.cuda.synchronize(); t0=time.perf_counter()
    n=5
    for _ in range(n): tr.step(x,y)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
    print(f"C={C} H={H} L={L} K={K} B={B} T={T}: {dt*1e3:9.2f} ms/step  {B/dt:10.0f} trials/s", flush=True)
