"""Two ranks driving the PRODUCT kernels (not the oracle) through the data-parallel step on one MI355X: the N > 1 launch
sequence  local gradients -> all-reduce of the flat gradient -> Adam  with uneven and EMPTY shards, fp32 path and bf16 path.
Both ranks share cuda:0 (this box has one GPU), so the collective runs over gloo; the arithmetic on either side of it is the
HIP path.  Checked against ONE rank stepping on the full batches with the same streams disabled (stochastic=False)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(precision):
    import nsd_amd
    torch.manual_seed(11)
    if precision == "bf16":
        return nsd_amd.EEG_LSTM(8, 64, 2, 5, dropout=0.6, precision="bf16")
    return nsd_amd.EEG_LSTM(8, 48, 2, 3, dropout=0.6)


def _batches(precision):
    from tests.golden.make_goldens import synth_labels, synth_x
    K = 5 if precision == "bf16" else 3
    return [(synth_x(B, 20, seed=40 + i), synth_labels(B, K, seed=40 + i)) for i, B in enumerate((7, 1, 6, 2))]


def _worker(rank, world, port, precision, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nsd_amd.trainer import Trainer, shard_range
        dev = torch.device("cuda:0")
        m = _make(precision)
        with torch.no_grad():                                   # replicas start different: the broadcast must fix that
            for p in m.parameters():
                p.add_(0.01 * rank)
        tr = Trainer(m.to(dev).train(), lr=1e-3, stochastic=False)
        for x, y in _batches(precision):
            lo, hi = shard_range(len(y), rank, world)
            tr.step(torch.from_numpy(x[lo:hi]).to(dev), torch.from_numpy(y[lo:hi]).to(dev), global_batch=len(y))
        torch.cuda.synchronize()
        assert tr.scan_status() == 0
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), tr.flat.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_ranks_on_product_kernels_equal_one_rank(tmp_path, precision):
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), precision, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(p0, p1)                               # ranks in lock step, bit for bit
    from nsd_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    tr = Trainer(_make(precision).to(dev).train(), lr=1e-3, stochastic=False)
    for x, y in _batches(precision):
        tr.step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    ref = tr.flat.cpu().numpy()
    # Adam moves an element by up to lr per step whatever the gradient's size: elements whose gradient is ~0 can differ by
    # 2*lr*steps between summation orders; everywhere else the trajectories coincide (same bound as tests/test_ddp_gloo_cpu.py)
    dp = np.abs(p0 - ref)
    assert dp.max() <= 2 * 1e-3 * 4 and (dp > (2e-5 if precision == "fp32" else 2e-4)).mean() < 0.02, (dp.max(), (dp > 2e-5).mean())
