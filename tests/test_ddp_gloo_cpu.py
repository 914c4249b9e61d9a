"""Data-parallel host logic on CPU with the gloo backend, world_size 2 (no GPU needed).

What is exercised is the part of the N>1 path that does not depend on the kernels: contiguous
sharding of the global batch, the 1/B_global scaling convention, ONE all-reduce(SUM) of the flat
gradient vector, and identical Adam updates on every rank.  The per-shard gradients come from the CPU
oracle (test infrastructure) so that "2 ranks x half batch == 1 rank x full batch" can be checked
numerically; on the GPU box the same reducer object sums the gradients the HIP kernels produce.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nsd_amd.trainer import FlatGradAllReducer, shard_range
from oracle import nsd_oracle as orc
from tests.golden.make_goldens import synth_labels, synth_x

D = orc.Dims()


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 8192, 8193):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, golden_dir, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = np.load(os.path.join(golden_dir, "weights_3class.npz"))
        flat = orc.flatten_state({k: w[k] for k in w.files}, D)
        B, T = 10, 24                                   # uneven split is impossible with 2 ranks; 10 -> 5 + 5
        x, y = synth_x(B, T, seed=11), synth_labels(B, seed=11)
        lo, hi = shard_range(B, rank, world)
        # every rank scales its CE gradient by 1/B_global, then ONE all-reduce(SUM) of the flat vector
        _, g_local, _ = orc.loss_and_grads(flat, x[lo:hi], y[lo:hi], D, scale=1.0 / B)
        g = torch.from_numpy(g_local.copy())
        reducer = FlatGradAllReducer()
        assert reducer.world == world
        reducer(g)
        # identical Adam step on every rank
        p, m, v = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
        orc.adam(p, g.numpy(), m, v, lr=1e-3, step=1)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), g=g.numpy(), p=p)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_equals_single_rank_full_batch(tmp_path):
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, golden_dir, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["g"], r1["g"]) and np.array_equal(r0["p"], r1["p"])     # ranks stay in lock step
    w = np.load(os.path.join(golden_dir, "weights_3class.npz"))
    flat = orc.flatten_state({k: w[k] for k in w.files}, D)
    x, y = synth_x(10, 24, seed=11), synth_labels(10, seed=11)
    _, g_full, _ = orc.loss_and_grads(flat, x, y, D)                                   # mean CE over the full batch
    scale = np.abs(g_full).max()
    assert np.abs(r0["g"] - g_full).max() <= 1e-5 * scale                               # 1e-5 relative


def test_reducer_is_identity_without_a_process_group():
    g = torch.arange(8, dtype=torch.float32)
    r = FlatGradAllReducer()
    assert r.world == 1 and torch.equal(r(g.clone()), g)


# ---- the rank-level control flow of a step (trainer.DataParallelStep), world 2 and 3, uneven and EMPTY shards -------------
def _dp_worker(rank, world, port, golden_dir, out_dir, batches):
    from nsd_amd.trainer import DataParallelStep, broadcast_parameters
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = np.load(os.path.join(golden_dir, "weights_3class.npz"))
        flat0 = orc.flatten_state({k: w[k] for k in w.files}, D)
        # replicas start DIFFERENT on purpose: the broadcast from rank 0 must make them identical
        flat = torch.from_numpy((flat0 + (0.01 * rank)).astype(np.float32))
        broadcast_parameters(flat)
        assert np.array_equal(flat.numpy(), flat0)
        grads = torch.zeros_like(flat)
        m, v = np.zeros_like(flat0), np.zeros_like(flat0)
        calls = {"grads": 0, "zero": 0, "update": 0, "step": 0}

        def local_grads(x, y, scale):                      # injected: the oracle stands in for the HIP kernels
            calls["grads"] += 1
            _, g, _ = orc.loss_and_grads(flat.numpy().copy(), x, y, D, scale=scale)
            grads.copy_(torch.from_numpy(g))

        def zero():
            calls["zero"] += 1
            grads.zero_()

        def update():
            calls["update"] += 1
            p = flat.numpy()
            orc.adam(p, grads.numpy(), m, v, lr=1e-3, step=calls["step"])

        step = DataParallelStep(grads, FlatGradAllReducer(), local_grads, zero, update)
        T = 12
        for k, B in enumerate(batches):
            calls["step"] = k + 1
            x, y = synth_x(B, T, seed=20 + k), synth_labels(B, seed=20 + k)
            lo, hi = shard_range(B, rank, world)
            step(x[lo:hi], y[lo:hi], global_batch=B)      # hi == lo on some ranks: must not deadlock
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p=flat.numpy(), g=grads.numpy(),
                 calls=np.array([calls["grads"], calls["zero"], calls["update"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,batches", [(2, (7, 1, 4)), (3, (8, 2, 1, 6))])
def test_data_parallel_step_uneven_and_empty_shards(tmp_path, world, batches):
    """Every rank enters the all-reduce every step (an empty shard contributes zeros), the mean is over the GLOBAL batch,
    replicas are made identical by a broadcast: N ranks == one rank on the full batches, step after step."""
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    mp.spawn(_dp_worker, args=(world, _free_port(), golden_dir, str(tmp_path), batches), nprocs=world, join=True)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r in res[1:]:
        assert np.array_equal(r["p"], res[0]["p"]) and np.array_equal(r["g"], res[0]["g"])
    n_empty = sum(1 for B in batches for r in range(world) if shard_range(B, r, world)[1] == shard_range(B, r, world)[0])
    assert n_empty > 0                                                       # the case the old CLI deadlocked on
    assert sum(int(r["calls"][1]) for r in res) == n_empty
    assert all(int(r["calls"][2]) == len(batches) for r in res)              # every rank applied every update
    # single-rank reference: same batches, mean CE over each full batch, same Adam
    w = np.load(os.path.join(golden_dir, "weights_3class.npz"))
    p = orc.flatten_state({k: w[k] for k in w.files}, D)
    m, v = np.zeros_like(p), np.zeros_like(p)
    for k, B in enumerate(batches):
        x, y = synth_x(B, 12, seed=20 + k), synth_labels(B, seed=20 + k)
        _, g, _ = orc.loss_and_grads(p.copy(), x, y, D)
        orc.adam(p, g, m, v, lr=1e-3, step=k + 1)
    assert np.abs(res[0]["g"] - g).max() <= 1e-5 * np.abs(g).max()
    # Adam divides by sqrt(v): where a gradient element is ~0 its update is +-lr whatever its size, so a summation-order
    # difference can move such an element by up to 2*lr per step; everywhere else the trajectories coincide
    dp = np.abs(res[0]["p"] - p)
    assert dp.max() <= 2 * 1e-3 * len(batches) and (dp > 2e-5).mean() < 0.01


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` OUTSIDE a torchrun environment: the parent (which never touches a GPU) spawns the two ranks with
    the rendezvous environment, they run the launcher / barrier / max-over-ranks / reporting code on a stub step (gloo), and
    exactly one JSON line comes back from rank 0.  The same command line inside a torchrun environment (WORLD_SIZE set) must
    NOT spawn again."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NSD_BENCH_STUB="1", NSD_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--preheat-steps", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["data"] == "stub" and out["metric"].startswith("STUB")
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"] and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["scaling"] == "weak" and out["ranks_seen"] == 2
    # the other BASELINE configs ride in the multi-rank line too: cfg4 = 1024 trials per GPU (global 8192 at 8 ranks), cfg5 = 512 per GPU
    oc = out["other_configs"]
    assert set(oc) == {"cfg3", "cfg4", "cfg5"} and not any("error" in v for v in oc.values()), oc
    assert oc["cfg4"]["config"]["batch_per_gpu"] == 1024 and oc["cfg4"]["config"]["global_batch"] == 1024 * 2
    assert oc["cfg5"]["config"]["batch_per_gpu"] == 512 and oc["cfg5"]["config"]["global_batch"] == 512 * 2
    assert all(v["n_gpus"] == 2 and v["ranks_seen"] == 2 for v in oc.values())
    # a mismatching launcher environment is refused (rc 2), not re-launched
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                        env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode == 2 and "WORLD_SIZE=1" in r2.stderr


@pytest.mark.timeout(300)
def test_bench_keeps_the_headline_when_another_config_fails():
    """The default run measures cfg3 / cfg4 / cfg5 after the headline config; whatever goes wrong in one of them (here: the stub's
    step raises MemoryError for cfg3) is recorded in its place, the remaining configs are still measured and the cfg2 line is printed
    with rc 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NSD_BENCH_STUB="1", NSD_DIST_BACKEND="gloo", NSD_BENCH_STUB_FAIL="cfg3")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--preheat-steps", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["config"]["name"] == "cfg2" and out["value"] > 0
    oc = out["other_configs"]
    assert "MemoryError" in oc["cfg3"]["error"] and oc["cfg4"]["value"] > 0 and oc["cfg5"]["value"] > 0
