"""GPU parity tests of the sequence-batched path for large hidden sizes (BASELINE cfg3 / cfg5): bf16 MFMA GEMM building
block, persistent scan kernels, time-major head.  Needs a real MI355X: run with `pytest -m gpu`.

Floating-point path computing in bf16 operands / fp32 accumulation: the tolerances are written at each assert; the oracle
is the fp32 C restatement (oracle/nsd_oracle.c, pinned by the reference's goldens) and, for the GEMM alone, a plain
PyTorch fp32 product of the same bf16-rounded operands.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nsd():
    import nsd_amd
    nsd_amd.load_library()
    return nsd_amd


def _bf(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.bfloat16).contiguous()


# ---------------------------------------------------------------------------------------------------
# GEMM building block
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (1024, 264, 520), (96, 40, 72), (32, 32, 8)])
@pytest.mark.parametrize("a_kmajor,b_kmajor", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_bf16_all_operand_layouts(nsd, dev, M, N, K, a_kmajor, b_kmajor):
    from nsd_amd import ops
    rs = np.random.RandomState(M + 3 * N + 7 * K)
    a = _bf(rs.standard_normal((K, M) if a_kmajor else (M, K)).astype(np.float32), dev)
    b = _bf(rs.standard_normal((K, N) if b_kmajor else (N, K)).astype(np.float32), dev)
    am = a.float().t() if a_kmajor else a.float()                     # [M,K]
    bm = b.float() if b_kmajor else b.float().t()                     # [K,N]
    ref = (am.double() @ bm.double()).float()
    c = ops.gemm_bf16(a, b, a_kmajor=a_kmajor, b_kmajor=b_kmajor)
    torch.cuda.synchronize()
    # fp32 accumulation of exact bf16 products: error ~ 1e-6 * sum|a*b|
    assert (c - ref).abs().max().item() < 2e-5 * K ** 0.5 * 4, (c - ref).abs().max().item()
    # asymmetric operands: a transposed C or a permuted fragment cannot pass
    assert not torch.allclose(c, ref.t()[:M, :N]) if M == N else True
    c16 = ops.gemm_bf16(a, b, a_kmajor=a_kmajor, b_kmajor=b_kmajor, epilogue=1)
    assert (c16.float() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-3


def test_gemm_bf16_split_k_shift_and_tile_epilogue(nsd, dev):
    from nsd_amd import ops
    rs = np.random.RandomState(5)
    # weight-gradient shape: contraction over rows = (t, b), operand shifted by one time step (nb rows)
    T, nb, G, Hh = 12, 32, 256, 128
    R = T * nb
    da = _bf(rs.standard_normal((R, G)).astype(np.float32), dev)
    h = _bf(rs.standard_normal((R, Hh)).astype(np.float32), dev)
    for shift in (-nb, nb, 0):
        hs = torch.zeros_like(h)
        if shift < 0:
            hs[nb:] = h[:-nb]
        elif shift > 0:
            hs[:-nb] = h[nb:]
        else:
            hs = h
        ref = (da.double().t() @ hs.double()).float()
        for splits in (1, 3, 8):
            c = ops.gemm_bf16(da, h, a_kmajor=True, b_kmajor=True, b_shift=shift, splits=splits)
            assert (c - ref).abs().max().item() < 1e-3, (shift, splits, (c - ref).abs().max().item())
    # tile epilogue: 32x32 accumulator tiles + bias[m]; lane l register r <-> row 8*(r/4) + 4*(l>>5) + r%4, column l & 31
    M, N, K = 256, 96, 64
    w = _bf(rs.standard_normal((M, K)).astype(np.float32), dev)
    xin = _bf(rs.standard_normal((N, K)).astype(np.float32), dev)
    bias = torch.from_numpy(rs.standard_normal(M).astype(np.float32)).to(dev)
    tiles = ops.gemm_bf16(w, xin, epilogue=2, bias=bias).float().cpu().numpy()        # [N/32, M/32, 64, 16]
    ref = (w.double() @ xin.double().t()).float() + bias[:, None]
    ref = ref.cpu().numpy()
    lane = np.arange(64)[:, None]
    r = np.arange(16)[None, :]
    rows = 8 * (r // 4) + 4 * (lane >> 5) + (r % 4)
    cols = np.broadcast_to(lane & 31, rows.shape)
    for nt in range(N // 32):
        for mt in range(M // 32):
            want = ref[32 * mt + rows, 32 * nt + cols]
            assert np.abs(tiles[nt, mt] - want).max() <= 2 ** -8 * np.abs(want).max() + 1e-3
