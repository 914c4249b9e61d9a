// nsd_gemm_bf16.hip -- bf16 MFMA GEMM of the sequence-batched path (large hidden sizes; BASELINE cfg3 / cfg5).
//
// Everything in that path that is NOT the serial recurrence is a contraction over the whole sequence and runs here:
//   input projection of a layer     xproj[4H, T*B]  = W_ih'[4H, I] . in[T*B, I]^T           (both operands k-contiguous)
//   gradient w.r.t. a layer's input  d_in[T*B, I]    = da[T*B, 4H] . W_ih^T'[I, 4H]^T         (both operands k-contiguous)
//   weight gradients                 dW[4H, I | H]   = da[T*B, 4H]^T . {in | h_prev}[T*B, .]  (both operands k-major: the
//                                                       contraction runs over the rows -> transposed LDS reads, split-K)
// (torch.nn.LSTM of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34 and autograd through it, for the layers whose
// input is a full sequence).  Two kernels: 128x128 tile, K chunks of 64, 4 waves as 2x2, each wave 2x2 v_mfma_f32_32x32x16_bf16
// tiles, operands staged global -> registers -> LDS with the next chunk's loads in flight during the MFMAs; and, where the
// problem has at least one 256x256 tile per CU, the double-buffered 256x256 kernel further down (namespace big).
// LDS images: k-contiguous operand [128 rows][64 k + 8 pad] (ds_read_b128 fragments, conflict-free), k-major operand
// [64 k][128 + 32 pad] (ds_read_b64_tr_b16 fragments: row stride = 16 dwords mod 64 keeps the 4 k-rows x 2 column blocks of
// a 32-lane half on disjoint banks).
#include "nsd_bf16.h"

namespace {

constexpr int GM = 128, GN = 128, GK = 64;
constexpr int LDR = GK + 8;        // halves per LDS row, k-contiguous image
constexpr int LDT = 128 + 32;      // halves per LDS row, k-major image
constexpr int OPBYTES = (GM * LDR > GK * LDT ? GM * LDR : GK * LDT) * 2;

struct Pieces { u32x4 v[4]; };

// k-contiguous operand P[rows][ld]: tile rows r0.., k chunk k0..
__device__ __forceinline__ Pieces load_rowmajor(const bf16_t *P, const long ld, const int r0, const int nrows, const long k0, const long k_hi,
                                                const int tid) {
    Pieces p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, row = e >> 3, kc = e & 7;
        const long k = k0 + 8 * kc;
        const bool ok = r0 + row < nrows && k < k_hi;
        p.v[i] = ok ? *reinterpret_cast<const u32x4 *>(P + (long)(r0 + row) * ld + k) : u32x4{0u, 0u, 0u, 0u};
    }
    return p;
}
__device__ __forceinline__ void store_rowmajor(bf16_t *S, const Pieces &p, const int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, row = e >> 3, kc = e & 7;
        *reinterpret_cast<u32x4 *>(S + row * LDR + 8 * kc) = p.v[i];
    }
}
// k-major operand P[K][ld]: chunk rows k0.. (shifted by `shift`, rows outside [0, K) are zero), columns c0..
__device__ __forceinline__ Pieces load_kmajor(const bf16_t *P, const long ld, const int c0, const int ncols, const long k0, const long k_hi,
                                              const long shift, const long period, const long K, const int tid) {
    Pieces p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, kr = e >> 4, cc = e & 15;
        const long k = k0 + kr, ks = k + shift;
        bool ok = k < k_hi && ks >= 0 && ks < K && c0 + 8 * cc < ncols;
        if (period > 0 && shift != 0) {                         // (K < 2^31 on this path: checked at launch)
            const long kin = (long)((unsigned)k % (unsigned)period) + shift;
            ok = ok && kin >= 0 && kin < period;
        }
        p.v[i] = ok ? *reinterpret_cast<const u32x4 *>(P + ks * ld + c0 + 8 * cc) : u32x4{0u, 0u, 0u, 0u};
    }
    return p;
}
__device__ __forceinline__ void store_kmajor(bf16_t *S, const Pieces &p, const int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, kr = e >> 4, cc = e & 15;
        *reinterpret_cast<u32x4 *>(S + kr * LDT + 8 * cc) = p.v[i];
    }
}
// fragment of the 32 rows (or columns) starting at `base` of the tile, k step ks (16 k)
__device__ __forceinline__ bf16x8 frag_rowmajor(const bf16_t *S, const int base, const int ks, const int lane) {
    return *reinterpret_cast<const bf16x8 *>(S + (base + (lane & 31)) * LDR + 16 * ks + 8 * (lane >> 5));
}
__device__ __forceinline__ bf16x8 frag_kmajor(const bf16_t *S, const int base, const int ks, const int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const bf16_t *a = S + (16 * ks + 8 * (g >> 1) + q) * LDT + base + 16 * (g & 1) + 4 * p;
    return cat_tr(lds_read_tr16(a), lds_read_tr16(a + 4 * LDT));
}

// epilogue shared by the kernels: the wave's MI x NJ accumulator tiles starting at (mb0, nb0)
template <int EPI, int MI, int NJ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, const f32x16 (&acc)[MI][NJ], const int mb0, const int nb0, const int lane) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int mb = mb0 + 32 * i, nb = nb0 + 32 * j;
            const int n = nb + (lane & 31);
            if (EPI == GEMM_EPI_TILE_BF16 || EPI == GEMM_EPI_TILE_WAVE_BF16) {
                if (mb + 32 <= g.M && nb + 32 <= g.N) {            // whole tiles only (M, N multiples of 32 on these routes)
                    unsigned w[8];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float b0 = g.bias ? g.bias[mb + mfma32_row(r, lane)] : 0.f;
                        const float b1 = g.bias ? g.bias[mb + mfma32_row(r + 1, lane)] : 0.f;
                        w[r >> 1] = pack_bf16x2(acc[i][j][r] + b0, acc[i][j][r + 1] + b1);
                    }
                    bf16_t *tile = reinterpret_cast<bf16_t *>(g.C) + ((long)(nb >> 5) * (g.M >> 5) + (mb >> 5)) * 1024;
                    if (EPI == GEMM_EPI_TILE_BF16) {
                        *reinterpret_cast<u32x4 *>(tile + lane * 16) = u32x4{w[0], w[1], w[2], w[3]};
                        *reinterpret_cast<u32x4 *>(tile + lane * 16 + 8) = u32x4{w[4], w[5], w[6], w[7]};
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x2 *>(tile + q * 256 + lane * 4) = u32x2{w[2 * q], w[2 * q + 1]};
                    }
                }
            } else {
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + mfma32_row(r, lane);
                    if (m >= g.M) continue;
                    if (EPI == GEMM_EPI_F32)
                        (reinterpret_cast<float *>(g.C) + (long)blockIdx.z * g.M * g.ldc)[(long)m * g.ldc + n] =
                            acc[i][j][r] + (g.add ? g.add[(long)m * g.ldc + n] : 0.f);
                    else
                        reinterpret_cast<bf16_t *>(g.C)[(long)m * g.ldc + n] = (bf16_t)acc[i][j][r];
                }
            }
        }
}

template <bool AK, bool BK, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs g) {
    __shared__ __align__(16) unsigned char smem[2 * OPBYTES];
    bf16_t *As = reinterpret_cast<bf16_t *>(smem), *Bs = reinterpret_cast<bf16_t *>(smem + OPBYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    // split-K range (whole chunks)
    const long nchunks = (g.K + GK - 1) / GK;
    const long per = (nchunks + g.splits - 1) / g.splits;
    const long c_lo = (long)blockIdx.z * per, c_hi = (c_lo + per < nchunks) ? c_lo + per : nchunks;
    const long k_lo = c_lo * GK, k_hi = (c_hi * GK < g.K) ? c_hi * GK : g.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    auto lda_ = [&](long k0) { return AK ? load_kmajor(g.A, g.lda, m0, g.M, k0, k_hi, 0, 0, g.K, tid) : load_rowmajor(g.A, g.lda, m0, g.M, k0, k_hi, tid); };
    // (columns from the second source where the tile lies beyond n_split)
    const bool second = BK && g.B2 != nullptr && n0 >= g.n_split;
    const bf16_t *Bsrc = second ? g.B2 : g.B;
    const long ldbs = second ? g.ldb2 : g.ldb, bsh = second ? g.b2_shift : g.b_shift;
    const int nloc = second ? n0 - g.n_split : n0, ncol = g.B2 == nullptr ? g.N : (second ? g.N - g.n_split : g.n_split);
    auto ldb_ = [&](long k0) { return BK ? load_kmajor(Bsrc, ldbs, nloc, ncol, k0, k_hi, bsh, g.b_period, g.K, tid) : load_rowmajor(g.B, g.ldb, n0, g.N, k0, k_hi, tid); };
    if (k_lo < k_hi) {
        Pieces pa = lda_(k_lo), pb = ldb_(k_lo);
        for (long k0 = k_lo; k0 < k_hi; k0 += GK) {
            __syncthreads();
            if (AK) store_kmajor(As, pa, tid); else store_rowmajor(As, pa, tid);
            if (BK) store_kmajor(Bs, pb, tid); else store_rowmajor(Bs, pb, tid);
            if (k0 + GK < k_hi) { pa = lda_(k0 + GK); pb = ldb_(k0 + GK); }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < GK / 16; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[i] = AK ? frag_kmajor(As, 64 * wm + 32 * i, ks, lane) : frag_rowmajor(As, 64 * wm + 32 * i, ks, lane);
                    b[i] = BK ? frag_kmajor(Bs, 64 * wn + 32 * i, ks, lane) : frag_rowmajor(Bs, 64 * wn + 32 * i, ks, lane);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    gemm_epilogue<EPI, 2, 2>(g, acc, m0 + 64 * wm, n0 + 64 * wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA tiles), LDS double-buffered, ONE barrier per K chunk.
// Why a second kernel: the 128 x 128 tile moves 32 KB from the L2 per 128 x 128 x 64 MACs = 64 FLOP per byte, and a CU takes
// 64 bytes per clock from the L2 -- 4 096 FLOP per clock, exactly the CU's bf16 MFMA peak: the small tile sits ON the L2
// bandwidth roof (measured 0.5-0.9 PFLOP/s).  The 256 x 256 tile needs half the bytes per FLOP.  One workgroup per CU (8 waves,
// 144 KB of LDS), so the overlap of loads and MFMAs comes from the double buffer, not from co-resident workgroups.
// ---------------------------------------------------------------------------------------------------------------------------
namespace big {
constexpr int TM = 256, TN = 256, NTH = 512;
constexpr int LDTB = 256 + 32;                                  // halves per LDS row, k-major image (16 dwords mod 64, as LDT)
constexpr int OPB = (TM * LDR > GK * LDTB ? TM * LDR : GK * LDTB) * 2;     // bytes of one operand image

__device__ __forceinline__ Pieces load_rowmajor_b(const bf16_t *P, const long ld, const int r0, const int nrows, const long k0, const long k_hi,
                                                const int tid) {
    Pieces p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + NTH * i, row = e >> 3, kc = e & 7;
        const long k = k0 + 8 * kc;
        const bool ok = r0 + row < nrows && k < k_hi;
        p.v[i] = ok ? *reinterpret_cast<const u32x4 *>(P + (long)(r0 + row) * ld + k) : u32x4{0u, 0u, 0u, 0u};
    }
    return p;
}
__device__ __forceinline__ void store_rowmajor_b(bf16_t *S, const Pieces &p, const int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + NTH * i, row = e >> 3, kc = e & 7;
        *reinterpret_cast<u32x4 *>(S + row * LDR + 8 * kc) = p.v[i];
    }
}
__device__ __forceinline__ Pieces load_kmajor_b(const bf16_t *P, const long ld, const int c0, const int ncols, const long k0, const long k_hi,
                                              const long shift, const long period, const long K, const int tid) {
    Pieces p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + NTH * i, kr = e >> 5, cc = e & 31;
        const long k = k0 + kr, ks = k + shift;
        bool ok = k < k_hi && ks >= 0 && ks < K && c0 + 8 * cc < ncols;
        if (period > 0 && shift != 0) {
            const long kin = (long)((unsigned)k % (unsigned)period) + shift;
            ok = ok && kin >= 0 && kin < period;
        }
        p.v[i] = ok ? *reinterpret_cast<const u32x4 *>(P + ks * ld + c0 + 8 * cc) : u32x4{0u, 0u, 0u, 0u};
    }
    return p;
}
__device__ __forceinline__ void store_kmajor_b(bf16_t *S, const Pieces &p, const int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + NTH * i, kr = e >> 5, cc = e & 31;
        *reinterpret_cast<u32x4 *>(S + kr * LDTB + 8 * cc) = p.v[i];
    }
}
__device__ __forceinline__ bf16x8 frag_kmajor_b(const bf16_t *S, const int base, const int ks, const int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const bf16_t *a = S + (16 * ks + 8 * (g >> 1) + q) * LDTB + base + 16 * (g & 1) + 4 * p;
    return cat_tr(lds_read_tr16(a), lds_read_tr16(a + 4 * LDTB));
}

template <bool AK, bool BK, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_big_kernel(const GemmArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];       // [2 buffers][A image | B image]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;                    // wave tile: rows 128 wm .., columns 64 wn ..
    // XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2: with the natural (x, y) order the
    // workgroups that share an operand tile land on different L2s and the tile is fetched from memory once per XCD (the
    // layer-1 projection at cfg5 re-read its 1-GB activation matrix 8 times).  Here XCD k = id % 8 takes a CONTIGUOUS range of the
    // tile sequence, and the sequence walks groups of up to 8 M-tiles (M fastest) before it moves along N: the ~32 tiles an
    // XCD works on at a time are an 8 x 4 block sharing 8 A tiles and 4 B tiles.
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN, nt = tiles_m * tiles_n, tpx = (nt + 7) / 8;
    // (only where there are tiles to share: with a handful of tiles per split-K slice the ids keep their natural order -- the
    // padded, remapped grid of fewer than 8 tiles would park every real tile of every slice on the same few XCDs)
    const bool remap = gridDim.x == (unsigned)(8 * tpx) && nt >= 16;
    const int q = remap ? (int)(blockIdx.x & 7) * tpx + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (q >= nt) return;                                        // (a remapped grid is padded to 8 * tpx; whole workgroups leave)
    const int GH = tiles_m < 8 ? tiles_m : 8, per_group = GH * tiles_n, grp = q / per_group, rr = q - grp * per_group;
    const int gh = (grp + 1) * GH <= tiles_m ? GH : tiles_m - grp * GH;          // (last group of a ragged M)
    const int tile_m = grp * GH + rr % gh, tile_n = rr / gh;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const long nchunks = (g.K + GK - 1) / GK;
    const long per = (nchunks + g.splits - 1) / g.splits;
    const long c_lo = (long)blockIdx.z * per, c_hi = (c_lo + per < nchunks) ? c_lo + per : nchunks;
    const long k_lo = c_lo * GK, k_hi = (c_hi * GK < g.K) ? c_hi * GK : g.K;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    auto lda_ = [&](long k0) { return AK ? load_kmajor_b(g.A, g.lda, m0, g.M, k0, k_hi, 0, 0, g.K, tid) : load_rowmajor_b(g.A, g.lda, m0, g.M, k0, k_hi, tid); };
    const bool second = BK && g.B2 != nullptr && n0 >= g.n_split;
    const bf16_t *Bsrc = second ? g.B2 : g.B;
    const long ldbs = second ? g.ldb2 : g.ldb, bsh = second ? g.b2_shift : g.b_shift;
    const int nloc = second ? n0 - g.n_split : n0, ncol = g.B2 == nullptr ? g.N : (second ? g.N - g.n_split : g.n_split);
    auto ldb_ = [&](long k0) { return BK ? load_kmajor_b(Bsrc, ldbs, nloc, ncol, k0, k_hi, bsh, g.b_period, g.K, tid) : load_rowmajor_b(g.B, g.ldb, n0, g.N, k0, k_hi, tid); };
    auto put = [&](const int buf, const Pieces &pa, const Pieces &pb) {
        bf16_t *As = reinterpret_cast<bf16_t *>(smem + (size_t)buf * 2 * OPB), *Bs = reinterpret_cast<bf16_t *>(smem + (size_t)buf * 2 * OPB + OPB);
        if (AK) store_kmajor_b(As, pa, tid); else store_rowmajor_b(As, pa, tid);
        if (BK) store_kmajor_b(Bs, pb, tid); else store_rowmajor_b(Bs, pb, tid);
    };
    if (k_lo < k_hi) {
        Pieces pa = lda_(k_lo), pb = ldb_(k_lo);
        put(0, pa, pb);
        __syncthreads();
        int buf = 0;
        for (long k0 = k_lo; k0 < k_hi; k0 += GK, buf ^= 1) {
            const bool more = k0 + GK < k_hi;
            if (more) { pa = lda_(k0 + GK); pb = ldb_(k0 + GK); }              // in flight during the MFMAs of this chunk
            const bf16_t *As = reinterpret_cast<const bf16_t *>(smem + (size_t)buf * 2 * OPB), *Bs = reinterpret_cast<const bf16_t *>(smem + (size_t)buf * 2 * OPB + OPB);
            // fragments of k-step ks+1 are requested before the MFMAs of k-step ks are issued (two register sets): the LDS latency
            // of a k-step hides behind the 8 MFMAs of the one before instead of being paid in front of each MFMA pair
            bf16x8 fa[2][4], fb[2][2];
            auto frags = [&](const int ks, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = AK ? frag_kmajor_b(As, 128 * wm + 32 * i, ks, lane) : frag_rowmajor(As, 128 * wm + 32 * i, ks, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = BK ? frag_kmajor_b(Bs, 64 * wn + 32 * j, ks, lane) : frag_rowmajor(Bs, 64 * wn + 32 * j, ks, lane);
            };
            // (with both operands k-major a fragment is two transposing reads and the second register set spills: one set there)
            constexpr bool TWO_SETS = !(AK && BK);
            frags(0, fa[0], fb[0]);
#pragma unroll
            for (int ks = 0; ks < GK / 16; ++ks) {
                if (TWO_SETS) {
                    if (ks + 1 < GK / 16) frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (ks > 0) {
                    frags(ks, fa[0], fb[0]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TWO_SETS ? (ks & 1) : 0][i], fb[TWO_SETS ? (ks & 1) : 0][j], acc[i][j], 0, 0, 0);
                if (TWO_SETS) __builtin_amdgcn_sched_barrier(0);
                // the next chunk goes to the OTHER buffer (its last readers passed the barrier of the previous chunk) in the middle of
                // this chunk's MFMAs: the wait for the loads and the ds_writes then overlap the partner wave's MFMAs instead of
                // both waves of a SIMD meeting at the barrier with their stores still to do
                if (ks == 1 && more) put(buf ^ 1, pa, pb);
            }
            __syncthreads();
        }
    }
    gemm_epilogue<EPI, 4, 2>(g, acc, m0 + 128 * wm, n0 + 64 * wn, lane);
}

template <bool AK, bool BK, int EPI>
int launch_one(const GemmArgs &g, const dim3 grid, hipStream_t st) {
    static bool once = false;                                   // 144 KB of dynamic LDS needs the opt-in (per kernel, once per process)
    auto *kp = &gemm_bf16_big_kernel<AK, BK, EPI>;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * OPB) != hipSuccess) {
            nsd_set_error("gemm_bf16: cannot reserve %d bytes of LDS", 4 * OPB);
            return NSD_E_INVALID;
        }
        once = true;
    }
    hipLaunchKernelGGL(kp, grid, dim3(NTH), 4 * OPB, st, g);
    NSD_CHECK_LAUNCH("gemm_bf16_big_kernel");
    return NSD_OK;
}
template <bool AK, bool BK>
int launch_epi(const GemmArgs &g, const dim3 grid, hipStream_t st) {
    switch (g.epi) {
    case GEMM_EPI_F32: return launch_one<AK, BK, GEMM_EPI_F32>(g, grid, st);
    case GEMM_EPI_BF16: return launch_one<AK, BK, GEMM_EPI_BF16>(g, grid, st);
    case GEMM_EPI_TILE_BF16: return launch_one<AK, BK, GEMM_EPI_TILE_BF16>(g, grid, st);
    case GEMM_EPI_TILE_WAVE_BF16: return launch_one<AK, BK, GEMM_EPI_TILE_WAVE_BF16>(g, grid, st);
    default: nsd_set_error("gemm_bf16: unknown epilogue %d", g.epi); return NSD_E_INVALID;
    }
}
}  // namespace big

// ---------------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile for TWO K-CONTIGUOUS operands (input projection, input gradient: the two largest GEMMs of cfg5), staged by LDS-DMA.
// What the counters said about the register-staged kernel above on these shapes (37 % MFMA busy, 39 % of the wave cycles in
// s_waitcnt / barrier): a chunk's global loads are requested at the top of the chunk BEFORE it and written to the LDS a quarter
// of a chunk later -- ~500 cycles of flight time against an L2 round trip of 500-800 -- and the staging registers leave no room
// for a second set.  Here the next chunk's bytes go global -> LDS directly (buffer_load_dwordx4 ... lds: no staging registers,
// no ds_write pass), requested right after the barrier that frees their buffer, i.e. a whole chunk (~2 000 cycles) ahead.
//   LDS image of an operand chunk: [256 rows][64 k] bf16, 128-byte rows, UNPADDED (a wave's DMA instruction fills 64 lanes x 16 B =
//   8 whole rows, lane-linear); the 16-byte piece c of row r sits at piece c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128
//   group (16 consecutive rows, one k piece) then cover all 16 slots of the 256-byte bank row -- conflict-free.  The permutation
//   is applied on the SOURCE address of each lane (the destination of an LDS-DMA is fixed) and on the fragment reads.
//   Rows beyond M / N and k beyond the split's range are sent out of the buffer descriptor's range: the DMA writes zeros.
// ---------------------------------------------------------------------------------------------------------------------------
// timing experiments only (-DNSD_GEMM_ABL=n, never shipped; results wrong): 1 no DMA inside the loop, 2 no MFMAs, 4 no fragment reads
#ifndef NSD_GEMM_ABL
#define NSD_GEMM_ABL 0
#endif
#if NSD_GEMM_ABL & 8
__device__ unsigned long long g_gemm_stamps[8];
extern "C" int nsd_debug_gemm_stamps(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(g_gemm_stamps)) == hipSuccess ? 0 : -2; }
#define GSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
namespace dma {
constexpr int TM = 256, TN = 256, NTH = 512, ROWB = GK * 2;     // bytes per LDS row
constexpr int OPB = TM * ROWB;                                   // bytes of one operand image (32 KB)
constexpr unsigned OOB = 0x7fffff00u;                            // a byte offset beyond num_records of the descriptors below

// SA / SB: LDS images (pipeline stages) of operand A / B.  Two stages = the next chunk is requested a chunk ahead: enough for an
// operand that sits in the L2 (the weights: 4-8 MB, shared by every workgroup of the XCD).  The OTHER operand of these GEMMs is a
// 1-4-GB activation matrix streamed from HBM exactly once, and every workgroup of an XCD that shares a chunk of it asks for it at
// the same moment: one chunk (~2 000 cycles) of flight time against an HBM latency of ~4 500 under load left every chunk waiting.
// It gets THREE stages -- requested two chunks ahead, its DMA stays in flight across the chunk barrier (counted vmcnt, raw
// s_barrier) -- and the LDS is full: (2 + 3) x 32 KB = 160 KB.
template <int EPI, int SA, int SB>
__global__ __launch_bounds__(512) void gemm_bf16_dma_kernel(const GemmArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];       // [SA A images | SB B images], the ONLY LDS object of the kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;                    // wave tile: rows 128 wm .., columns 64 wn ..
    // XCD-aware tile order: as gemm_bf16_big_kernel
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN, nt = tiles_m * tiles_n, tpx = (nt + 7) / 8;
    const bool remap = gridDim.x == (unsigned)(8 * tpx) && nt >= 16;
    const int q = remap ? (int)(blockIdx.x & 7) * tpx + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (q >= nt) return;
    const int GH = tiles_m < 8 ? tiles_m : 8, per_group = GH * tiles_n, grp = q / per_group, rr = q - grp * per_group;
    const int gh = (grp + 1) * GH <= tiles_m ? GH : tiles_m - grp * GH;
    const int tile_m = grp * GH + rr % gh, tile_n = rr / gh;
    const int m0 = tile_m * TM, n0 = tile_n * TN;
    const long nchunks = (g.K + GK - 1) / GK;
    const long per = (nchunks + g.splits - 1) / g.splits;
    const long c_lo = (long)blockIdx.z * per, c_hi = (c_lo + per < nchunks) ? c_lo + per : nchunks;
    const long k_lo = c_lo * GK, k_hi = (c_hi * GK < g.K) ? c_hi * GK : g.K;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    // descriptors based at the tile's first row: lane offsets stay far below 2^31 (checked at launch)
    const nsd_rsrc ra = make_rsrc(g.A + (long)m0 * g.lda, 0x7ffffe00u), rb = make_rsrc(g.B + (long)n0 * g.ldb, 0x7ffffe00u);
    // this lane's piece of instruction i: row 64 i + 8 wave + (lane >> 3), LDS piece lane & 7 = source piece (lane & 7) ^ ((row >> 1) & 7)
    unsigned va[4], vb[4];
    int kpiece;                                                 // source k piece of this lane (the same for all 4 rows: 64 i + 8 wave keeps (row >> 1) & 7)
    {
        const int rl = lane >> 3, sw = (((8 * wave + rl) >> 1) & 7);
        kpiece = (lane & 7) ^ sw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 64 * i + 8 * wave + rl;
            va[i] = m0 + r < g.M ? (unsigned)(r * g.lda * 2 + kpiece * 16) : OOB;
            vb[i] = n0 + r < g.N ? (unsigned)(r * g.ldb * 2 + kpiece * 16) : OOB;
        }
    }
    typedef __attribute__((address_space(3))) void *lds_ptr;
    unsigned char *const Abase = smem, *const Bbase = smem + (size_t)SA * OPB;
    const long ntile = k_lo < k_hi ? (k_hi - k_lo + GK - 1) / GK : 0;
    // request chunk t of one operand into its image t % S (nothing is issued beyond the split's range: the counted waits below
    // assume exactly 4 DMA instructions per operand and chunk, so a chunk that does not exist is "requested" out of range: zeros)
    // DMA instruction i (0..3) of chunk t of an operand: rows 64 i + 8 wave .. of the image t % S
    auto dma_a = [&](const long t, const int i) {
        unsigned char *dst = Abase + (size_t)(t % SA) * OPB + (size_t)wave * 8 * ROWB;
        const long k0 = k_lo + t * GK;
        const bool kok = t < ntile && k0 + 8 * kpiece < k_hi;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(dst + i * 64 * ROWB), 16, kok ? va[i] : OOB, t < ntile ? (int)(k0 * 2) : 0, 0, 0);
    };
    auto dma_b = [&](const long t, const int i) {
        unsigned char *dst = Bbase + (size_t)(t % SB) * OPB + (size_t)wave * 8 * ROWB;
        const long k0 = k_lo + t * GK;
        const bool kok = t < ntile && k0 + 8 * kpiece < k_hi;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(dst + i * 64 * ROWB), 16, kok ? vb[i] : OOB, t < ntile ? (int)(k0 * 2) : 0, 0, 0);
    };
    auto stage_a = [&](const long t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(t, i);
    };
    auto stage_b = [&](const long t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b(t, i);
    };
    // fragment addresses: row base + (lane & 31), k piece (2 ks + (lane >> 5)) ^ (((lane & 31) >> 1) & 7)   (row bases are multiples of 32)
    const int fsw = ((lane & 31) >> 1) & 7, fkq = lane >> 5;
    int koff[GK / 16];
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) koff[ks] = (lane & 31) * ROWB + 16 * ((2 * ks + fkq) ^ fsw);

    if (ntile > 0) {
        // prologue: chunk 0 of both, then what each operand keeps in flight ahead (S - 1 chunks)
        stage_a(0); stage_b(0);
        if (SA == 3) stage_a(1);
        if (SB == 3) stage_b(1);
        // issue order inside chunk t of the loop: [2-stage operand: chunk t+1] [3-stage operand: chunk t+2]; what must have landed
        // at the end of chunk t is chunk t+1 of both: everything but the 4 youngest instructions where a 3-stage operand exists
        if (SA == 3 || SB == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#if NSD_GEMM_ABL & 8
        unsigned long long st_acc[4] = {0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
        for (long t = 0; t < ntile; ++t) {
            // (the barrier before this point ended every wave's reads of chunk t-1: its images are free)
            GSTAMP(0);
            const unsigned char *As = Abase + (size_t)(t % SA) * OPB + (size_t)(128 * wm) * ROWB, *Bs = Bbase + (size_t)(t % SB) * OPB + (size_t)(64 * wn) * ROWB;
            bf16x8 fa[2][4], fb[2][2];
            auto frags = [&](const int ks, bf16x8 (&a)[4], bf16x8 (&b)[2]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(As + i * 32 * ROWB + koff[ks]);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(Bs + j * 32 * ROWB + koff[ks]);
            };
            // The chunk's 8 DMA instructions are spread over its 32 MFMAs, one behind every fourth: issued in one burst at the top
            // of the chunk they took a quarter of the chunk's time (the CU's vector-memory path moves 64 B per clock: 64 KB per
            // chunk) during which both waves of every SIMD stood in the issue queue and no MFMA ran.  Order: first the operand with
            // two stages (its chunk t+1 must land by the end of THIS chunk), then the one with three (chunk t+2: a chunk of slack).
            auto dma_step = [&](const int n) {                   // n = 0..7
                if (NSD_GEMM_ABL & 1) return;
                constexpr bool A_FIRST = SA == 2;
                const bool first_half = n < 4;
                const int i = n & 3;
                if (first_half == A_FIRST) dma_a(t + (SA - 1), i); else dma_b(t + (SB - 1), i);
            };
            if (!(NSD_GEMM_ABL & 4) || t == 0) frags(0, fa[0], fb[0]);
#pragma unroll
            for (int ks = 0; ks < GK / 16; ++ks) {
                if (ks + 1 < GK / 16 && (!(NSD_GEMM_ABL & 4) || t == 0)) frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (!(NSD_GEMM_ABL & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
                        else acc[i][j][0] += (float)fa[ks & 1][i][0] * (float)fb[ks & 1][j][0];
                    if (i & 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_step(2 * ks + (i >> 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            GSTAMP(1);                                          // fragment reads + MFMAs
            if (SA == 3 || SB == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GSTAMP(2);                                          // wait for the DMA
            __builtin_amdgcn_s_barrier();
            GSTAMP(3);                                          // barrier
        }
#if NSD_GEMM_ABL & 8
        if (blockIdx.x == 8 && blockIdx.z == 0 && tid == 0) { for (int i = 0; i < 4; ++i) g_gemm_stamps[i] = st_acc[i]; g_gemm_stamps[4] = (unsigned long long)ntile; }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (out-of-range requests of the last chunks: nothing may be in flight when the LDS is released)
    }
    gemm_epilogue<EPI, 4, 2>(g, acc, m0 + 128 * wm, n0 + 64 * wn, lane);
}

template <int EPI, int SA, int SB>
int launch_one(const GemmArgs &g, const dim3 grid, hipStream_t st) {
    static bool once = false;                                   // up to 160 KB of dynamic LDS needs the opt-in (per kernel, once per process)
    auto *kp = &gemm_bf16_dma_kernel<EPI, SA, SB>;
    constexpr int LDS = (SA + SB) * OPB;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            nsd_set_error("gemm_bf16: cannot reserve %d bytes of LDS", LDS);
            return NSD_E_INVALID;
        }
        once = true;
    }
    hipLaunchKernelGGL(kp, grid, dim3(NTH), LDS, st, g);
    NSD_CHECK_LAUNCH("gemm_bf16_dma_kernel");
    return NSD_OK;
}
template <int SA, int SB>
int launch_stages(const GemmArgs &g, const dim3 grid, hipStream_t st) {
    switch (g.epi) {
    case GEMM_EPI_F32: return launch_one<GEMM_EPI_F32, SA, SB>(g, grid, st);
    case GEMM_EPI_BF16: return launch_one<GEMM_EPI_BF16, SA, SB>(g, grid, st);
    case GEMM_EPI_TILE_BF16: return launch_one<GEMM_EPI_TILE_BF16, SA, SB>(g, grid, st);
    case GEMM_EPI_TILE_WAVE_BF16: return launch_one<GEMM_EPI_TILE_WAVE_BF16, SA, SB>(g, grid, st);
    default: nsd_set_error("gemm_bf16: unknown epilogue %d", g.epi); return NSD_E_INVALID;
    }
}
int launch_epi(const GemmArgs &g, const dim3 grid, hipStream_t st) {
    // the operand with (many) more rows is the one streamed from HBM: it gets the third stage
    static const int force = [] { const char *e = getenv("NSD_GEMM_DMA_STAGES"); return e ? atoi(e) : 0; }();   // test hook: 22 / 23 / 32
    const int mode = force ? force : (g.M >= 4 * g.N ? 32 : (g.N >= 4 * g.M ? 23 : 22));
    if (mode == 32) return launch_stages<3, 2>(g, grid, st);
    if (mode == 23) return launch_stages<2, 3>(g, grid, st);
    return launch_stages<2, 2>(g, grid, st);
}
}  // namespace dma

template <bool AK, bool BK>
int launch_epi(const GemmArgs &g, const dim3 grid, hipStream_t st) {
    switch (g.epi) {
    case GEMM_EPI_F32: hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK, GEMM_EPI_F32>), grid, dim3(256), 0, st, g); break;
    case GEMM_EPI_BF16: hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK, GEMM_EPI_BF16>), grid, dim3(256), 0, st, g); break;
    case GEMM_EPI_TILE_BF16: hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK, GEMM_EPI_TILE_BF16>), grid, dim3(256), 0, st, g); break;
    case GEMM_EPI_TILE_WAVE_BF16: hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK, GEMM_EPI_TILE_WAVE_BF16>), grid, dim3(256), 0, st, g); break;
    default: nsd_set_error("gemm_bf16: unknown epilogue %d", g.epi); return NSD_E_INVALID;
    }
    NSD_CHECK_LAUNCH("gemm_bf16_kernel");
    return NSD_OK;
}

}  // namespace

// test hook: NSD_GEMM_SMALL_TILES=1 forces the 128 x 128 kernel (both kernels must give the same numbers)
// test hook: NSD_GEMM_REG_STAGING=1 keeps the register-staged 256 x 256 kernel for two k-contiguous operands (A/B runs, parity)
static bool getenv_reg_staging() { static const bool v = [] { const char *e = getenv("NSD_GEMM_REG_STAGING"); return e && e[0] == '1'; }(); return v; }
static bool getenv_small_tiles() { static const bool v = [] { const char *e = getenv("NSD_GEMM_SMALL_TILES"); return e && e[0] == '1'; }(); return v; }

int nsd_gemm_bf16_launch(const GemmArgs &g, hipStream_t st) {
    if (!g.A || !g.B || !g.C || g.M < 1 || g.N < 1 || g.K < 1) { nsd_set_error("gemm_bf16: null pointer or empty problem"); return NSD_E_INVALID; }
    // 16-byte pieces along the contiguous dimension of each operand
    if ((g.a_kmajor ? g.M % 8 : g.K % 8) || (g.b_kmajor ? g.N % 8 : g.K % 8) || g.lda % 8 || g.ldb % 8) {
        nsd_set_error("gemm_bf16: contiguous dimensions and leading dimensions must be multiples of 8 (M=%d N=%d K=%ld lda=%ld ldb=%ld)",
                      g.M, g.N, g.K, g.lda, g.ldb);
        return NSD_E_INVALID;
    }
    if ((g.epi == GEMM_EPI_TILE_BF16 || g.epi == GEMM_EPI_TILE_WAVE_BF16) && (g.M % 32 || g.N % 32)) { nsd_set_error("gemm_bf16: tile output needs M, N multiples of 32"); return NSD_E_INVALID; }
    const int splits = (g.epi == GEMM_EPI_F32 && g.splits > 1) ? g.splits : 1;
    if (g.b_shift != 0 && !g.b_kmajor) { nsd_set_error("gemm_bf16: b_shift needs a k-major B"); return NSD_E_INVALID; }
    // (|b_shift| == b_period is legal: no row has a partner inside its block -- one time step per batch tile -- and the product is zero)
    if (g.b_period < 0 || (g.b_period > 0 && (g.K >= (1L << 31) || g.b_shift > g.b_period || -g.b_shift > g.b_period))) { nsd_set_error("gemm_bf16: bad b_period"); return NSD_E_INVALID; }
    if (g.B2 && (!g.b_kmajor || g.n_split <= 0 || g.n_split >= g.N || g.n_split % 256 || g.ldb2 % 8 || (g.N - g.n_split) % 8)) {
        nsd_set_error("gemm_bf16: a second B source needs a k-major B and 0 < n_split < N, n_split a multiple of 256");
        return NSD_E_INVALID;
    }
    if (g.add && (g.epi != GEMM_EPI_F32 || splits != 1)) { nsd_set_error("gemm_bf16: addend needs the fp32 epilogue without split-K"); return NSD_E_INVALID; }
    GemmArgs a = g;
    a.splits = splits;
    // the 256 x 256 kernel where it fills the machine: at least one workgroup per CU (split-K included)
    const long big_wgs = (long)((g.N + big::TN - 1) / big::TN) * ((g.M + big::TM - 1) / big::TM) * splits;
    if (g.M >= big::TM && g.N >= big::TN && big_wgs >= nsd_num_cus() && !getenv_small_tiles()) {
        const int tiles = ((g.N + big::TN - 1) / big::TN) * ((g.M + big::TM - 1) / big::TM);
        const dim3 bgrid(tiles >= 16 ? 8 * ((tiles + 7) / 8) : tiles, 1, splits);   // 1-D; >= 16 tiles: padded to a multiple of 8, ids mapped XCD-aware
        if (g.a_kmajor) return g.b_kmajor ? big::launch_epi<true, true>(a, bgrid, st) : big::launch_epi<true, false>(a, bgrid, st);
        if (g.b_kmajor) return big::launch_epi<false, true>(a, bgrid, st);
        // both operands k-contiguous: the LDS-DMA kernel (lane offsets inside a tile's 256 rows must stay below the descriptor range)
        if (!getenv_reg_staging() && 256L * 2 * (g.lda > g.ldb ? g.lda : g.ldb) + 2 * g.K < 0x7f000000L) return dma::launch_epi(a, bgrid, st);
        return big::launch_epi<false, false>(a, bgrid, st);
    }
    const dim3 grid((g.N + GN - 1) / GN, (g.M + GM - 1) / GM, splits);
    if (g.a_kmajor) return g.b_kmajor ? launch_epi<true, true>(a, grid, st) : launch_epi<true, false>(a, grid, st);
    return g.b_kmajor ? launch_epi<false, true>(a, grid, st) : launch_epi<false, false>(a, grid, st);
}
