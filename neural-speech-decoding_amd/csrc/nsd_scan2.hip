// nsd_scan2.hip -- TWO unidirectional LSTM layers in one persistent launch, layer 1 one time step behind layer 0 (BASELINE
// cfg3: L = 2).  Same grouping, register residency and flag protocol as nsd_scan.hip (read its header first); what changes:
//
//  * a workgroup owns its 32 hidden units in BOTH layers and holds W_hh0, W_ih1 and W_hh1 rows in VGPRs (192 registers per
//    lane at H = 256).  At macro step s layer 0 advances to t = s and layer 1 to t = s - 1: both need only what the group
//    published at the end of step s - 1 (h0_{s-1}, its multiplied copy when dropout is on, h1_{s-2}), so the stack costs
//    T + 1 exchanges instead of 2 T, and layer 1's input projection W_ih1 . in1_t runs inside the scan (no GEMM, no
//    accumulator-tile round trip through HBM).
//  * backward: layer 1 works on t = T-1-s, layer 0 on t = T-s.  The tile da1_{t+1} the group exchanges serves twice: the
//    recurrent term of layer 1 (W_hh1^T) and the input gradient of layer 1 (W_ih1^T), which is layer 0's upstream gradient
//    -- the same B fragments feed two MFMAs; no input-gradient GEMM, no din round trip.
// Semantics as everywhere: torch.nn.LSTM(num_layers=2, dropout=p) of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34.
#include "nsd_scan_common.h"

namespace {

constexpr int ST2_FWD_TIMEOUT = 1, ST2_BWD_TIMEOUT = 2;

template <int H, int NT>
__global__ __launch_bounds__(256) void scan2_fwd_kernel(const Scan2FwdArgs a) {
    constexpr int KS = H / 16, P = H / 32, MG = 32 * NT, LDB = H + 8, G = 4 * H;
    constexpr int PIECES = MG * (H / 8) / 256;
    // [buffer s & 1][h0_{s-1} | in1_{s-1} (multiplied h0) | h1_{s-2}][trial][unit]: one barrier per step (see nsd_scan.hip)
    __shared__ __align__(16) bf16_t tiles[2][3][MG * LDB];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Member me = member_of(blockIdx.x, a.groups, P, a.spread_groups);
    const int gt = 4 * me.p + wave;
    const int b0 = (a.group0 + me.group) * MG;
    const int col = lane & 31, hh = lane >> 5;

    bf16x8 w0[KS], wx[KS], w1[KS];
    {
        const long ro = (long)(32 * gt + col) * H + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            w0[ks] = *reinterpret_cast<const bf16x8 *>(a.wf0 + ro + 16 * ks);
            wx[ks] = *reinterpret_cast<const bf16x8 *>(a.wx1 + ro + 16 * ks);
            w1[ks] = *reinterpret_cast<const bf16x8 *>(a.wf1 + ro + 16 * ks);
        }
    }
    float bias1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias1[r] = a.bsum1[32 * gt + mfma32_row(r, lane)];
    float c0[NT][4], c1[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0[nt][j] = 0.f; c1[nt][j] = 0.f; }
    if (tid == 0) s_abort = 0;
    for (int i = tid; i < 2 * 3 * MG * LDB / 8; i += 256) reinterpret_cast<u32x4 *>(&tiles[0][0][0])[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    unsigned *gflags = a.flags + (long)me.group * GROUP_WORDS;
    const int rv = group_rendezvous<P>(gflags, me.p, wave, lane);
    if (rv < 0 && lane == 0) { s_abort = 1; atomicExch(a.status, ST2_FWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);
    const long Bp = a.Bp;
    const int u0 = 8 * gt + 4 * hh;
    const bool train = a.cs0 != nullptr, masked = a.lk0 != nullptr;
    const int T = a.T;

    // layer-0 input projection tiles: the one of step s+1 is requested AFTER the gather of step s has landed (vector memory
    // returns in issue order: an HBM read issued ahead of the gather would put its latency on every step's critical path) and
    // AFTER the tile of step s has been unpacked into the accumulators -- requested before, hipcc guards the unpack with
    // vmcnt(0) and the wave sits out the whole HBM latency every step (2 300 of 9 200 cycles)
    u32x4 xp[NT][2];
    auto load_xp = [&](const int t, u32x4 (&dst)[NT][2]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16_t *src = a.xproj0 + ((((long)t * (Bp >> 5) + (b0 >> 5) + nt) * (G >> 5) + gt) * 64 + lane) * 16;
            dst[nt][0] = *reinterpret_cast<const u32x4 *>(src);
            dst[nt][1] = *reinterpret_cast<const u32x4 *>(src + 8);
        }
    };
    load_xp(0, xp);
    Stamps stp;
    stp.start();
    for (int s = 0; s <= T; ++s) {
        const bool do0 = s < T, do1 = s >= 1;
        const int t0 = s, t1 = s - 1;
        float mult[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) drop_mult4(a.rng, masked && a.rng.on && do0, 0, a.B, T, b0 + 32 * nt + col, t0, H, u0, mult[nt]);
        const bf16_t *TA = tiles[s & 1][0], *TB = tiles[s & 1][masked ? 1 : 0], *TC = tiles[s & 1][2];
        if (s >= 1) {
            if (!(NSD_SCAN_ABLATE & 1) && !wait_group<4 * P>(gflags, (unsigned)s, lane) && lane == 0) {
                s_abort = 1;
                atomicExch(a.status, ST2_FWD_TIMEOUT);
            }
            stp.mark(0);
            // the exchange ring: per tensor and step parity one block of MG*H bf16 per batch tile, laid out [gate tile = 4p+wave]
            // [tile nt][trial][8 units]: every producer wave writes its 512-byte blocks as WHOLE 128-byte lines with one store
            // instruction, and a consumer's 16-byte pieces are linear in the block (piece e = (unit group)*MG + trial).
            // (Exchanging through hs[t] itself -- 8-byte pieces of a line shared by 8 producer waves -- made every gather
            // load wait ~5 000 cycles: partially written lines are merged beyond the L2.)
            constexpr long XB = (long)MG * H;                   // elements per block
            const bf16_t *ring = a.xch + ((long)((s - 1) & 1) * a.groups_total + a.group0 + me.group) * 3 * XB;
            const nsd_rsrc rr = make_rsrc(ring, (unsigned)(3 * XB * 2));
            u32x4 pa[PIECES], pb[PIECES], pc[PIECES];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const unsigned off = (unsigned)((tid + 256 * i) * 16);
                pa[i] = ld_sc1_b128(rr, off);
                pb[i] = masked ? ld_sc1_b128(rr, (unsigned)(XB * 2) + off) : pa[i];
                pc[i] = s >= 2 ? ld_sc1_b128(rr, (unsigned)(2 * XB * 2) + off) : u32x4{0u, 0u, 0u, 0u};
            }
            stp.mark<true>(6);                                   // (diagnostic build) the gather loads have arrived
            bf16_t *WA = tiles[s & 1][0], *WB = tiles[s & 1][1], *WC = tiles[s & 1][2];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int e = tid + 256 * i, pc8 = e / MG, row = e % MG;          // piece e of the block: unit group pc8, trial row
                *reinterpret_cast<u32x4 *>(WA + row * LDB + 8 * pc8) = pa[i];
                if (masked) *reinterpret_cast<u32x4 *>(WB + row * LDB + 8 * pc8) = pb[i];
                if (s >= 2) *reinterpret_cast<u32x4 *>(WC + row * LDB + 8 * pc8) = pc[i];
            }
            stp.mark(7);                                         // ds_writes done
            __syncthreads();
            if (s_abort) break;
            stp.mark(1);
        }
        f32x16 acc0[NT], acc1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc0[nt] = unpack_tile(xp[nt][0], xp[nt][1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[nt][r] = bias1[r];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < T) load_xp(s + 1, xp);
        __builtin_amdgcn_sched_barrier(0);
        if (s >= 1) {
            // all three products every step, branch-free: at s == 1 the h1 tile is the zero state the buffers start with, at
            // s == T the layer-0 product feeds nothing (its cell is guarded by do0)
            const bf16x8 *const ws3[3] = {w0, wx, w1};
            const bf16_t *const ts3[3] = {TA, TB, TC};
            mfma_pipe<NT, KS, LDB, 3, 12>(ws3, ts3, col, hh, [&](const int st, const int nt) -> f32x16 & { return st == 0 ? acc0[nt] : acc1[nt]; });
        }
        stp.mark(2);
        // ---- the two cells; registers 4j..4j+3 = gates i,f,g,o of unit u0 + j for trial b0 + 32nt + col
        float g0[NT][4][4], g1[NT][4][4];
        unsigned hw0[NT][2], hw1[NT][2], lw0[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float h0v[4], h1v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g0[nt][j][0] = fast_sigmoid(acc0[nt][4 * j]); g0[nt][j][1] = fast_sigmoid(acc0[nt][4 * j + 1]);
                g0[nt][j][2] = fast_tanh(acc0[nt][4 * j + 2]); g0[nt][j][3] = fast_sigmoid(acc0[nt][4 * j + 3]);
                if (do0) c0[nt][j] = fmaf(g0[nt][j][1], c0[nt][j], g0[nt][j][0] * g0[nt][j][2]);
                h0v[j] = g0[nt][j][3] * fast_tanh(c0[nt][j]);
                g1[nt][j][0] = fast_sigmoid(acc1[nt][4 * j]); g1[nt][j][1] = fast_sigmoid(acc1[nt][4 * j + 1]);
                g1[nt][j][2] = fast_tanh(acc1[nt][4 * j + 2]); g1[nt][j][3] = fast_sigmoid(acc1[nt][4 * j + 3]);
                if (do1) c1[nt][j] = fmaf(g1[nt][j][1], c1[nt][j], g1[nt][j][0] * g1[nt][j][2]);
                h1v[j] = g1[nt][j][3] * fast_tanh(c1[nt][j]);
            }
            hw0[nt][0] = pack_bf16x2(h0v[0], h0v[1]); hw0[nt][1] = pack_bf16x2(h0v[2], h0v[3]);
            hw1[nt][0] = pack_bf16x2(h1v[0], h1v[1]); hw1[nt][1] = pack_bf16x2(h1v[2], h1v[3]);
            // the multiplier acts on the value layer 1 really reads: the bf16 h0
            lw0[nt][0] = pack_bf16x2(bf16_lo(hw0[nt][0]) * mult[nt][0], bf16_hi(hw0[nt][0]) * mult[nt][1]);
            lw0[nt][1] = pack_bf16x2(bf16_lo(hw0[nt][1]) * mult[nt][2], bf16_hi(hw0[nt][1]) * mult[nt][3]);
        }
        stp.mark(3);
        // ---- publish h0_t0, its multiplied copy, h1_t1 into the ring slot of this step
        {
            constexpr long XB = (long)MG * H;
            bf16_t *ring = a.xch + ((long)(s & 1) * a.groups_total + a.group0 + me.group) * 3 * XB;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const long off = ((long)(gt * NT + nt) * 32 + col) * 8 + 4 * hh;       // [gate tile][nt][trial][8 units]
                if (do0) {
                    st_xchg_u64(same_l2, ring + off, ((unsigned long long)hw0[nt][1] << 32) | hw0[nt][0]);
                    if (masked) st_xchg_u64(same_l2, ring + XB + off, ((unsigned long long)lw0[nt][1] << 32) | lw0[nt][0]);
                }
                if (do1) st_xchg_u64(same_l2, ring + 2 * XB + off, ((unsigned long long)hw1[nt][1] << 32) | hw1[nt][0]);
            }
        }
        if (!(NSD_SCAN_ABLATE & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_xchg_u32(same_l2, gflags + 4 * me.p + wave, (unsigned)(s + 1));
        stp.mark(4);
        // ---- row-major copies (the head reads hs1; the weight-gradient GEMMs read hs0 / lk0 / hs1) and the saves for the
        // backward pass leave behind the flag: nobody waits for them inside this launch
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int b = b0 + 32 * nt + col;
            if (do0 && train) {
                const long row = (long)t0 * Bp + b;
                *reinterpret_cast<u32x2 *>(a.hs0 + row * H + u0) = u32x2{hw0[nt][0], hw0[nt][1]};
                if (masked) *reinterpret_cast<u32x2 *>(a.lk0 + row * H + u0) = u32x2{lw0[nt][0], lw0[nt][1]};
            }
            if (do1) *reinterpret_cast<u32x2 *>(a.hs1 + ((long)t1 * Bp + b) * H + u0) = u32x2{hw1[nt][0], hw1[nt][1]};
        }
        if (train) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int b = b0 + 32 * nt + col;
                if (do0) {
                    const long row = (long)t0 * Bp + b;
                    *reinterpret_cast<u32x2 *>(a.cs0 + row * H + u0) = u32x2{pack_bf16x2(c0[nt][0], c0[nt][1]), pack_bf16x2(c0[nt][2], c0[nt][3])};
                    bf16_t *gd = a.ga0 + row * G + 4 * u0;
                    *reinterpret_cast<u32x4 *>(gd) = u32x4{pack_bf16x2(g0[nt][0][0], g0[nt][0][1]), pack_bf16x2(g0[nt][0][2], g0[nt][0][3]),
                                                           pack_bf16x2(g0[nt][1][0], g0[nt][1][1]), pack_bf16x2(g0[nt][1][2], g0[nt][1][3])};
                    *reinterpret_cast<u32x4 *>(gd + 8) = u32x4{pack_bf16x2(g0[nt][2][0], g0[nt][2][1]), pack_bf16x2(g0[nt][2][2], g0[nt][2][3]),
                                                               pack_bf16x2(g0[nt][3][0], g0[nt][3][1]), pack_bf16x2(g0[nt][3][2], g0[nt][3][3])};
                }
                if (do1) {
                    const long row = (long)t1 * Bp + b;
                    *reinterpret_cast<u32x2 *>(a.cs1 + row * H + u0) = u32x2{pack_bf16x2(c1[nt][0], c1[nt][1]), pack_bf16x2(c1[nt][2], c1[nt][3])};
                    bf16_t *gd = a.ga1 + row * G + 4 * u0;
                    *reinterpret_cast<u32x4 *>(gd) = u32x4{pack_bf16x2(g1[nt][0][0], g1[nt][0][1]), pack_bf16x2(g1[nt][0][2], g1[nt][0][3]),
                                                           pack_bf16x2(g1[nt][1][0], g1[nt][1][1]), pack_bf16x2(g1[nt][1][2], g1[nt][1][3])};
                    *reinterpret_cast<u32x4 *>(gd + 8) = u32x4{pack_bf16x2(g1[nt][2][0], g1[nt][2][1]), pack_bf16x2(g1[nt][2][2], g1[nt][2][3]),
                                                               pack_bf16x2(g1[nt][3][0], g1[nt][3][1]), pack_bf16x2(g1[nt][3][2], g1[nt][3][3])};
                }
            }
        }
        stp.mark(5);
    }
    stp.store(a.status, blockIdx.x == 0 && tid == 0);
}

__device__ __forceinline__ void st_xchg_b128x2(const bool same_l2, const nsd_rsrc rs, const unsigned off, const unsigned (&dw)[8]) {
    if (same_l2) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{dw[0], dw[1], dw[2], dw[3]}, rs, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{dw[4], dw[5], dw[6], dw[7]}, rs, (int)off + 16, 0, 0);
    } else {
        st_sc1_b128(rs, off, u32x4{dw[0], dw[1], dw[2], dw[3]});
        st_sc1_b128(rs, off + 16u, u32x4{dw[4], dw[5], dw[6], dw[7]});
    }
}

// one layer's cell backward for a lane's 4 units of one trial; returns da (16 values, unit-major) packed + accumulates dbs
__device__ __forceinline__ void cell_bwd4(const u32x4 gq0, const u32x4 gq1, const u32x2 cq, const u32x2 cpq, const float (&dh)[4], float (&dc)[4],
                                          float (&dbs)[16], unsigned (&dw)[8]) {
    const float cv[4] = {bf16_lo(cq[0]), bf16_hi(cq[0]), bf16_lo(cq[1]), bf16_hi(cq[1])};
    const float cp[4] = {bf16_lo(cpq[0]), bf16_hi(cpq[0]), bf16_lo(cpq[1]), bf16_hi(cpq[1])};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned w0 = j < 2 ? gq0[2 * (j & 1)] : gq1[2 * (j & 1)], w1 = j < 2 ? gq0[2 * (j & 1) + 1] : gq1[2 * (j & 1) + 1];
        const float ig = bf16_lo(w0), fg = bf16_hi(w0), gg = bf16_lo(w1), og = bf16_hi(w1);
        const float tc = fast_tanh(cv[j]);
        const float dct = fmaf(dh[j] * og, 1.f - tc * tc, dc[j]);
        dc[j] = dct * fg;
        const float dai = dct * gg * ig * (1.f - ig);
        const float daf = dct * cp[j] * fg * (1.f - fg);
        const float dag = dct * ig * (1.f - gg * gg);
        const float dao = dh[j] * tc * og * (1.f - og);
        dw[2 * j] = pack_bf16x2(dai, daf);
        dw[2 * j + 1] = pack_bf16x2(dag, dao);
        dbs[4 * j] += dai; dbs[4 * j + 1] += daf; dbs[4 * j + 2] += dag; dbs[4 * j + 3] += dao;
    }
}

template <int H, int NT>
__global__ __launch_bounds__(256) void scan2_bwd_kernel(const Scan2BwdArgs a) {
    constexpr int P = H / 32, MG = 32 * NT, G = 4 * H, KQ = G / 16 / 4;
    constexpr int NX = P * NT * 2, NB = 2 * NX, NSLOT = 8, FA = 3;   // 1-KB blocks (= MFMA k-steps) of a wave's quarter: X, then Y
    static_assert(NB >= NSLOT, "");
    // partial tiles of the 4 waves: [buffer][wave][rec1 | din0 | rec0][tile][unit of the workgroup][trial]
    __shared__ __align__(16) float red2[2][4][3][NT][32][32];
    __shared__ __align__(16) bf16_t dring[4][NSLOT][512];      // per wave: LDS-DMA landing ring of 1-KB blocks
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Member me = member_of(blockIdx.x, a.groups, P, a.spread_groups);
    const int b0 = (a.group0 + me.group) * MG;
    const int col = lane & 31, hh = lane >> 5;

    bf16x8 wq1[KQ], wqx[KQ], wq0[KQ];
    {
        // k-step ks = 2 * (gate tile) + half carries, in lane half hh, the columns 16hh + 8half + 0..7 of the gate tile: the order
        // of the ring's 1-KB blocks (nsd_scan_common.h)
        const long ro = (long)(32 * me.p + col) * G + wave * (G / 4) + 16 * hh;
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            const int co = 32 * (ks >> 1) + 8 * (ks & 1);
            wq1[ks] = *reinterpret_cast<const bf16x8 *>(a.wb1 + ro + co);
            wqx[ks] = *reinterpret_cast<const bf16x8 *>(a.wxt1 + ro + co);
            wq0[ks] = *reinterpret_cast<const bf16x8 *>(a.wb0 + ro + co);
        }
    }
    float dc1[NT][4], dc0[NT][4], dbs1[16], dbs0[16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dc1[nt][j] = 0.f; dc0[nt][j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 16; ++k) { dbs1[k] = 0.f; dbs0[k] = 0.f; }
    if (tid == 0) s_abort = 0;
    __syncthreads();

    unsigned *gflags = a.flags + (long)me.group * GROUP_WORDS;
    const int rv = group_rendezvous<P>(gflags, me.p, wave, lane);
    if (rv < 0 && lane == 0) { s_abort = 1; atomicExch(a.status, ST2_BWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);
    const long Bp = a.Bp;
    const int T = a.T;
    const int u0 = 32 * me.p + 8 * wave + 4 * hh;
    float dpl[NT][4], aw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aw[j] = a.attn_w[u0 + j];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dpl[nt][j] = a.dpooled[(long)(b0 + 32 * nt + col) * H + u0 + j];
    bf16_t (*land)[512] = dring[wave];
    const unsigned land_addr = __builtin_amdgcn_readfirstlane(lds_addr_of(&dring[wave][0][0]));
    constexpr long XB = (long)MG * G;                          // elements of one da block of the exchange ring

    // Saved activations / upstream terms of a step (HBM reads, independent of the recurrence).  Vector memory returns in issue
    // order, so an HBM read issued just before the flag poll or before the publishing drain puts its whole latency on the
    // step (6 400 of 16 500 cycles when these loads sat at the top of the step).  The set of step s+1 is therefore requested in
    // the middle of step s, right after the last staged tile has landed: ~5 000 cycles of MFMA, reduction and cell work follow
    // before the wave waits on memory again.
    struct Saved {
        u32x4 q1[NT][2], q0[NT][2];
        u32x2 cq1[NT], cp1[NT], cq0[NT], cp0[NT];
        float al[NT], ds[NT];
    };
    auto load_saved = [&](const int sx, Saved &v) {
        const bool d1 = sx < T, d0 = sx >= 1;
        const int x1 = T - 1 - sx, x0 = T - sx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int b = b0 + 32 * nt + col;
            if (d1) {
                const long row = (long)x1 * Bp + b;
                const bf16_t *gs = a.ga1 + row * G + 4 * u0;
                v.q1[nt][0] = *reinterpret_cast<const u32x4 *>(gs); v.q1[nt][1] = *reinterpret_cast<const u32x4 *>(gs + 8);
                v.cq1[nt] = *reinterpret_cast<const u32x2 *>(a.cs1 + row * H + u0);
                v.cp1[nt] = x1 == 0 ? u32x2{0u, 0u} : *reinterpret_cast<const u32x2 *>(a.cs1 + ((long)(x1 - 1) * Bp + b) * H + u0);
                v.al[nt] = a.alpha[row]; v.ds[nt] = a.dscore[row];
            } else {
                v.q1[nt][0] = u32x4{0u, 0u, 0u, 0u}; v.q1[nt][1] = u32x4{0u, 0u, 0u, 0u}; v.cq1[nt] = u32x2{0u, 0u}; v.cp1[nt] = u32x2{0u, 0u};
                v.al[nt] = 0.f; v.ds[nt] = 0.f;
            }
            if (d0) {
                const long row = (long)x0 * Bp + b;
                const bf16_t *gs = a.ga0 + row * G + 4 * u0;
                v.q0[nt][0] = *reinterpret_cast<const u32x4 *>(gs); v.q0[nt][1] = *reinterpret_cast<const u32x4 *>(gs + 8);
                v.cq0[nt] = *reinterpret_cast<const u32x2 *>(a.cs0 + row * H + u0);
                v.cp0[nt] = x0 == 0 ? u32x2{0u, 0u} : *reinterpret_cast<const u32x2 *>(a.cs0 + ((long)(x0 - 1) * Bp + b) * H + u0);
            } else {
                v.q0[nt][0] = u32x4{0u, 0u, 0u, 0u}; v.q0[nt][1] = u32x4{0u, 0u, 0u, 0u}; v.cq0[nt] = u32x2{0u, 0u}; v.cp0[nt] = u32x2{0u, 0u};
            }
        }
    };
    Saved sv, svn;
    load_saved(0, sv);
    Stamps stp;
    stp.start();
    for (int s = 0; s <= T; ++s) {
        const bool do1 = s < T, do0 = s >= 1;
        const int t1 = T - 1 - s, t0 = T - s;
        float dup1[NT][4], m0[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dup1[nt][j] = do1 ? fmaf(sv.al[nt], dpl[nt][j], sv.ds[nt] * aw[j]) : 0.f;
            drop_mult4(a.rng, a.rng.on != 0 && do0, 0, a.B, T, b0 + 32 * nt + col, t0, H, u0, m0[nt]);
        }
        if (s == 0 && T >= 1) load_saved(1, svn);              // (step 0 has no exchange phase to wait behind)
        float drec1[NT][4], dinx[NT][4], drec0[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { drec1[nt][j] = 0.f; dinx[nt][j] = 0.f; drec0[nt][j] = 0.f; }
        if (s >= 1) {
            float (*red)[3][NT][32][32] = red2[s & 1];
            if (!(NSD_SCAN_ABLATE & 1) && !wait_group<4 * P>(gflags, (unsigned)s, lane) && lane == 0) {
                s_abort = 1;
                atomicExch(a.status, ST2_BWD_TIMEOUT);
            }
            stp.mark(0);
            f32x16 aR1[NT], aX0[NT], aR0[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { aR1[nt] = zero16(); aX0[nt] = zero16(); aR0[nt] = zero16(); }
            // the ring slot the group filled at step s-1: [X = da1_{t0} | Y = da0_{t0+1}].  X feeds the recurrent term of layer 1
            // AND layer 1's input gradient (= layer 0's upstream term): the same fragment, two MFMAs; Y the recurrent term of
            // layer 0 (zeros at s == 1: step 0 publishes them).  This wave's quarter of the columns is NX contiguous 1-KB blocks
            // of each; they come by LDS-DMA, NSLOT in flight, the fragment read FA blocks ahead of its MFMA -- after the first
            // block has landed the matrix pipe does not wait (register staging + ds_write cost 6 800 cycles of a 13 500-cycle
            // step for 1 540 cycles of MFMA).
            const bf16_t *ring = a.xch + ((long)((s - 1) & 1) * a.groups_total + a.group0 + me.group) * 2 * XB + ((long)wave * NX * 64 + lane) * 8;
            auto src_of = [&](const int j) { return ring + (j < NX ? (long)j * 512 : XB + (long)(j - NX) * 512); };
            auto frag_of = [&](const int j) { return *reinterpret_cast<const bf16x8 *>(&land[j % NSLOT][lane * 8]); };
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) dma_block_sc1(src_of(j), land_addr + 1024u * j);
            bf16x8 fr[FA + 1];
#pragma unroll
            for (int j = 0; j < FA; ++j) { wait_vm(NSLOT - 1 - j); fr[j] = frag_of(j); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j + FA < NB) {
                    const int issued = (NSLOT + j) < NB ? (NSLOT + j) : NB;
                    wait_vm(issued - 1 - (j + FA));
                    fr[(j + FA) % (FA + 1)] = frag_of(j + FA);
                    // the last block has landed: request the next step's saved set (HBM) -- nothing of this step waits behind it
                    if (j + FA == NB - 1 && s + 1 <= T) load_saved(s + 1, svn);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int jj = j < NX ? j : j - NX, half = jj & 1, nt = (jj >> 1) % NT, ks = 2 * (jj / (2 * NT)) + half;
                    if (j < NX) {
                        aR1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq1[ks], fr[j % (FA + 1)], aR1[nt], 0, 0, 0);
                        aX0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqx[ks], fr[j % (FA + 1)], aX0[nt], 0, 0, 0);
                    } else {
                        aR0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq0[ks], fr[j % (FA + 1)], aR0[nt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (j + NSLOT < NB) dma_block_sc1(src_of(j + NSLOT), land_addr + 1024u * (j % NSLOT));
                if (j == NX - 1) stp.mark(1);
            }
            stp.mark(2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = mfma32_row(r, lane);
                    red[wave][0][nt][n][col] = aR1[nt][r];
                    red[wave][1][nt][n][col] = aX0[nt][r];
                    red[wave][2][nt][n][col] = aR0[nt][r];
                }
            stp.mark(3);
            __syncthreads();
            if (s_abort) break;
            stp.mark(4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = 8 * wave + 4 * hh + j;
                    drec1[nt][j] = (red[0][0][nt][n][col] + red[1][0][nt][n][col]) + (red[2][0][nt][n][col] + red[3][0][nt][n][col]);
                    dinx[nt][j] = (red[0][1][nt][n][col] + red[1][1][nt][n][col]) + (red[2][1][nt][n][col] + red[3][1][nt][n][col]);
                    drec0[nt][j] = (red[0][2][nt][n][col] + red[1][2][nt][n][col]) + (red[2][2][nt][n][col] + red[3][2][nt][n][col]);
                }
        }
        stp.mark<true>(5);                                       // (diagnostic build: + arrival of this step's saved activations)
        // ---- cell backward of both layers; da1_{t1}, da0_{t0} -> ring slot of this step (exchange), row-major copies for the
        // weight-gradient GEMMs behind the flag
        unsigned dw1[NT][8], dw0[NT][8];
        bf16_t *slot = a.xch + ((long)(s & 1) * a.groups_total + a.group0 + me.group) * 2 * XB;
        const int gtw = 4 * me.p + wave;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (do1) {
                float dh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dh[j] = dup1[nt][j] + drec1[nt][j];
                cell_bwd4(sv.q1[nt][0], sv.q1[nt][1], sv.cq1[nt], sv.cp1[nt], dh, dc1[nt], dbs1, dw1[nt]);
                ring_put_da(same_l2, slot, gtw, nt, NT, col, hh, dw1[nt]);
            }
            if (do0) {
                float dh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dh[j] = fmaf(dinx[nt][j], m0[nt][j], drec0[nt][j]);
                cell_bwd4(sv.q0[nt][0], sv.q0[nt][1], sv.cq0[nt], sv.cp0[nt], dh, dc0[nt], dbs0, dw0[nt]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) dw0[nt][j] = 0u;    // step 0: the Y block step 1 reads is da0_{T} = 0
            }
            ring_put_da(same_l2, slot + XB, gtw, nt, NT, col, hh, dw0[nt]);
        }
        stp.mark(6);
        if (!(NSD_SCAN_ABLATE & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_xchg_u32(same_l2, gflags + 4 * me.p + wave, (unsigned)(s + 1));
        stp.mark(7);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long rb = b0 + 32 * nt + col;
            if (do1) {
                bf16_t *d = a.da1 + ((long)t1 * Bp + rb) * G + 4 * u0;
                *reinterpret_cast<u32x4 *>(d) = u32x4{dw1[nt][0], dw1[nt][1], dw1[nt][2], dw1[nt][3]};
                *reinterpret_cast<u32x4 *>(d + 8) = u32x4{dw1[nt][4], dw1[nt][5], dw1[nt][6], dw1[nt][7]};
            }
            if (do0) {
                bf16_t *d = a.da0 + ((long)t0 * Bp + rb) * G + 4 * u0;
                *reinterpret_cast<u32x4 *>(d) = u32x4{dw0[nt][0], dw0[nt][1], dw0[nt][2], dw0[nt][3]};
                *reinterpret_cast<u32x4 *>(d + 8) = u32x4{dw0[nt][4], dw0[nt][5], dw0[nt][6], dw0[nt][7]};
            }
        }
        sv = svn;
    }
    stp.store(a.status, blockIdx.x == 0 && tid == 0);
    // ---- bias gradients of this batch tile, both layers
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float v1 = dbs1[k], v0 = dbs0[k];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) { v1 += __shfl_xor(v1, m, 64); v0 += __shfl_xor(v0, m, 64); }
        dbs1[k] = v1; dbs0[k] = v0;
    }
    if (col == 0) {
        float *d1 = a.dbp1 + (long)(a.group0 + me.group) * G + 4 * u0, *d0 = a.dbp0 + (long)(a.group0 + me.group) * G + 4 * u0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            *reinterpret_cast<f32x4 *>(d1 + k) = f32x4{dbs1[k], dbs1[k + 1], dbs1[k + 2], dbs1[k + 3]};
            *reinterpret_cast<f32x4 *>(d0 + k) = f32x4{dbs0[k], dbs0[k + 1], dbs0[k + 2], dbs0[k + 3]};
        }
    }
}

}  // namespace

// LDS of the forward kernel: 2 x 3 tiles of MG x (H + 8) bf16; of the backward kernel: 96 KB x NT of partials + 32 KB of DMA landing rings
bool nsd_scan2_supported(int H, int MG) {
    if (!(H == 64 || H == 128 || H == 256)) return false;
    const long fwd = 2L * 3 * MG * (H + 8) * 2, bwd = 2L * 4 * 3 * (MG / 32) * 4096 + 4L * 8 * 1024;
    return fwd <= 150 * 1024 && bwd <= 150 * 1024;
}

template <int H>
static int launch2_fwd(const Scan2FwdArgs &a, int MG, const dim3 grid, hipStream_t st) {
    if (MG == 32) hipLaunchKernelGGL((scan2_fwd_kernel<H, 1>), grid, dim3(256), 0, st, a);
    else { nsd_set_error("scan2_fwd: batch tile %d not built", MG); return NSD_E_INVALID; }
    NSD_CHECK_LAUNCH("scan2_fwd_kernel");
    return NSD_OK;
}
template <int H>
static int launch2_bwd(const Scan2BwdArgs &a, int MG, const dim3 grid, hipStream_t st) {
    if (MG == 32) hipLaunchKernelGGL((scan2_bwd_kernel<H, 1>), grid, dim3(256), 0, st, a);
    else { nsd_set_error("scan2_bwd: batch tile %d not built", MG); return NSD_E_INVALID; }
    NSD_CHECK_LAUNCH("scan2_bwd_kernel");
    return NSD_OK;
}

int nsd_scan2_fwd_launch(const Scan2FwdArgs &a, int H, int MG, hipStream_t st) {
    if (!nsd_scan2_supported(H, MG) || a.groups * (H / 32) > nsd_num_cus()) { nsd_set_error("scan2_fwd: unsupported geometry H=%d MG=%d groups=%d", H, MG, a.groups); return NSD_E_INVALID; }
    const dim3 grid(a.groups * (H / 32));
    switch (H) {
    case 64: return launch2_fwd<64>(a, MG, grid, st);
    case 128: return launch2_fwd<128>(a, MG, grid, st);
    default: return launch2_fwd<256>(a, MG, grid, st);
    }
}
int nsd_scan2_bwd_launch(const Scan2BwdArgs &a, int H, int MG, hipStream_t st) {
    if (!nsd_scan2_supported(H, MG) || a.groups * (H / 32) > nsd_num_cus()) { nsd_set_error("scan2_bwd: unsupported geometry H=%d MG=%d groups=%d", H, MG, a.groups); return NSD_E_INVALID; }
    const dim3 grid(a.groups * (H / 32));
    switch (H) {
    case 64: return launch2_bwd<64>(a, MG, grid, st);
    case 128: return launch2_bwd<128>(a, MG, grid, st);
    default: return launch2_bwd<256>(a, MG, grid, st);
    }
}
