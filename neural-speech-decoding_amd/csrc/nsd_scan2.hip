// nsd_scan2.hip -- TWO unidirectional LSTM layers in one persistent launch, layer 1 one time step behind layer 0 (BASELINE
// cfg3: L = 2).  Same grouping, register residency and exchange protocols as nsd_scan.hip (read its header first: the forward
// scans exchange self-validating tagged granules, the backward scans flags + consume counters); what changes:
//
//  * a workgroup owns its 32 hidden units in BOTH layers and holds W_hh0, W_ih1 and W_hh1 rows in AGPRs (192 registers per
//    lane at H = 256).  At macro step s layer 0 advances to t = s and layer 1 to t = s - 1: both need only what the group
//    published at the end of step s - 1 (h0_{s-1}, its multiplied copy when dropout is on, h1_{s-2}), so the stack costs
//    T + 1 exchanges instead of 2 T, and layer 1's input projection W_ih1 . in1_t runs inside the scan (no GEMM, no
//    accumulator-tile round trip through HBM).
//  * backward: layer 1 works on t = T-1-s, layer 0 on t = T-s.  What a workgroup needs from the group at step s -- W_hh1^T da1
//    (recurrent term of layer 1), W_ih1^T da1 (input gradient of layer 1 = layer 0's upstream gradient) and W_hh0^T da0 -- all
//    derive from what the members computed at step s-1, and is exchanged as partial sums (see "backward" below): no
//    input-gradient GEMM, no din round trip.
//  * the layer-0 input projection W_ih0 x_t (K = channels padded to 16, at most 64) is one to four MFMAs per step inside the scan.
// Semantics as everywhere: torch.nn.LSTM(num_layers=2, dropout=p) of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34.
#include "nsd_scan_common.h"

namespace {

constexpr int ST2_FWD_TIMEOUT = 1, ST2_BWD_TIMEOUT = 2;
#ifndef NSD_LOOK_POS
#define NSD_LOOK_POS 1
#endif
// In inference a step has no saves between its publishing store and the first look at the group's granules: without a pause 0.7 looks
// per step come too early and cost a second L2 round trip and 32 KB per CU of L2 traffic each (counted in the diagnostic build).
// s_sleep 4 (256 cycles): 1 024 windows x 250 steps in 886-891 us against 907-922 without, 920 at 6-8, 935 at 12.
#ifndef NSD_LOOK_DELAY_INFER
#define NSD_LOOK_DELAY_INFER 4
#endif
#ifndef NSD_LOOK_DELAY
#define NSD_LOOK_DELAY 0
#endif

template <int H, int NT>
__global__ __launch_bounds__(256) void scan2_fwd_kernel(const Scan2FwdArgs a) {
    constexpr int KS = H / 16, P = H / 32, MG = 32 * NT, LDB = H + 8;
    constexpr int PIECES = MG * (H / 4) / 256;                  // 16-byte granules of the exchange ring per thread (one per producer lane)
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
    // Dropout between the layers: layer 1 reads h0 (x) m * keep.  The consumers of the exchange rebuild h0 (x) m with two bit
    // operations from the keep / drop bits that travel in the granules, and the scale sits in the weights: W_ih1 * keep, rounded to
    // bf16 once here (the row-major copy for the weight-gradient GEMM stays bf16(h0 * m * keep), as the producer forms it).
    if (a.lk0 != nullptr && a.rng.on) {
        const float keep = a.rng.keep_lstm;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) wx[ks][e] = (bf16_t)((float)wx[ks][e] * keep);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(wx[ks]));        // (the scaled rows live in AGPRs like the loaded ones: no copy per MFMA)
    float bias1[16], bias0[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { bias1[r] = a.bsum1[32 * gt + mfma32_row(r, lane)]; bias0[r] = a.bsum0[32 * gt + mfma32_row(r, lane)]; }
    // the layer-0 input projection rides in the scan: K = CP (the channels, padded to 16) is 1 to 4 k-steps
    const int CP = a.CP, ks0 = CP >> 4;
    bf16x8 wx0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(a.wx0 + (long)(32 * gt + col) * CP + 16 * (k < ks0 ? k : 0) + 8 * hh);
        wx0[k] = k < ks0 ? wv : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};             // (k-steps beyond the padded channel count: zero weights)
    }
    float c0[NT][4], c1[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0[nt][j] = 0.f; c1[nt][j] = 0.f; }
    if (tid == 0) s_abort = 0;
    for (int i = tid; i < 2 * 3 * MG * LDB / 8; i += 256) reinterpret_cast<u32x4 *>(&tiles[0][0][0])[i] = u32x4{0u, 0u, 0u, 0u};
    // ---- the exchange ring of the group (two slots, step parity): ONE 16-byte granule per producer lane and step,
    //   [gate tile = 4p + wave][nt][trial][half] x {h0 of the lane's 4 units | h1 of the same units},
    // and the granule carries its own validity: |h| < 1, so bit 14 of every bf16 h (set only for |x| >= 2) is free -- in the h1 half it
    // holds the TAG of the step that wrote the granule ((s >> 1) & 1: it flips every time a slot is rewritten), in the h0 half the
    // keep / drop bit of the unit's dropout multiplier (the multiplied copy layer 1 reads is rebuilt by the consumer, bit-identically).
    // A consumer simply loads its granules until every tag is the expected one: one L2 round trip behind the slowest producer's
    // store, where "stores -> drain -> flag -> poll -> loads" was three (round 2: ~3 800 of a step's ~9 900 cycles were protocol).
    // A lane's 16 bytes are written by one store instruction and read by one load: they arrive together -- measured, not assumed:
    // tools/micro/granule_litmus.hip, profiles/r04_granule_litmus.md (1.6e9 observations of granules being rewritten under the loads,
    // in both exchange modes and across XCDs: no torn granule, no torn 8-byte halves).  The slots start with the
    // tag their first writer will NOT use (the workspace may hold a previous launch's granules), drained before the start barrier.
    // NON-FINITE h (a NaN / Inf window, diverged or NaN weights: torch's nn.LSTM of lstm_eeg_model.py:34 propagates them) has bit 14
    // set by itself -- 0x7FC0, 0x7F80 -- and would read as a wrong tag (a consumer spinning to the time-out) or, masked, as 1.5.  A
    // producer lane therefore checks its packed words: once one of its values of a trial is not finite the lane is POISONED for that
    // trial -- it publishes zeros (right tag: nobody waits, nobody computes on a fake finite value that matters) and writes NaN into
    // every row-major h it owns from then on, so the head's pooling turns all logits of THAT trial into NaN exactly as the
    // reference does, the weight gradients become NaN as the reference's do, and the evaluation's status word gets NSD_SEQ_ST_NONFINITE.
    constexpr unsigned TAGBITS = 0x40004000u;
    constexpr long XG = (long)MG * H * 2;                       // bf16 elements of one slot: MG * H / 4 granules of 8
    bf16_t *const ring0 = a.xch + (long)(a.group0 + me.group) * XG, *const ring1 = ring0 + (long)a.groups_total * XG;
    const long gran_off = (((long)(4 * me.p + wave) * NT) * 32 + col) * 16 + 8 * hh;       // nt = 0; + 512 per nt
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        st_sc1_b128(make_rsrc(ring0, (unsigned)(XG * 2)), (unsigned)((gran_off + 512 * nt) * 2), u32x4{0u, 0u, TAGBITS, TAGBITS});
        st_sc1_b128(make_rsrc(ring1, (unsigned)(XG * 2)), (unsigned)((gran_off + 512 * nt) * 2), u32x4{0u, 0u, TAGBITS, TAGBITS});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned *gflags = a.flags + (long)me.group * GROUP_WORDS;
    const int rv = group_rendezvous<P>(gflags, me.p, wave, lane);
    if (rv < 0 && lane == 0) { s_abort = 1; report_timeout(a.status, ST2_FWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);
    const int u0 = 8 * gt + 4 * hh;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);    // (wave-uniform copy for address bases)
    const bool train = a.cs0 != nullptr, masked = a.lk0 != nullptr;
    const int T = a.T;

    // x_t of the tile as MFMA B fragments (lane (trial, hh): channels 16k + 8hh .. + 7: the wave reads one contiguous KB of xbf per
    // k-step).  The fragments of step s+1 are requested AFTER the gather of step s has landed (vector memory returns in issue
    // order: an HBM read issued ahead of the gather would put its latency on the step) and AFTER the MFMAs that read the current
    // ones have been issued.
    bf16x8 xf[NT][4];
    auto load_x = [&](const int t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {                       // (wave-uniform 64-bit base + 32-bit lane offset, as in the backward scan)
            const bf16_t *src = a.xbf + (((long)(b0 >> 5) + nt) * T + t) * 32 * CP;
            const unsigned lo = (unsigned)(col * CP + 8 * hh);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < ks0) xf[nt][k] = ld_stream<bf16x8>(src + lo + 16 * k);
        }
    };
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) xf[nt][k] = wx0[0];        // (defined values in the unused k-steps)
    load_x(0);
    u32x4 pg[PIECES];                                           // the granules a thread gathers (requested a step ahead, see "publish")
#pragma unroll
    for (int i = 0; i < PIECES; ++i) pg[i] = u32x4{0u, 0u, 0u, 0u};
    unsigned poison[NT];                                        // 0, or 0x7FC07FC0 once a value of this lane's units of trial (nt, col) was not finite
    {
        // W_hh0 . h_{-1} is not formed at s = 0 (h_{-1} = 0): a non-finite W_hh0 row would go unnoticed for a step where the reference
        // has NaN at once (W_hh1 and W_ih1 meet their zero / first tiles in the MFMAs of s = 1: IEEE does the rest)
        bool wbad = false;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wbad = wbad || frag_nonfinite(w0[ks]);
        wbad = __any(wbad);                                     // (a lane holds a weight ROW; the row acts on every trial = every lane of the wave)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) poison[nt] = wbad ? 0x7FC07FC0u : 0u;
    }
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
            // the granules of step s-1, loaded until every tag says so (bounded: a group that cannot complete reports and leaves)
            const nsd_rsrc rr = make_rsrc(((s - 1) & 1) ? ring1 : ring0, (unsigned)(XG * 2));
            const unsigned want = (((s - 1) >> 1) & 1) ? 0x4000u : 0u;
            // (the first look was requested right behind the publishing stores of step s-1, ahead of that step's saves)
            bool ok = false;
            for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
                bool mine = true;
#pragma unroll
                for (int i = 0; i < PIECES; ++i) mine = mine && ((pg[i][2] & 0x4000u) == want);
                ok = __all(mine) || (NSD_SCAN_ABLATE & 1) != 0;
                if (ok) break;
                if (NSD_SCAN_STAMPS) stp.acc[0] += 1000;             // (diagnostic build: looks that failed, x 1000)
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int i = 0; i < PIECES; ++i) pg[i] = ld_sc1_b128(rr, (unsigned)((tid + 256 * i) * 16));
            }
            if (!ok && lane == 0) {
                s_abort = 1;
                report_timeout(a.status, ST2_FWD_TIMEOUT);
            }
            stp.mark<true>(6);                                   // (diagnostic build) the granules have arrived
            bf16_t *WA = tiles[s & 1][0], *WB = tiles[s & 1][1], *WC = tiles[s & 1][2];
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int e = tid + 256 * i, ghh = e & 1, row = ((e >> 6) % NT) * 32 + ((e >> 1) & 31), ggt = (e >> 6) / NT;   // granule e: [gate tile][nt][trial][half]
                bf16_t *dst = nullptr;
                const int at = row * LDB + 8 * ggt + 4 * ghh;
                const unsigned w0 = pg[i][0], w1 = pg[i][1];
                *reinterpret_cast<u32x2 *>(WA + at) = u32x2{w0 & ~TAGBITS, w1 & ~TAGBITS};
                if (masked) {
                    // h0 (x) m: bit 14 / 30 -> a 16-bit mask per value (the scale keep is in the weights, see the prologue)
                    *reinterpret_cast<u32x2 *>(WB + at) = u32x2{w0 & ~TAGBITS & (((w0 >> 14) & 0x00010001u) * 0xffffu), w1 & ~TAGBITS & (((w1 >> 14) & 0x00010001u) * 0xffffu)};
                }
                *reinterpret_cast<u32x2 *>(WC + at) = u32x2{pg[i][2] & ~TAGBITS, pg[i][3] & ~TAGBITS};
                (void)dst;
            }
            stp.mark(7);                                         // ds_writes done
            __syncthreads();
            if (s_abort) break;
            stp.mark(1);
        }
        f32x16 acc0[NT], acc1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[nt][r] = bias0[r]; acc1[nt][r] = bias1[r]; }
        // All four k-steps, unconditionally (zero weights beyond the channel count; at s == T the result feeds nothing), and settled
        // before any control flow: the accumulators of asm MFMAs are ordinary values to hipcc, and around a run-time branch it copies
        // them (AGPR -> VGPR) at once -- reading a result the matrix pipe has not written yet.
        mfma_lead_in();
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_acc_v(acc0[nt], wx0[k], xf[nt][k]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mfma_settle(acc0[nt]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < T) load_x(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (s >= 1) {
            // all three products every step, branch-free: at s == 1 the h1 tile is the zero state the buffers start with, at
            // s == T the layer-0 product feeds nothing (its cell is guarded by do0)
            const bf16x8 *const ws3[3] = {w0, wx, w1};
            const bf16_t *const ts3[3] = {TA, TB, TC};
            if (!(NSD_SCAN_ABLATE & 256)) mfma_pipe<NT, KS, LDB, 3, 12>(ws3, ts3, col, hh, [&](const int st, const int nt) -> f32x16 & { return st == 0 ? acc0[nt] : acc1[nt]; });
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
            // |h| <= 1 for finite arithmetic: bit 14 of a packed h is set only by Inf / NaN (see the ring's description above)
            if (((hw0[nt][0] | hw0[nt][1] | hw1[nt][0] | hw1[nt][1]) & TAGBITS) != 0u) poison[nt] = 0x7FC07FC0u;
        }
        stp.mark(3);
        // ---- publish h0_t0 (+ the keep / drop bits of its multiplied copy) and h1_t1 as ONE tagged granule per lane; no drain, no flag
        {
            const nsd_rsrc rw = make_rsrc((s & 1) ? ring1 : ring0, (unsigned)(XG * 2));
            const unsigned tag = ((s >> 1) & 1) ? TAGBITS : 0u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                unsigned k0 = 0u, k1 = 0u;
                if (masked) {
                    k0 = (mult[nt][0] != 0.f ? 0x4000u : 0u) | (mult[nt][1] != 0.f ? 0x40000000u : 0u);
                    k1 = (mult[nt][2] != 0.f ? 0x4000u : 0u) | (mult[nt][3] != 0.f ? 0x40000000u : 0u);
                }
                const bool pub0 = do0 && poison[nt] == 0u, pub1 = do1 && poison[nt] == 0u;
                const u32x4 gr = {pub0 ? (hw0[nt][0] | k0) : 0u, pub0 ? (hw0[nt][1] | k1) : 0u, (pub1 ? hw1[nt][0] : 0u) | tag, (pub1 ? hw1[nt][1] : 0u) | tag};
                st_ring_b128(same_l2, rw, (unsigned)((gran_off + 512 * nt) * 2), gr);
            }
        }
        // ... and the first look at the group's granules of this step.  Vector memory completes in issue order, so the look goes in
        // FRONT of most of the step's saves; the other members publish at about the same time and a store needs ~300 cycles to be
        // visible in the L2, so a few of the saves go first (LOOK_POS) -- a look that comes too early costs a second round trip.
        auto first_look = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            if (NSD_LOOK_DELAY) __builtin_amdgcn_s_sleep(NSD_LOOK_DELAY);
            else if (!train) __builtin_amdgcn_s_sleep(NSD_LOOK_DELAY_INFER);   // (inference: nothing separates the publish from the look -- see NSD_LOOK_DELAY_INFER)
            if (s < T) {
                const nsd_rsrc rn = make_rsrc((s & 1) ? ring1 : ring0, (unsigned)(XG * 2));
#pragma unroll
                for (int i = 0; i < PIECES; ++i) pg[i] = ld_sc1_b128(rn, (unsigned)((tid + 256 * i) * 16));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (NSD_LOOK_POS == 0 || !train) first_look();
        stp.mark(4);
        // ---- row-major copies (the head reads hs1; the weight-gradient GEMMs read hs0 / lk0 / hs1) and the saves for the
        // backward pass: nobody waits for them inside this launch
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned lane_off = (unsigned)(col * H + u0);
            const long row0 = (((long)(b0 >> 5) + nt) * T + t0) * 32, row1 = (((long)(b0 >> 5) + nt) * T + t1) * 32;    // seq_row(t, b) = row + col
            if (do0 && train) {
                st_stream<u32x2>(a.hs0 + row0 * H + lane_off, u32x2{hw0[nt][0] | poison[nt], hw0[nt][1] | poison[nt]});
                if (masked) st_stream<u32x2>(a.lk0 + row0 * H + lane_off, u32x2{lw0[nt][0] | poison[nt], lw0[nt][1] | poison[nt]});
            }
            if (do1) st_stream<u32x2>(a.hs1 + row1 * H + lane_off, u32x2{hw1[nt][0] | poison[nt], hw1[nt][1] | poison[nt]});
        }
        if (train) {
            if (NSD_LOOK_POS == 1) first_look();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (do0) {
                    const long blk = saved_block((b0 >> 5) + nt, P, me.p, T, t0, wave_s);
                    st_stream<u32x2>(a.cs0 + blk * 256 + lane * 4, u32x2{pack_bf16x2(c0[nt][0], c0[nt][1]), pack_bf16x2(c0[nt][2], c0[nt][3])});
                    bf16_t *gd = a.ga0 + blk * 1024 + lane * 8;
                    // (sign of the saved i = the unit's output survived the dropout between the layers: saved_keep_bits)
                    st_stream<u32x4>(gd, u32x4{pack_bf16x2(g0[nt][0][0], g0[nt][0][1]) | (mult[nt][0] != 0.f ? 0x8000u : 0u), pack_bf16x2(g0[nt][0][2], g0[nt][0][3]),
                                                pack_bf16x2(g0[nt][1][0], g0[nt][1][1]) | (mult[nt][1] != 0.f ? 0x8000u : 0u), pack_bf16x2(g0[nt][1][2], g0[nt][1][3])});
                    st_stream<u32x4>(gd + 512, u32x4{pack_bf16x2(g0[nt][2][0], g0[nt][2][1]) | (mult[nt][2] != 0.f ? 0x8000u : 0u), pack_bf16x2(g0[nt][2][2], g0[nt][2][3]),
                                                      pack_bf16x2(g0[nt][3][0], g0[nt][3][1]) | (mult[nt][3] != 0.f ? 0x8000u : 0u), pack_bf16x2(g0[nt][3][2], g0[nt][3][3])});
                }
                if (NSD_LOOK_POS == 2 && nt == NT - 1) first_look();
                if (do1) {
                    const long blk = saved_block((b0 >> 5) + nt, P, me.p, T, t1, wave_s);
                    st_stream<u32x2>(a.cs1 + blk * 256 + lane * 4, u32x2{pack_bf16x2(c1[nt][0], c1[nt][1]), pack_bf16x2(c1[nt][2], c1[nt][3])});
                    bf16_t *gd = a.ga1 + blk * 1024 + lane * 8;
                    st_stream<u32x4>(gd, u32x4{pack_bf16x2(g1[nt][0][0], g1[nt][0][1]), pack_bf16x2(g1[nt][0][2], g1[nt][0][3]),
                                                pack_bf16x2(g1[nt][1][0], g1[nt][1][1]), pack_bf16x2(g1[nt][1][2], g1[nt][1][3])});
                    st_stream<u32x4>(gd + 512, u32x4{pack_bf16x2(g1[nt][2][0], g1[nt][2][1]), pack_bf16x2(g1[nt][2][2], g1[nt][2][3]),
                                                      pack_bf16x2(g1[nt][3][0], g1[nt][3][1]), pack_bf16x2(g1[nt][3][2], g1[nt][3][3])});
                }
            }
            if (NSD_LOOK_POS == 3) first_look();
        }
        stp.mark(5);
    }
    {
        bool any = false;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) any = any || poison[nt] != 0u;
        if (__any(any) && lane == 0) atomicOr(a.status, NSD_SEQ_ST_NONFINITE);
    }
    stp.store(a.status, blockIdx.x == 0 && tid == 0);
}

// ---- backward -------------------------------------------------------------------------------------------------------------
// The recurrent terms are exchanged as a REDUCE-SCATTER of partial sums, not as an all-gather of da.  A workgroup owns the gate
// columns of its 32 units (128 columns of da1 and of da0) and holds the rows of W_hh1^T, W_ih1^T, W_hh0^T for those columns and
// ALL H units.  Per step it multiplies its own fresh da (K = 128, straight from LDS) into partial sums of dh for every unit
// and sends each member of the group the 32 rows that member owns, as bf16; a member adds the P partials of its rows in fp32,
// in member order.  Against the all-gather (every member reading the whole 2 x MG x 4H da tile: 128 KB per step at H = 256,
// 64 B/clk from the L2 -> 2 000+ cycles, then a K-split reduction through 96 KB of LDS and a second barrier) a member now
// reads 48 KB, the K reduction ends inside one wave, and nothing of the cell waits for LDS.  Rounding the partial sums to bf16
// costs about what rounding da to bf16 already costs (both ~2^-9 relative on terms of the same sum); parity tests unchanged.
//
// The ring of a group has ONE slot, and it lives in the XCD's L2: the L2 is write-back (tools/micro/l2_writeback.hip), a group on one
// XCD stores with plain stores, and a slot that is rewritten every step stays there as dirty lines as long as it fits -- 1.5 MB per
// XCD at cfg3, where two parity slots (3 MB of the 4-MB L2, beside the step's streaming bytes) were written back every step:
// 3 GB of the launch's 8 GB of fabric traffic, and as much again in reads that missed.  One slot needs a second handshake: a wave
// counts what it has CONSUMED (consume counter s = "I hold the sums of step s-1"), and a producer looks at the group's consume
// counters before its first ring store of step s.  The counters are requested right after the step barrier and were written
// ~2 000 cycles earlier, so the look is a register compare; the bounded poll behind it is the rare path.
// ring slot of a group:  R16 [consumer member][producer member][nt][consumer wave 4][lane 64] x 16 B =
//   {rec1 = W_hh1^T da1 (4 bf16) | din0 = W_ih1^T da1 (4 bf16)} of the lane's 4 units, R8 the same index x 8 B = rec0 = W_hh0^T da0.
//   A producer wave's accumulator registers 4q..4q+3 of lane (trial, hh) ARE consumer wave q's lane (trial, hh) units, so
//   every store / load instruction moves one contiguous 1-KB / 512-B block (whole lines, one producer each).
template <int P, int NT>
struct PartRing {
    static constexpr long NBLK = (long)P * P * NT * 4, SLOT_BYTES = NBLK * (1024 + 512);
    __device__ static __forceinline__ unsigned off16(const int cons, const int prod, const int nt, const int q) {
        return (unsigned)(((((cons * P + prod) * NT + nt) * 4 + q)) << 10);
    }
    __device__ static __forceinline__ unsigned off8(const int cons, const int prod, const int nt, const int q) {
        return (unsigned)(NBLK << 10) + (unsigned)(((((cons * P + prod) * NT + nt) * 4 + q)) << 9);
    }
};

template <int H, int NT>
__global__ __launch_bounds__(256) void scan2_bwd_kernel(const Scan2BwdArgs a) {
    constexpr int P = H / 32, MG = 32 * NT, G = 4 * H, RT = P >= 4 ? P / 4 : 1, KS = 8;
    using Ring = PartRing<P, NT>;
    // the workgroup's own da of the step, as MFMA B operands: k-step 2*wave + half = the 1-KB lane-linear block producer wave
    // `wave` writes with one ds_write_b128 (lane (trial, hh): columns 16hh + 8half + 0..7 of the wave's 32)
    __shared__ __align__(16) bf16_t dab[2][2][KS][NT][512];      // [step parity][layer][k-step][nt][lane * 8]
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Member me = member_of(blockIdx.x, a.groups, P, a.spread_groups);
    const int b0 = (a.group0 + me.group) * MG;
    const int col = lane & 31, hh = lane >> 5;
    const bool has_rows = wave < P;                              // (P < 4: only the first P waves own a consumer's row tile)

    // rows of W^T for the consumers r = wave + 4 ri (their 32 units), columns = this workgroup's 128 gate columns in k-step order
    bf16x8 wq1[RT][KS], wqx[RT][KS], wq0[RT][KS];
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
        const int r = (wave + 4 * ri) < P ? wave + 4 * ri : 0;
        const long ro = (long)(32 * r + col) * G + 128 * me.p + 16 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int co = 32 * (ks >> 1) + 8 * (ks & 1);
            wq1[ri][ks] = *reinterpret_cast<const bf16x8 *>(a.wb1 + ro + co);
            wqx[ri][ks] = *reinterpret_cast<const bf16x8 *>(a.wxt1 + ro + co);
            wq0[ri][ks] = *reinterpret_cast<const bf16x8 *>(a.wb0 + ro + co);
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
    if (rv < 0 && lane == 0) { s_abort = 1; report_timeout(a.status, ST2_BWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);
    const int T = a.T;
    const int u0 = 32 * me.p + 8 * wave + 4 * hh;
    float dpl[NT][4], aw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aw[j] = a.attn_w[u0 + j];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dpl[nt][j] = a.dpooled[(long)(b0 + 32 * nt + col) * H + u0 + j];
    const char *ring0 = reinterpret_cast<const char *>(a.xch) + (long)(a.group0 + me.group) * Ring::SLOT_BYTES;
    unsigned *gacks = gflags + ACK_WORD;                      // consume counters of the group: the ring has ONE slot (see "backward" above)

    // Saved activations / upstream terms of a step (HBM reads, independent of the recurrence).  Vector memory returns in issue
    // order: an HBM read issued just before the flag poll or before the publishing drain puts its whole latency on the step.
    // The set of step s+1 is requested right after the barrier of step s: the MFMA phase (~1 800 cycles) follows.
    struct Saved {
        u32x4 q1[NT][2], q0[NT][2];
        u32x2 cq1[NT], cp1[NT], cq0[NT], cp0[NT];
        float al[NT], ds[NT];
    };
    // (inside the step loop the requests are unconditional -- clamped addresses, values of an inactive cell are never used and the
    // "previous c" of t = 0 is zeroed by a select: the compiler's vmcnt for the consume-counter load issued just before them is then
    // exact.  With conditional requests it assumes the shortest path and the look waits for most of these HBM reads.)
    // (addresses: a wave-uniform 64-bit base per step -- scalar arithmetic, SGPR base operand -- plus a 32-bit lane offset that never
    // changes; formed per lane in 64 bits they were ~120 VALU instructions of a step that is bound by instruction issue)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // (the cell state is read ONCE: c_t of a step is the c_{t-1} the step before it used -- carried in registers; only the very first
    // set loads a c_t.  Layer 0 is idle at step 0, and the "c_{t-1}" it loads there is c0[T-1], exactly step 1's c_t.)
    auto load_saved = [&](const int sx, Saved &v, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int x1 = T - 1 - sx > 0 ? T - 1 - sx : 0, x0 = T - sx < T ? T - sx : T - 1;     // (sx = T: layer 1 is idle; sx = 0: layer 0 is)
        const int xp1 = T - 2 - sx > 0 ? T - 2 - sx : 0, xp0 = T - 1 - sx > 0 ? T - 1 - sx : 0;   // c_{t-1}: t1 - 1, t0 - 1 (zeroed by fix_saved where t = 0)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            {
                const long row_u = (((long)(b0 >> 5) + nt) * T + x1) * 32;
                const long blk = saved_block((b0 >> 5) + nt, P, me.p, T, x1, wave_s);
                const bf16_t *gs = a.ga1 + blk * 1024, *cp = a.cs1 + saved_block((b0 >> 5) + nt, P, me.p, T, xp1, wave_s) * 256;
                v.q1[nt][0] = ld_stream<u32x4>(gs + lane * 8); v.q1[nt][1] = ld_stream<u32x4>(gs + 512 + lane * 8);
                if constexpr (FIRST) v.cq1[nt] = ld_stream<u32x2>(a.cs1 + blk * 256 + lane * 4); else v.cq1[nt] = v.cp1[nt];
                v.cp1[nt] = ld_stream<u32x2>(cp + lane * 4);
                v.al[nt] = ld_stream<float>(a.alpha + row_u + col); v.ds[nt] = ld_stream<float>(a.dscore + row_u + col);
            }
            {
                const long blk = saved_block((b0 >> 5) + nt, P, me.p, T, x0, wave_s);
                const bf16_t *gs = a.ga0 + blk * 1024, *cp = a.cs0 + saved_block((b0 >> 5) + nt, P, me.p, T, xp0, wave_s) * 256;
                v.q0[nt][0] = ld_stream<u32x4>(gs + lane * 8); v.q0[nt][1] = ld_stream<u32x4>(gs + 512 + lane * 8);
                if constexpr (FIRST) v.cq0[nt] = u32x2{0u, 0u}; else v.cq0[nt] = v.cp0[nt];
                v.cp0[nt] = ld_stream<u32x2>(cp + lane * 4);
            }
        }
    };
    // c_{t-1} of t = 0 is the zero state (applied where the set is used)
    auto fix_saved = [&](const int sx, Saved &v) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (T - 1 - sx <= 0) v.cp1[nt] = u32x2{0u, 0u};
            if (T - sx <= 0) v.cp0[nt] = u32x2{0u, 0u};
        }
    };
    Saved sv;
    load_saved(0, sv, std::true_type{});
    const float keep0 = a.rng.on ? a.rng.keep_lstm : 1.f;
    Stamps stp;
    stp.start();
    unsigned long long pass_acc[6] = {0, 0, 0, 0, 0, 0}, pass_last = 0;
    for (int s = 0; s <= T; ++s) {
        const bool do1 = s < T, do0 = s >= 1;
        const int t1 = T - 1 - s, t0 = T - s;
        // ---- ahead of the exchange: everything of the two cells that does not need dh
        CellFac f1[NT], f0[NT];
        fix_saved(s, sv);
        float dup1[NT][4], m0[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            cell_factors(sv.q1[nt][0], sv.q1[nt][1], sv.cq1[nt], sv.cp1[nt], f1[nt]);
            cell_factors(sv.q0[nt][0], sv.q0[nt][1], sv.cq0[nt], sv.cp0[nt], f0[nt]);
#pragma unroll
            for (int j = 0; j < 4; ++j) dup1[nt][j] = do1 ? fmaf(sv.al[nt], dpl[nt][j], sv.ds[nt] * aw[j]) : 0.f;
            saved_keep_bits(sv.q0[nt][0], sv.q0[nt][1], keep0, m0[nt]);      // layer 0's dropout multipliers at t0
            pin(f1[nt]); pin(f0[nt]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { pin(dup1[nt][j]); pin(m0[nt][j]); }
        }
        stp.mark(7);
        float drec1[NT][4], dinx[NT][4], drec0[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { drec1[nt][j] = 0.f; dinx[nt][j] = 0.f; drec0[nt][j] = 0.f; }
        if (s >= 1) {
            if (!(NSD_SCAN_ABLATE & 1) && !wait_group<4 * P>(gflags, (unsigned)s, lane) && lane == 0) {
                s_abort = 1;
                report_timeout(a.status, ST2_BWD_TIMEOUT);
            }
            stp.mark(0);
            // the partial sums the P members sent this wave at step s-1, added in member order
            const nsd_rsrc rr = make_rsrc(ring0, (unsigned)Ring::SLOT_BYTES);
            u32x4 v16[P][NT];
            u32x2 v8[P][NT];
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    v16[q][nt] = ld_sc1_b128(rr, Ring::off16(me.p, q, nt, wave) + 16u * lane);
                    v8[q][nt] = ld_sc1_b64(rr, Ring::off8(me.p, q, nt, wave) + 8u * lane);
                }
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc_bf16x2(drec1[nt][0], drec1[nt][1], v16[q][nt][0]); acc_bf16x2(drec1[nt][2], drec1[nt][3], v16[q][nt][1]);
                    acc_bf16x2(dinx[nt][0], dinx[nt][1], v16[q][nt][2]); acc_bf16x2(dinx[nt][2], dinx[nt][3], v16[q][nt][3]);
                    acc_bf16x2(drec0[nt][0], drec0[nt][1], v8[q][nt][0]); acc_bf16x2(drec0[nt][2], drec0[nt][3], v8[q][nt][1]);
                }
            // this wave has taken its partial sums of step s-1 out of the ring: the producers may rewrite the slot
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_xchg_u32(same_l2, gacks + 4 * me.p + wave, (unsigned)s);
            stp.mark(1);
        }
        // ---- the dh-dependent rest of both cells: da1_{t1}, da0_{t0}
        unsigned dw1[NT][8], dw0[NT][8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (do1) {
                float dh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dh[j] = dup1[nt][j] + drec1[nt][j];
                cell_apply(f1[nt], dh, dc1[nt], dbs1, dw1[nt]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) dw1[nt][j] = 0u;
            }
            if (do0) {
                float dh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dh[j] = fmaf(dinx[nt][j], m0[nt][j], drec0[nt][j]);
                cell_apply(f0[nt], dh, dc0[nt], dbs0, dw0[nt]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) dw0[nt][j] = 0u;    // step 0: da0_{T} = 0 (its products are published as zeros)
            }
        }
        stp.mark(2);
        // row-major da for the weight-gradient GEMMs: nobody waits for it inside this launch.  Issued right behind the step barrier,
        // ~3 000 cycles ahead of the drain that precedes the flag (the registers are free during the MFMA stream)
        auto store_da_rows = [&]() {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned lane_off = (unsigned)(col * G + 4 * u0);             // (the trial's row inside the tile's 32, the lane's 16 gate columns)
                if (do1) {
                    bf16_t *d = a.da1 + (((long)(b0 >> 5) + nt) * T + t1) * 32 * G;
                    st_stream<u32x4>(d + lane_off, u32x4{dw1[nt][0], dw1[nt][1], dw1[nt][2], dw1[nt][3]});
                    st_stream<u32x4>(d + lane_off + 8, u32x4{dw1[nt][4], dw1[nt][5], dw1[nt][6], dw1[nt][7]});
                }
                if (do0) {
                    bf16_t *d = a.da0 + (((long)(b0 >> 5) + nt) * T + t0) * 32 * G;
                    st_stream<u32x4>(d + lane_off, u32x4{dw0[nt][0], dw0[nt][1], dw0[nt][2], dw0[nt][3]});
                    st_stream<u32x4>(d + lane_off + 8, u32x4{dw0[nt][4], dw0[nt][5], dw0[nt][6], dw0[nt][7]});
                }
            }
        };
        if (s < T) {                                            // (after the last step nobody reads a partial sum)
            const int par = s & 1;
            if (NSD_SCAN_ABLATE & 64) load_saved(s + 1, sv, std::false_type{});
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    *reinterpret_cast<u32x4 *>(&dab[par][1][2 * wave + h][nt][lane * 8]) = u32x4{dw1[nt][4 * h], dw1[nt][4 * h + 1], dw1[nt][4 * h + 2], dw1[nt][4 * h + 3]};
                    *reinterpret_cast<u32x4 *>(&dab[par][0][2 * wave + h][nt][lane * 8]) = u32x4{dw0[nt][4 * h], dw0[nt][4 * h + 1], dw0[nt][4 * h + 2], dw0[nt][4 * h + 3]};
                }
            __syncthreads();
            if (s_abort) break;
            stp.mark(3);
            // the consume counters of the group, requested now and looked at before the first ring store of the step
            unsigned ackv = ld_sc1_u32(gacks + (lane < 4 * P ? lane : 0));     // (every lane loads: no exec-masked block for the compare to be pulled into)
            __builtin_amdgcn_sched_barrier(0);                  // FIRST in the queue: its wait must not include the HBM-bound requests below
            store_da_rows();
            if (!(NSD_SCAN_ABLATE & (8 | 16 | 32 | 64))) load_saved(s + 1, sv, std::false_type{});   // (its factors were taken at the top of the step: the registers are free)
            __builtin_amdgcn_sched_barrier(0);
            // ---- partial sums of dh for every unit of the group from this workgroup's 128 + 128 columns: 3 RT passes of KS MFMAs
            // (consumer row tile r = wave + 4 ri: W_hh1^T da1, W_ih1^T da1, W_hh0^T da0), ONE instruction stream in which the wave's
            // other work rides in the gaps -- a 32x32x16 MFMA occupies the matrix pipe for 64 cycles and the wave for ~8.  Behind
            // MFMA k of a pass the wave converts a quarter of the PREVIOUS pass's accumulator and sends it (that pass's last MFMA
            // left the pipe when this pass's second one was issued); 
            // No run-time branch inside the stream except the consume-counter look, which sits behind a settle (hipcc copies
            // asm-MFMA accumulators around branches without knowing they may still be in flight).
            auto stream = [&](auto same_c) {                    // (one copy per exchange mode: the ring stores' cache bits are immediates)
                constexpr bool SAME = decltype(same_c)::value;
                const nsd_rsrc rw = make_rsrc(ring0, (unsigned)Ring::SLOT_BYTES);
                constexpr int NF = KS * NT, NPASS = 3 * RT, NTOT = NPASS * NF, D = 3;
                f32x16 aR1[NT], aX0[NT], aR0[NT];
                unsigned pk1[NT][4][2];
                auto frag_of = [&](const int i) {
                    const int ps = i / NF, kk = i % NF;
                    return *reinterpret_cast<const bf16x8 *>(&dab[par][ps % 3 == 2 ? 0 : 1][kk / NT][kk % NT][lane * 8]);
                };
                auto conv = [&](auto psc, auto qc) {             // quarter q (= consumer wave q's units) of the accumulators of pass ps
                    constexpr int ps = decltype(psc)::value, q = decltype(qc)::value, m = ps % 3;
                    const int r = wave + 4 * (ps / 3);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (m == 0) {
                            pk1[nt][q][0] = pack_bf16x2(aR1[nt][4 * q], aR1[nt][4 * q + 1]); pk1[nt][q][1] = pack_bf16x2(aR1[nt][4 * q + 2], aR1[nt][4 * q + 3]);
                        } else if constexpr (m == 1) {
                            const u32x4 o16 = {pk1[nt][q][0], pk1[nt][q][1], pack_bf16x2(aX0[nt][4 * q], aX0[nt][4 * q + 1]), pack_bf16x2(aX0[nt][4 * q + 2], aX0[nt][4 * q + 3])};
                            st_ring_b128(SAME, rw, Ring::off16(r, me.p, nt, q) + 16u * lane, o16);
                        } else {
                            const u32x2 o8 = {pack_bf16x2(aR0[nt][4 * q], aR0[nt][4 * q + 1]), pack_bf16x2(aR0[nt][4 * q + 2], aR0[nt][4 * q + 3])};
                            st_ring_b64(SAME, rw, Ring::off8(r, me.p, nt, q) + 8u * lane, o8);
                        }
                    }
                };
                bf16x8 f[D];
#pragma unroll
                for (int i = 0; i < D; ++i) f[i] = frag_of(i);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, NTOT>([&](auto ic) {               // (compile-time indices by construction: a rolled loop would select weights and accumulators at run time)
                    constexpr int i = decltype(ic)::value;
                    constexpr int ps = i / NF, kk = i % NF, ks = kk / NT, nt = kk % NT, ri = ps / 3, m = ps % 3;
                    if constexpr (ps == 2 && kk == 0) {
                        // the first ring store of the step follows: every wave of the group must have taken step s-1 out of the slot
#pragma unroll
                        for (int n2 = 0; n2 < NT; ++n2) mfma_settle(aR1[n2], aX0[n2]);
                        pin(ackv);                                       // the compare stays HERE: at the load it would expose an L2 round trip per step
                        if (!__all(ackv >= (unsigned)s)) {               // (rare: the counters were read ~1 500 cycles after they were written)
                            if (!wait_group<4 * P>(gacks, (unsigned)s, lane) && lane == 0) { s_abort = 1; report_timeout(a.status, ST2_BWD_TIMEOUT); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (m == 0) { if constexpr (ks == 0) mfma_new_a(aR1[nt], wq1[ri][ks], f[i % D]); else mfma_acc_a(aR1[nt], wq1[ri][ks], f[i % D]); }
                    else if constexpr (m == 1) { if constexpr (ks == 0) mfma_new_a(aX0[nt], wqx[ri][ks], f[i % D]); else mfma_acc_a(aX0[nt], wqx[ri][ks], f[i % D]); }
                    else { if constexpr (ks == 0) mfma_new_a(aR0[nt], wq0[ri][ks], f[i % D]); else mfma_acc_a(aR0[nt], wq0[ri][ks], f[i % D]); }
                    if constexpr (i + D < NTOT) f[i % D] = frag_of(i + D);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ps >= 1 && kk >= 1 && kk <= 4) {
                        if constexpr (kk == 1) {
#pragma unroll
                            for (int n2 = 0; n2 < NT; ++n2) {
                                if constexpr ((ps - 1) % 3 == 0) mfma_fence(aR1[n2]); else if constexpr ((ps - 1) % 3 == 1) mfma_fence(aX0[n2]); else mfma_fence(aR0[n2]);
                            }
                        }
                        conv(std::integral_constant<int, ps - 1>{}, std::integral_constant<int, kk - 1>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (NSD_SCAN_STAMPS != 0 && kk == NF - 1) {       // (diagnostic build: cycles of each pass, behind the stamps proper)
                        const unsigned long long t_ = __builtin_amdgcn_s_memtime();
                        pass_acc[ps < 6 ? ps : 5] += t_ - pass_last; pass_last = t_;
                    }
                    if constexpr (NSD_SCAN_STAMPS != 0 && i == 0) pass_last = __builtin_amdgcn_s_memtime();
                });
#pragma unroll
                for (int n2 = 0; n2 < NT; ++n2) mfma_settle(aR0[n2]);
                static_for<0, 4>([&](auto qc) { conv(std::integral_constant<int, NPASS - 1>{}, qc); });
                __builtin_amdgcn_sched_barrier(0);
            };
            if (has_rows) {
                if (same_l2) stream(std::true_type{}); else stream(std::false_type{});
            }
            stp.mark(4);
            if (NSD_SCAN_ABLATE & 32) load_saved(s + 1, sv, std::false_type{});
            stp.mark(5);
            if (!(NSD_SCAN_ABLATE & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_xchg_u32(same_l2, gflags + 4 * me.p + wave, (unsigned)(s + 1));
            stp.mark(6);
        }
        if (s == T) store_da_rows();
    }
    stp.store(a.status, blockIdx.x == 0 && tid == 0);
    if (NSD_SCAN_STAMPS && blockIdx.x == 0 && tid == 0) for (int i = 0; i < 6; ++i) reinterpret_cast<unsigned long long *>(a.status + 20)[i] = pass_acc[i];
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

// LDS of the forward kernel: 2 x 3 tiles of MG x (H + 8) bf16; of the backward kernel: 32 KB x NT (the workgroup's own da as B operands)
bool nsd_scan2_supported(int H, int MG) {
    if (!(H == 64 || H == 128 || H == 256)) return false;
    if (MG != 32) return false;                                  // only the 32-trial instantiations are built: larger batches take the layer-by-layer scans
    const long fwd = 2L * 3 * MG * (H + 8) * 2, bwd = 2L * 2 * 8 * (MG / 32) * 1024;
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
    const dim3 grid(a.groups * (H / 32) - a.diag_short_grid);
    switch (H) {
    case 64: return launch2_fwd<64>(a, MG, grid, st);
    case 128: return launch2_fwd<128>(a, MG, grid, st);
    default: return launch2_fwd<256>(a, MG, grid, st);
    }
}
int nsd_scan2_bwd_launch(const Scan2BwdArgs &a, int H, int MG, hipStream_t st) {
    if (!nsd_scan2_supported(H, MG) || a.groups * (H / 32) > nsd_num_cus()) { nsd_set_error("scan2_bwd: unsupported geometry H=%d MG=%d groups=%d", H, MG, a.groups); return NSD_E_INVALID; }
    const dim3 grid(a.groups * (H / 32) - a.diag_short_grid);
    switch (H) {
    case 64: return launch2_bwd<64>(a, MG, grid, st);
    case 128: return launch2_bwd<128>(a, MG, grid, st);
    default: return launch2_bwd<256>(a, MG, grid, st);
    }
}
