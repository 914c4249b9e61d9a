// nsd_scan.hip -- the serial part of the sequence-batched LSTM path: persistent recurrence kernels with the recurrent
// weights resident in registers for the whole sequence (self.lstm(x), Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34:
// torch.nn.LSTM semantics -- gate order i,f,g,o, zero initial state, one direction per scan -- and BPTT through it).
//
// Decomposition.  The input projection of a layer is hoisted out of the recurrence (one GEMM over the whole sequence,
// nsd_gemm_bf16.hip); what is left per step is gates[4H, trials] = W_hh[4H, H] . h_{t-1}[H, trials] + xproj_t.  A GROUP of
// P = H/32 workgroups advances one batch tile of MG trials through all T steps; workgroup p owns 32 hidden units: its 4 waves
// hold the 4 x 32 matching rows of W_hh (bf16, 16-byte A fragments of v_mfma_f32_32x32x16_bf16) in VGPRs for the whole
// launch -- 16 KB per wave at H = 256, 32 KB at H = 512 -- so the weights never move again.  The gate rows of a wave's tile
// are ordered so that accumulator registers 4j..4j+3 of a lane are the gates i,f,g,o of ONE unit for ONE trial (the lane's
// column): the LSTM cell runs entirely in the lane's registers, the cell state never leaves them, and the 4 units a lane
// owns are adjacent in memory (8-byte stores).
//
// Exchange.  Each step every workgroup needs h_{t-1} of ALL H units of its batch tile: the members of a group exchange their
// 32-unit slices through a small ring (two slots, step parity) in which every producer wave owns whole 128-byte lines
// (nsd_scan_common.h); hs[t] itself is written row-major for the GEMMs and the head.
// Forward scans (round 3): the ring validates itself.  |h| < 1, so bit 14 of every bf16 h is free and carries the TAG of the step that
// wrote it ((s >> 1) & 1: it flips every time a parity slot is rewritten); a consuming wave loads its 16-byte pieces until every
// value carries the expected tag and strips the bits on the way into LDS -- no drain of the stores, no flag, no poll: one L2 round
// trip behind the slowest producer instead of three.  Both slots start with the tag their first writer will not use.
// Backward scans: the storing wave drains vmcnt and publishes the step count in ITS OWN flag word (4 words per workgroup, no
// workgroup barrier on the publishing side); every consuming wave polls all 4P words of its group with sc1 loads (one word per
// lane) and only then issues its sc1 loads (guide: Guideline 16 R1 with sc1 loads in place of the acquire fence; every handed-off
// byte is stored and loaded sc1 in the write-through mode, every flag follows the drain of the stores it stands for); the
// partial-sum ring has ONE slot, guarded by consume counters (nsd_scan2.hip, "backward").
// Same-XCD shortcut: before the first step the members of a group exchange their XCC ids (HW_REG_XCC_ID) through the slow
// protocol -- which doubles as a start barrier: nobody enters the time loop before every member is resident.  When all ids
// are equal the group shares ONE L2, and the exchange switches to plain stores (kept in that L2; vmcnt is acknowledged by
// the L2) read by the same L1-bypassing loads: an L2 round trip per hop instead of a memory round trip.  Which mode a
// group runs in is decided from what the hardware reports at run time, identically by all its members; a group spread
// over XCDs simply keeps the write-through protocol.  Results are the same in both modes.  Groups are independent of each
// other: no grid-wide barrier exists.  Every spin is bounded: a member that does not see its group for ~1 s sets the status
// word and leaves (the others of the group then time out the same way).
// Layer 0 with at most 64 channels computes its input projection in the scan (INPROJ: 4 MFMAs per step from xbf).
//
// Backward.  Same grouping and ownership; the recurrent term is a REDUCE-SCATTER of bf16 partial sums: a workgroup multiplies its
// own 128 gate columns of da_t (from LDS) into partial dh rows for every unit of the layer and sends each member the 32 rows
// it owns; a member adds the P partials in fp32 (design, ring layout and measurements: nsd_scan2.hip and DESIGN.md 4.3b).
#include "nsd_scan_common.h"

namespace {


// ---------------------------------------------------------------------------------------------------------------------------
// operand preparation
// ---------------------------------------------------------------------------------------------------------------------------
// Row order of a 32-row accumulator tile: row n = 8j + 4hh + g  <->  unit 4hh + j of the tile's 8 units, gate g
__device__ __forceinline__ int tile_row_to_param_row(const int tile, const int n, const int H) {
    const int j = n >> 3, hh = (n >> 2) & 1, g = n & 3;
    return g * H + 8 * tile + 4 * hh + j;
}

__global__ __launch_bounds__(256) void seq_prep_kernel(const PrepArgs a) {
    const int H = a.H, G = 4 * H, I = a.I, Ip = a.Ipad;
    const long n_wf = (long)G * H, n_wb = (long)H * G, n_wx = (long)G * Ip, n_wxt = a.wxt ? (long)I * G : 0, n_b = G;
    const long total = n_wf + n_wb + n_wx + n_wxt + n_b;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        long i = e;
        if (i < n_wf) {                                        // wf[row'][k] = W_hh[row(row')][k]
            const int rp = (int)(i / H), k = (int)(i - (long)rp * H);
            a.wf[i] = (bf16_t)a.w_hh[(long)tile_row_to_param_row(rp >> 5, rp & 31, H) * H + k];
            continue;
        }
        i -= n_wf;
        if (i < n_wb) {                                        // wb[j][c = 4u + g] = W_hh[g*H + u][j]
            const int j = (int)(i / G), c = (int)(i - (long)j * G), u = c >> 2, g = c & 3;
            a.wb[i] = (bf16_t)a.w_hh[(long)(g * H + u) * H + j];
            continue;
        }
        i -= n_wb;
        if (i < n_wx) {                                        // wx[row'][i] = W_ih[row(row')][i], zero padded to Ipad
            const int rp = (int)(i / Ip), c = (int)(i - (long)rp * Ip);
            a.wx[i] = c < I ? (bf16_t)a.w_ih[(long)tile_row_to_param_row(rp >> 5, rp & 31, H) * I + c] : (bf16_t)0.f;
            continue;
        }
        i -= n_wx;
        if (i < n_wxt) {                                       // wxt[i][off + c] = W_ih[g*H + u][i]
            const int ii = (int)(i / G), c = (int)(i - (long)ii * G), u = c >> 2, g = c & 3;
            a.wxt[(long)ii * a.wxt_ld + a.wxt_off + c] = (bf16_t)a.w_ih[(long)(g * H + u) * I + ii];
            continue;
        }
        i -= n_wxt;
        {
            const int rp = (int)i, row = tile_row_to_param_row(rp >> 5, rp & 31, H);
            a.bsum[rp] = a.b_ih[row] + a.b_hh[row];
        }
    }
}

// x [B][T][C] fp32 -> xbf [seq_row(t, b)][CP] bf16, zero padded
__global__ __launch_bounds__(256) void seq_xbf_kernel(const float *x, bf16_t *xbf, int B, int Bp, int T, int C, int CP) {
    const long total = (long)T * Bp * CP;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % CP);
        const long r = e / CP;
        const int b = (int)(r / ((long)T * 32)) * 32 + (int)(r & 31), t = (int)((r >> 5) % T);
        xbf[e] = (b < B && c < C) ? (bf16_t)x[((long)b * T + t) * C + c] : (bf16_t)0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward scan
// ---------------------------------------------------------------------------------------------------------------------------
// INPROJ: the input projection is computed in the scan from xbf (layer 0, CP <= 64) instead of read as accumulator tiles
template <int H, int NT, bool INPROJ>
__global__ __launch_bounds__(256) void scan_fwd_kernel(const ScanFwdArgs a) {
    constexpr int KS = H / 16, P = H / 32, MG = 32 * NT, LDB = H + 8, G = 4 * H;
    // h_{t-1} of the batch tile, [trial][unit]; two buffers alternate by step so that ONE barrier per step is enough (a wave
    // refills buffer s&1 only after it passed the barrier of step s-1, i.e. after every wave finished reading it at step s-2)
    __shared__ __align__(16) bf16_t Bt2[2][MG * LDB];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Member me = member_of(blockIdx.x, a.groups, P, a.spread_groups);
    const int dir = me.dir;
    const int gt = 4 * me.p + wave;                            // this wave's accumulator tile = units 8gt .. 8gt+7
    const int b0 = (a.group0 + me.group) * MG;
    const int col = lane & 31, hh = lane >> 5;                 // MFMA column (trial of the tile) / k half

    // recurrent weights -> registers for the whole sequence (A operand: lane holds row `col` of the tile, k = 16ks + 8hh ..)
    bf16x8 w[KS];
    {
        const bf16_t *wrow = a.wf[dir] + (long)(32 * gt + col) * H + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[ks] = *reinterpret_cast<const bf16x8 *>(wrow + 16 * ks);
    }
    float c[NT][4];
    unsigned poison[NT];                                        // non-finite h of this lane's units of trial (nt, col): see nsd_scan2.hip, forward
    bool wbad = false;                                          // (W_hh . h_{-1} is skipped at s = 0: a non-finite row is NaN at once in the reference)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wbad = wbad || frag_nonfinite(w[ks]);
    wbad = __any(wbad);                                         // (a lane holds a weight ROW; the row acts on every trial = every lane of the wave)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        poison[nt] = wbad ? 0x7FC07FC0u : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) c[nt][j] = 0.f;
    }
    if (tid == 0) s_abort = 0;
    // The exchange ring validates itself (design: nsd_scan2.hip, forward): |h| < 1, so bit 14 of every bf16 h is free and carries the
    // TAG of the step that wrote it ((s >> 1) & 1); a consumer loads its pieces until all tags are the expected ones -- no drain, no
    // flag, no poll.  Both slots start with the tag their first writer will not use, drained before the start barrier.
    constexpr unsigned TAGBITS = 0x40004000u;
    constexpr long XB0 = (long)MG * H;
    {
        const long slot_stride = (long)a.D * a.groups_total * XB0;
        bf16_t *r0 = a.xch + ((long)dir * a.groups_total + a.group0 + me.group) * XB0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long off = ring_h_off(gt, nt, NT, col, hh);
            st_sc1_u64(r0 + off, ((unsigned long long)TAGBITS << 32) | TAGBITS);
            st_sc1_u64(r0 + slot_stride + off, ((unsigned long long)TAGBITS << 32) | TAGBITS);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    unsigned *gflags = a.flags + (long)(dir * a.groups + me.group) * GROUP_WORDS;
    const int rv = group_rendezvous<P>(gflags, me.p, wave, lane);
    if (rv < 0 && lane == 0) { s_abort = 1; report_timeout(a.status, ST_FWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);     // diagnostics: groups on one XCD / spread over several
    const long ld = a.ld;
    constexpr long XB = (long)MG * H;                           // elements of a batch tile's block of the exchange ring
    auto ring_of = [&](const int step) { return a.xch + (((long)(step & 1) * a.D + dir) * a.groups_total + a.group0 + me.group) * XB; };
    const int u0 = 8 * gt + 4 * hh;                            // first of this lane's 4 units
    const bool train = a.cs[0] != nullptr;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave), gt_s = 4 * me.p + wave_s;     // (wave-uniform copies for address bases)

    // input projection: accumulator tiles from the hoisted GEMM, or (INPROJ) x_t of the tile as MFMA B fragments (lane (trial,
    // hh): channels 16k + 8hh .. + 7).  Either way the data of step s+1 is requested AFTER the tile gather of step s has landed
    // (vector memory returns in issue order: an HBM read issued ahead of the gather would put its latency on every step's
    // critical path) and after the data of step s has been consumed.
    u32x4 xp[NT][2];
    bf16x8 xf[NT][4], wx0[4];
    float bias0[16];
    const int CP = INPROJ ? a.CP : 16, ks0 = CP >> 4;
    auto load_xp = [&](const int t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // (addresses: wave-uniform 64-bit base + 32-bit lane offset, as in the backward scans)
            if constexpr (INPROJ) {
                const bf16_t *src = a.xbf + (((long)(b0 >> 5) + nt) * a.T + t) * 32 * CP;
                const unsigned lo = (unsigned)(col * CP + 8 * hh);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < ks0) xf[nt][k] = ld_stream<bf16x8>(src + lo + 16 * k);
            } else {
                const bf16_t *src = a.xproj[dir] + ((((long)((b0 >> 5) + nt) * a.T + t) * (G >> 5) + gt_s) * 64) * 16;
                xp[nt][0] = ld_stream<u32x4>(src + lane * 16);
                xp[nt][1] = ld_stream<u32x4>(src + lane * 16 + 8);
            }
        }
    };
    if constexpr (INPROJ) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                           // (k-steps beyond the padded channel count: zero weights)
            const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(a.wx0[dir] + (long)(32 * gt + col) * CP + 16 * (k < ks0 ? k : 0) + 8 * hh);
            wx0[k] = k < ks0 ? wv : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bias0[r] = a.bsum0[dir][32 * gt + mfma32_row(r, lane)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) xf[nt][k] = wx0[0];    // (defined values in the unused k-steps)
    }
    load_xp(dir == 0 ? 0 : a.T - 1);
    for (int s = 0; s < a.T; ++s) {
        const int t = dir == 0 ? s : a.T - 1 - s;
        const int tnx = dir == 0 ? t + 1 : t - 1;              // next step's time index
        f32x16 acc[NT];
        if (s > 0) {
            bf16_t *Bt = Bt2[s & 1];
            // gather h_{t-1} of the whole tile (all H units) from the exchange ring: the block of a batch tile is laid out [gate tile
            // = 4p + wave][nt][trial][8 units] -- every producer wave writes whole 128-byte lines with one store instruction, and a
            // consumer's 16-byte pieces are linear in the block (piece e = (unit group) * MG + trial).  The pieces are loaded until
            // every value carries the tag of step s-1 (bounded; a group that cannot complete reports and leaves).
            const nsd_rsrc rh = make_rsrc(ring_of(s - 1), (unsigned)(XB * 2));
            constexpr int PIECES = MG * (H / 8) / 256;
            const unsigned want = (((s - 1) >> 1) & 1) ? 0x4000u : 0u;
            u32x4 pv[PIECES];
            bool ok = false;
            for (unsigned spins = 0; spins < SPIN_LIMIT && !ok; ++spins) {
                bool mine = true;
#pragma unroll
                for (int i = 0; i < PIECES; ++i) pv[i] = ld_sc1_b128(rh, (unsigned)((tid + 256 * i) * 16));
#pragma unroll
                for (int i = 0; i < PIECES; ++i) mine = mine && ((pv[i][0] & 0x4000u) == want) && ((pv[i][2] & 0x4000u) == want);   // (two producer lanes per piece)
                ok = __all(mine) || (NSD_SCAN_ABLATE & 1) != 0;
                if (!ok) __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && lane == 0) {
                s_abort = 1;                                    // (the wave still walks to the barrier below: the exit is uniform)
                report_timeout(a.status, ST_FWD_TIMEOUT);
            }
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int e = tid + 256 * i, pc = e / MG, row = e % MG;
                *reinterpret_cast<u32x4 *>(Bt + row * LDB + 8 * pc) = u32x4{pv[i][0] & ~TAGBITS, pv[i][1] & ~TAGBITS, pv[i][2] & ~TAGBITS, pv[i][3] & ~TAGBITS};
            }
            __syncthreads();
            if (s_abort) break;                                 // uniform: every thread reads the same word after the barrier
        }
        // the next step's projection tile is requested only AFTER this step's has been unpacked: requested before, hipcc guards
        // the unpack with vmcnt(0) and the wave sits out the HBM latency of the new request every step
        if constexpr (INPROJ) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = bias0[r];
            // All four k-steps, unconditionally (zero weights beyond the channel count), and settled before any control flow: the
            // accumulators of asm MFMAs are ordinary values to hipcc, and around a run-time branch it copies them (AGPR -> VGPR) at
            // once -- reading a result the matrix pipe has not written yet (found as wrong logits with 64-trial tiles).
            mfma_lead_in();
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mfma_acc_v(acc[nt], wx0[k], xf[nt][k]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) mfma_settle(acc[nt]);
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = unpack_tile(xp[nt][0], xp[nt][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < a.T) load_xp(tnx);
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
            const bf16x8 *const ws1[1] = {w};
            const bf16_t *const ts1[1] = {Bt2[s & 1]};
            mfma_pipe<NT, KS, LDB, 1, 12>(ws1, ts1, col, hh, [&](const int, const int nt) -> f32x16 & { return acc[nt]; });
        }
        // ---- LSTM cell in registers: registers 4j..4j+3 = gates i,f,g,o of unit u0 + j for trial b0 + 32nt + col
        float hv[NT][4], gi[NT][4], gf[NT][4], gg[NT][4], go[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gi[nt][j] = fast_sigmoid(acc[nt][4 * j]);
                gf[nt][j] = fast_sigmoid(acc[nt][4 * j + 1]);
                gg[nt][j] = fast_tanh(acc[nt][4 * j + 2]);
                go[nt][j] = fast_sigmoid(acc[nt][4 * j + 3]);
                c[nt][j] = fmaf(gf[nt][j], c[nt][j], gi[nt][j] * gg[nt][j]);
                hv[nt][j] = go[nt][j] * fast_tanh(c[nt][j]);
            }
        // ---- publish h_t into the ring slot of this step, drain, this wave's flag; the row-major copy goes out behind the flag
        unsigned hw[NT][2];
        {
            bf16_t *slot = ring_of(s);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                hw[nt][0] = pack_bf16x2(hv[nt][0], hv[nt][1]);
                hw[nt][1] = pack_bf16x2(hv[nt][2], hv[nt][3]);
                const unsigned tag = ((s >> 1) & 1) ? TAGBITS : 0u;
                // a non-finite h carries bit 14 by itself: the lane is poisoned for this trial from here on -- zeros with the right tag
                // into the ring, NaN into the row-major sequence (the next layer's projection / the head turn the trial's logits into
                // NaN, as torch's nn.LSTM does), NSD_SEQ_ST_NONFINITE into the status word
                if (((hw[nt][0] | hw[nt][1]) & TAGBITS) != 0u) poison[nt] = 0x7FC07FC0u;
                const unsigned p0 = poison[nt] ? 0u : hw[nt][0], p1 = poison[nt] ? 0u : hw[nt][1];
                st_xchg_u64(same_l2, slot + ring_h_off(gt, nt, NT, col, hh), ((unsigned long long)(p1 | tag) << 32) | (p0 | tag));
                hw[nt][0] |= poison[nt]; hw[nt][1] |= poison[nt];
            }
        }
        // ---- everything else of the step leaves behind the flag (nobody waits for it inside this launch)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int b = b0 + 32 * nt + col;
            const long row_u = (((long)(b0 >> 5) + nt) * a.T + t) * 32;                 // seq_row(t, b) = row_u + col
            const unsigned lane_off = (unsigned)(col * (int)ld + dir * H + u0);
            st_stream<u32x2>(a.hs + row_u * ld + lane_off, u32x2{hw[nt][0], hw[nt][1]});
            float m[4] = {1.f, 1.f, 1.f, 1.f};
            if (a.lk) {
                if (a.rng.on && b < a.B) {
                    const uint64_t base = (((uint64_t)a.layer * a.B + b) * a.T + t) * (uint64_t)ld + (uint64_t)(dir * H + u0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = nsd_rand_u32(a.rng.seed, a.rng.base, base + j) >= a.rng.thr_lstm ? a.rng.keep_lstm : 0.f;
                }
                // the multiplier acts on the value the next layer really reads: the bf16 h (+ the residual input, extension)
                float h0 = bf16_lo(hw[nt][0]), h1 = bf16_hi(hw[nt][0]), h2 = bf16_lo(hw[nt][1]), h3 = bf16_hi(hw[nt][1]);
                if (a.res) {
                    const u32x2 rv2 = ld_stream<u32x2>(a.res + row_u * ld + lane_off);
                    h0 += bf16_lo(rv2[0]); h1 += bf16_hi(rv2[0]); h2 += bf16_lo(rv2[1]); h3 += bf16_hi(rv2[1]);
                }
                u32x2 v = {pack_bf16x2(h0 * m[0], h1 * m[1]), pack_bf16x2(h2 * m[2], h3 * m[3])};
                st_stream<u32x2>(a.lk + row_u * ld + lane_off, v);
            }
            if (train) {
                u32x2 cv = {pack_bf16x2(c[nt][0], c[nt][1]), pack_bf16x2(c[nt][2], c[nt][3])};
                const long blk = saved_block((b0 >> 5) + nt, P, me.p, a.T, t, wave_s);
                st_stream<u32x2>(a.cs[dir] + blk * 256 + lane * 4, cv);
                unsigned gw[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gw[2 * j] = pack_bf16x2(gi[nt][j], gf[nt][j]) | (m[j] != 0.f ? 0x8000u : 0u);   // (sign of the saved i: the output survived the dropout behind this layer -- saved_keep_bits)
                    gw[2 * j + 1] = pack_bf16x2(gg[nt][j], go[nt][j]);
                }
                bf16_t *gd = a.ga[dir] + blk * 1024;
                st_stream<u32x4>(gd + lane * 8, u32x4{gw[0], gw[1], gw[2], gw[3]});
                st_stream<u32x4>(gd + 512 + lane * 8, u32x4{gw[4], gw[5], gw[6], gw[7]});
            }
        }
    }
    {
        bool any = false;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) any = any || poison[nt] != 0u;
        if (__any(any) && lane == 0) atomicOr(a.status, NSD_SEQ_ST_NONFINITE);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward scan
// ---------------------------------------------------------------------------------------------------------------------------
// TOP: the layer under the head (upstream term from alpha / dscore / dpooled) or a lower layer (from din).  A template parameter,
// not a run-time branch: with the branch inside load_saved hipcc merges the two arms through copies and guards them with
// vmcnt(2) right behind the loads -- a full HBM latency per 32-trial half and step (11 500 of 20 000 cycles at H = 512).
// DINT (lower layers): the upstream gradient comes as bf16 accumulator tiles written by the input-gradient GEMM in the order this
// kernel's lanes own them (8 contiguous bytes per lane, 512 per wave instruction) instead of row-major fp32, where a lane's 16
// bytes sat in a row of their own: 64 line requests per instruction, ~2 300 cycles of a 16 300-cycle step at H = 512.
template <int H, int NT, bool TOP, bool DINT>
__global__ __launch_bounds__(256) void scan_bwd_kernel(const ScanBwdArgs a) {
    // The recurrent term dh_rec = W_hh^T da_{t+1} is exchanged as a reduce-scatter of bf16 partial sums (design and ring layout:
    // nsd_scan2.hip, "backward"): the workgroup multiplies its OWN 128 gate columns of da (from LDS, K = 128) into partial dh
    // rows for every unit of the layer and sends each member the 32 rows it owns; a member adds the P partials in fp32.
    constexpr int P = H / 32, MG = 32 * NT, G = 4 * H, RT = P >= 4 ? P / 4 : 1, KS = 8;
    constexpr long NBLK = (long)P * P * NT * 4, SLOT_BYTES = NBLK * 512;          // [consumer][producer][nt][consumer wave] x (64 lanes x 8 B)
    __shared__ __align__(16) bf16_t dab[2][KS][NT][512];        // [step parity][k-step = 2 * wave + half][nt][lane * 8]
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Member me = member_of(blockIdx.x, a.groups, P, a.spread_groups);
    const int dir = me.dir;
    const int b0 = (a.group0 + me.group) * MG;
    const int col = lane & 31, hh = lane >> 5;
    const bool has_rows = wave < P;

    // rows of W_hh^T for the consumers r = wave + 4 ri, columns = this workgroup's 128 gate columns in k-step order
    bf16x8 wq[RT][KS];
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
        const int r = (wave + 4 * ri) < P ? wave + 4 * ri : 0;
        const bf16_t *wrow = a.wb[dir] + (long)(32 * r + col) * G + 128 * me.p + 16 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wq[ri][ks] = *reinterpret_cast<const bf16x8 *>(wrow + 32 * (ks >> 1) + 8 * (ks & 1));
    }
    float dc[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dc[nt][j] = 0.f;
    if (tid == 0) s_abort = 0;
    __syncthreads();

    unsigned *gflags = a.flags + (long)(dir * a.groups + me.group) * GROUP_WORDS;
    const int rv = group_rendezvous<P>(gflags, me.p, wave, lane);
    if (rv < 0 && lane == 0) { s_abort = 1; report_timeout(a.status, ST_BWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);
    const long ld = a.ld, ldda = (long)a.D * G;
    const int T = a.T;
    float dbs[16];                                             // bias gradient of this lane's 16 gate columns, summed over time and tiles
#pragma unroll
    for (int k = 0; k < 16; ++k) dbs[k] = 0.f;
    const int u0 = 32 * me.p + 8 * wave + 4 * hh;              // this lane's 4 units (same ownership as the forward scan)
    // loop invariants of the top layer: d out_t = alpha_t * dpooled + dscore_t * attn_w
    float dpl[NT][4], aw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aw[j] = TOP ? a.attn_w[dir * H + u0 + j] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dpl[nt][j] = TOP ? a.dpooled[(long)(b0 + 32 * nt + col) * ld + dir * H + u0 + j] : 0.f;
    const char *ring0 = reinterpret_cast<const char *>(a.xch) + ((long)dir * a.groups_total + a.group0 + me.group) * SLOT_BYTES;
    unsigned *gacks = gflags + ACK_WORD;                      // consume counters of the group (single-slot ring: nsd_scan2.hip, "backward")
    // block (consumer, producer, nt, consumer wave) of 64 lanes x 8 B; with NT == 2 the two halves share one block of 64 lanes x 16 B
    auto blk_off = [&](const int cons, const int prod, const int nt, const int q) {
        return NT == 2 ? (unsigned)((((cons * P + prod) * 4 + q)) << 10) : (unsigned)(((((cons * P + prod) * NT + nt) * 4 + q)) << 9);
    };

    // saved activations and upstream gradient of a step: independent of the recurrence, requested one step ahead (after the
    // barrier of the step before: the MFMA phase follows)
    struct Saved {
        u32x4 gq[NT][2];
        u32x2 cq[NT], cpq[NT];
        f32x4 dv[NT];
        u32x2 dvt[NT];
        float al[NT], ds[NT];
    };
    auto t_of = [&](const int s) { return dir == 0 ? T - 1 - s : s; };   // reverse of the forward order
    // (addresses: a wave-uniform 64-bit base per step -- scalar arithmetic, SGPR base operand -- plus a 32-bit lane offset)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // (the cell state is read once: c_t of a step is the c_{t-1} the step before it used, carried in registers)
    auto load_saved = [&](const int s, Saved &v, auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const int t = t_of(s), tprev = dir == 0 ? t - 1 : t + 1;         // tprev: earlier in forward time (c_{t-1})
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long row_u = (((long)(b0 >> 5) + nt) * T + t) * 32;    // seq_row(t, b0 + 32 nt + col) = row_u + col
            const long blk = saved_block((b0 >> 5) + nt, P, me.p, T, t, wave_s);
            const bf16_t *gs = a.ga[dir] + blk * 1024;
            v.gq[nt][0] = ld_stream<u32x4>(gs + lane * 8);
            v.gq[nt][1] = ld_stream<u32x4>(gs + 512 + lane * 8);
            if constexpr (FIRST) v.cq[nt] = ld_stream<u32x2>(a.cs[dir] + blk * 256 + lane * 4); else v.cq[nt] = v.cpq[nt];
            const bool first = dir == 0 ? t == 0 : t == T - 1;
            v.cpq[nt] = first ? u32x2{0u, 0u} : ld_stream<u32x2>(a.cs[dir] + (blk + 4 * (tprev - t)) * 256 + lane * 4);
            if constexpr (TOP) { v.al[nt] = ld_stream<float>(a.alpha + row_u + col); v.ds[nt] = ld_stream<float>(a.dscore + row_u + col); }
            else if constexpr (DINT) v.dvt[nt] = ld_stream<u32x2>(a.din_tiles + ((((long)((b0 >> 5) + nt) * T + t) * (a.D * P) + dir * P + me.p) * 1024 + wave_s * 256) + lane * 4);
            else v.dv[nt] = ld_stream<f32x4>(a.din + row_u * ld + (unsigned)(col * (int)ld + dir * H + u0));
        }
    };
    Saved sv;
    load_saved(0, sv, std::true_type{});
    Stamps stp;
    stp.start();
    for (int s = 0; s < T; ++s) {
        const int t = t_of(s);
        // ---- ahead of the exchange: the upstream term and everything of the cell that does not need dh
        CellFac fc[NT];
        float dup[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int b = b0 + 32 * nt + col;
            const long row = seq_row(t, b, T);
            cell_factors(sv.gq[nt][0], sv.gq[nt][1], sv.cq[nt], sv.cpq[nt], fc[nt]);
            if constexpr (!TOP) {
                float m[4];
                saved_keep_bits(sv.gq[nt][0], sv.gq[nt][1], a.rng.on ? a.rng.keep_lstm : 1.f, m);   // (the forward scan left them in the saved gates)
                if constexpr (DINT) {
                    dup[nt][0] = bf16_lo(sv.dvt[nt][0]) * m[0]; dup[nt][1] = bf16_hi(sv.dvt[nt][0]) * m[1];
                    dup[nt][2] = bf16_lo(sv.dvt[nt][1]) * m[2]; dup[nt][3] = bf16_hi(sv.dvt[nt][1]) * m[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dup[nt][j] = sv.dv[nt][j] * m[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dup[nt][j] = fmaf(sv.al[nt], dpl[nt][j], sv.ds[nt] * aw[j]);
            }
            if (a.dres) st_stream<f32x4>(a.dres + row * ld + dir * H + u0, f32x4{dup[nt][0], dup[nt][1], dup[nt][2], dup[nt][3]});
            pin(fc[nt]);
#pragma unroll
            for (int j = 0; j < 4; ++j) pin(dup[nt][j]);
        }
        stp.mark(7);
        float drec[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) drec[nt][j] = 0.f;
        if (s > 0) {
            if (!(NSD_SCAN_ABLATE & 1) && !wait_group<4 * P>(gflags, (unsigned)s, lane) && lane == 0) {
                s_abort = 1;
                report_timeout(a.status, ST_BWD_TIMEOUT);
            }
            stp.mark(0);
            // the partial sums the P members sent this wave at step s-1, added in member order
            const nsd_rsrc rr = make_rsrc(ring0, (unsigned)SLOT_BYTES);
            if constexpr (NT == 2) {                             // both 32-trial halves of a lane in ONE 16-byte piece (half the instructions)
                u32x4 v16[P];
#pragma unroll
                for (int q = 0; q < P; ++q) v16[q] = ld_sc1_b128(rr, blk_off(me.p, q, 0, wave) + 16u * lane);
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc_bf16x2(drec[nt][0], drec[nt][1], v16[q][2 * nt]); acc_bf16x2(drec[nt][2], drec[nt][3], v16[q][2 * nt + 1]);
                    }
            } else {
                u32x2 v8[P][NT];
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) v8[q][nt] = ld_sc1_b64(rr, blk_off(me.p, q, nt, wave) + 8u * lane);
#pragma unroll
                for (int q = 0; q < P; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc_bf16x2(drec[nt][0], drec[nt][1], v8[q][nt][0]); acc_bf16x2(drec[nt][2], drec[nt][3], v8[q][nt][1]);
                    }
            }
            // this wave has taken its partial sums of step s-1 out of the ring: the producers may rewrite the slot
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_xchg_u32(same_l2, gacks + 4 * me.p + wave, (unsigned)s);
        }
        stp.mark(1);
        // ---- the dh-dependent rest of the cell: da_t
        unsigned dw[NT][8];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float dh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) dh[j] = dup[nt][j] + drec[nt][j];
            cell_apply(fc[nt], dh, dc[nt], dbs, dw[nt]);
        }
        stp.mark(2);
        if (s + 1 < T) {                                        // (after the last step nobody reads a partial sum)
            const int par = s & 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    *reinterpret_cast<u32x4 *>(&dab[par][2 * wave + h][nt][lane * 8]) = u32x4{dw[nt][4 * h], dw[nt][4 * h + 1], dw[nt][4 * h + 2], dw[nt][4 * h + 3]};
            __syncthreads();
            if (s_abort) break;
            stp.mark(3);
            // the consume counters of the group, requested now and looked at before the first ring store of the step (the MFMAs of
            // a consumer's row tile lie in between): every member must have taken step s-1's sums before the slot is rewritten
            unsigned ackv = ld_sc1_u32(gacks + (lane < 4 * P ? lane : 0));     // (every lane loads: no exec-masked block for the compare to be pulled into)
            load_saved(s + 1, sv, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            if (has_rows) {
                const nsd_rsrc rw = make_rsrc(ring0, (unsigned)SLOT_BYTES);
                constexpr int NF = KS * NT, D = NF < 4 ? NF : 4;
                auto frag = [&](const int i) { return *reinterpret_cast<const bf16x8 *>(&dab[par][i / NT][i % NT][lane * 8]); };
#pragma unroll
                for (int ri = 0; ri < RT; ++ri) {
                    const int r = wave + 4 * ri;
                    f32x16 acc[NT];
                    bf16x8 g[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) g[i] = frag(i);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const int ks = i / NT, nt = i % NT;
                        if (ks == 0) mfma_new_a(acc[nt], wq[ri][ks], g[i % D]); else mfma_acc_a(acc[nt], wq[ri][ks], g[i % D]);
                        if (i + D < NF) g[i % D] = frag(i + D);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mfma_settle(acc[nt]);
                    if (ri == 0) pin(ackv);                              // the compare stays HERE: at the load it would expose an L2 round trip per step
                    if (ri == 0 && !__all(ackv >= (unsigned)s)) {   // (rare: the counters were read ~2 000 cycles after they were written)
                        if (!wait_group<4 * P>(gacks, (unsigned)s, lane) && lane == 0) { s_abort = 1; report_timeout(a.status, ST_BWD_TIMEOUT); }
                    }
                    if constexpr (NT == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            st_ring_b128(same_l2, rw, blk_off(r, me.p, 0, q) + 16u * lane,
                                         u32x4{pack_bf16x2(acc[0][4 * q], acc[0][4 * q + 1]), pack_bf16x2(acc[0][4 * q + 2], acc[0][4 * q + 3]),
                                               pack_bf16x2(acc[1][4 * q], acc[1][4 * q + 1]), pack_bf16x2(acc[1][4 * q + 2], acc[1][4 * q + 3])});
                    } else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                st_ring_b64(same_l2, rw, blk_off(r, me.p, nt, q) + 8u * lane,
                                            u32x2{pack_bf16x2(acc[nt][4 * q], acc[nt][4 * q + 1]), pack_bf16x2(acc[nt][4 * q + 2], acc[nt][4 * q + 3])});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            stp.mark(4);
            if (!(NSD_SCAN_ABLATE & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_xchg_u32(same_l2, gflags + 4 * me.p + wave, (unsigned)(s + 1));
            stp.mark(6);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- row-major da_t for the weight-gradient / input-gradient GEMMs: behind the flag
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16_t *d = a.da + (((long)(b0 >> 5) + nt) * T + t) * 32 * ldda;
            const unsigned lane_off = (unsigned)(col * (int)ldda + dir * G + 4 * u0);
            st_stream<u32x4>(d + lane_off, u32x4{dw[nt][0], dw[nt][1], dw[nt][2], dw[nt][3]});
            st_stream<u32x4>(d + lane_off + 8, u32x4{dw[nt][4], dw[nt][5], dw[nt][6], dw[nt][7]});
        }
    }
    stp.store(a.status, blockIdx.x == 0 && tid == 0);
    // ---- bias gradients of this batch tile: sum over the 32 trials of each half-wave, one row of dbp per (direction, tile)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float v = dbs[k];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
        dbs[k] = v;
    }
    if (col == 0) {
        float *dst = a.dbp + ((long)dir * a.groups_total + a.group0 + me.group) * G + 4 * u0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) *reinterpret_cast<f32x4 *>(dst + k) = f32x4{dbs[k], dbs[k + 1], dbs[k + 2], dbs[k + 3]};
    }
}

template <int H>
int launch_fwd_h(const ScanFwdArgs &a, int MG, const dim3 grid, hipStream_t st) {
    if (a.wx0[0] != nullptr) {
        if (MG == 32) hipLaunchKernelGGL((scan_fwd_kernel<H, 1, true>), grid, dim3(256), 0, st, a);
        else          hipLaunchKernelGGL((scan_fwd_kernel<H, 2, true>), grid, dim3(256), 0, st, a);
    } else {
        if (MG == 32) hipLaunchKernelGGL((scan_fwd_kernel<H, 1, false>), grid, dim3(256), 0, st, a);
        else          hipLaunchKernelGGL((scan_fwd_kernel<H, 2, false>), grid, dim3(256), 0, st, a);
    }
    NSD_CHECK_LAUNCH("scan_fwd_kernel");
    return NSD_OK;
}
template <int H>
int launch_bwd_h(const ScanBwdArgs &a, int MG, const dim3 grid, hipStream_t st) {
    if (a.din == nullptr && a.din_tiles == nullptr) {
        if (MG == 32) hipLaunchKernelGGL((scan_bwd_kernel<H, 1, true, false>), grid, dim3(256), 0, st, a);
        else          hipLaunchKernelGGL((scan_bwd_kernel<H, 2, true, false>), grid, dim3(256), 0, st, a);
    } else if (a.din_tiles != nullptr) {
        if (MG == 32) hipLaunchKernelGGL((scan_bwd_kernel<H, 1, false, true>), grid, dim3(256), 0, st, a);
        else          hipLaunchKernelGGL((scan_bwd_kernel<H, 2, false, true>), grid, dim3(256), 0, st, a);
    } else {
        if (MG == 32) hipLaunchKernelGGL((scan_bwd_kernel<H, 1, false, false>), grid, dim3(256), 0, st, a);
        else          hipLaunchKernelGGL((scan_bwd_kernel<H, 2, false, false>), grid, dim3(256), 0, st, a);
    }
    NSD_CHECK_LAUNCH("scan_bwd_kernel");
    return NSD_OK;
}

}  // namespace

bool nsd_scan_supported(int H) { return H == 64 || H == 128 || H == 256 || H == 512; }

static int check_grid(const char *who, int H, int MG, int groups, int D) {
    if (!nsd_scan_supported(H) || (MG != 32 && MG != 64) || groups < 1 || D < 1 || D > NSD_SEQ_MAX_DIRS) {
        nsd_set_error("%s: unsupported geometry H=%d MG=%d groups=%d D=%d", who, H, MG, groups, D);
        return NSD_E_INVALID;
    }
    // every member of every group must be resident at the same time: one workgroup per CU
    const int need = D * groups * (H / 32);
    if (need > nsd_num_cus()) {
        nsd_set_error("%s: %d workgroups needed but only %d CUs: split the batch (internal error of the caller)", who, need, nsd_num_cus());
        return NSD_E_INVALID;
    }
    return NSD_OK;
}

int nsd_scan_fwd_launch(const ScanFwdArgs &a, int H, int MG, hipStream_t st) {
    if (const int rc = check_grid("scan_fwd", H, MG, a.groups, a.D)) return rc;
    const dim3 grid(a.D * a.groups * (H / 32) - a.diag_short_grid);
    switch (H) {
    case 64: return launch_fwd_h<64>(a, MG, grid, st);
    case 128: return launch_fwd_h<128>(a, MG, grid, st);
    case 256: return launch_fwd_h<256>(a, MG, grid, st);
    default: return launch_fwd_h<512>(a, MG, grid, st);
    }
}
int nsd_scan_bwd_launch(const ScanBwdArgs &a, int H, int MG, hipStream_t st) {
    if (const int rc = check_grid("scan_bwd", H, MG, a.groups, a.D)) return rc;
    const dim3 grid(a.D * a.groups * (H / 32) - a.diag_short_grid);
    switch (H) {
    case 64: return launch_bwd_h<64>(a, MG, grid, st);
    case 128: return launch_bwd_h<128>(a, MG, grid, st);
    case 256: return launch_bwd_h<256>(a, MG, grid, st);
    default: return launch_bwd_h<512>(a, MG, grid, st);
    }
}

int nsd_seq_prep_launch(const PrepArgs &a, hipStream_t st) {
    const long total = 2L * 4 * a.H * a.H + 4L * a.H * a.Ipad + (a.wxt ? 4L * a.H * a.I : 0) + 4L * a.H;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(seq_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    NSD_CHECK_LAUNCH("seq_prep_kernel");
    return NSD_OK;
}
int nsd_seq_xbf_launch(const float *x, bf16_t *xbf, int B, int Bp, int T, int C, int CP, hipStream_t st) {
    const long total = (long)T * Bp * CP;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(seq_xbf_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, xbf, B, Bp, T, C, CP);
    NSD_CHECK_LAUNCH("seq_xbf_kernel");
    return NSD_OK;
}
