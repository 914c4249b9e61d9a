// nsd_lstm2_bwd48.hip -- BPTT of the two-layer H=48 LSTM, role-split workgroup (gfx950).
//
// Replaces autograd through self.lstm(x) (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:34) for the reference
// model shape (H=48, L=2, C<=8).  One 1024-thread workgroup (16 waves) owns NB trials and walks time backwards with ONE barrier per
// step; the waves have different jobs so that only the true recurrence is on the critical path and every register array stays small.
//
// ONE trial per workgroup (NB = 1: the benchmark's 256 trials on 256 CUs) -- per-wave stamps say a step is as long as the instruction
// stream of its busiest wave, so everything that is not the recurrence has been taken out of the six recurrence waves:
//   "chain 1" x3  layer-1 cell backward: dh_rec = W_hh1^T da1[t+1] (48 FMA/lane as 24 v_pk_fma_f32, weights in VGPRs, 12 operands per lane
//                 from LDS, DPP reduction over 16 lanes), then FOUR operations on the cell's ready-made factors -> da1[t] into an LDS ring.
//   "chain 0" x3  the same for layer 0, FIVE macro steps behind layer 1.
//   "x1" x3       the hand-off d_in1 = W_ih1^T da1 for FOUR layer-1 steps at a time on the matrix pipe (v_mfma_f32_4x4x1, the steps as the
//                 instruction's columns: exact fp32; x1m_role); waves 0 / 1 also PREPARE the next step's factors of layer 1 / 0 once per
//                 unit from the staged records (prep_load / prep_finish: 32-byte records {dc/dh, f, gradient from above or multiplier, -,
//                 i'g, c f', i g', tanh(c) o'}) -- what each of the four gate lanes of a unit used to form for itself.
//   "dW" x6       weight gradients dW_hh1, dW_ih1, dW_hh0, dW_ih0 = sum_t da[t] (x) operand[t] as SPLIT-bf16 products on
//                 v_mfma_f32_32x32x16_bf16, K = 16 macro steps per instruction (dw16_role): 180 matrix-pipe cycles per step instead of
//                 864 for the fp32 form; waves 0..3 also bring the saved rows (four steps per request), waves 4, 5 convert da.
//   "loader"      LDS-DMA stream of the saved activations, one 8-step chunk ahead; the factors of a trial's first step.
//   Roles are placed by SIMD (wave & 3), priorities follow the measured critical waves (kernel body).
// TWO trials per workgroup (NB = 2, batches of 257 .. 575 trials): the first-generation roles -- per-gate-lane factors in the chains,
// x1 step by step on the VALU with dW_ih0, weight gradients as v_mfma_f32_16x16x4_f32 with K = 4 time steps (dw_role).
//
// LDS: da ring [2 layers][8 steps][NB][192] + staged records + (NB = 1) the bf16 operand windows, factor records and the d_in1 ring.
// HBM traffic = saved activations read once (gates, c, h, in1, x) + one slab of partial gradients per workgroup at the end.
#include "nsd_args.h"
#include "nsd_prof.h"
#include "nsd_bf16.h"
#include <type_traits>
#include <utility>

namespace {

#ifndef NSD_B48_X1M
#define NSD_B48_X1M 1             // one trial per workgroup: the hand-off d_in1 = W_ih1^T da1 as v_mfma_f32_4x4x1 over FOUR steps at a time (layer 0 runs 5 macro steps behind layer 1 instead of 2)
#endif
constexpr int H = 48;
constexpr int G4 = 192;
// macro steps layer 0 runs behind layer 1, minus the 2 of the step-by-step hand-off
constexpr int dl0(const int nb) { return (nb == 1 && NSD_B48_X1M) ? 3 : 0; }
constexpr int RING = 8;
constexpr int NTHREADS = 1024;
#ifndef NSD_DW_SLEEP
#define NSD_DW_SLEEP 0
#endif
#ifndef NSD_LD_SLEEP
#define NSD_LD_SLEEP 0
#endif
constexpr int DW_SLEEP = NSD_DW_SLEEP, LD_SLEEP = NSD_LD_SLEEP;   // post-barrier s_sleep of the waves off the critical path (step_barrier)
constexpr int CHUNK = 8;            // macro steps per staged chunk of saved activations
constexpr int REC = 288;            // floats per (layer, step) record: gates[192] | c[t-1][48] | aux[48]
constexpr int REC4 = REC / 4;
constexpr int STAGE_F4 = 2 * CHUNK * REC4;     // float4 per trial per chunk (both layers) = 1152 = 18 x 64 lanes
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The saved activations a chain wave needs at a step (its unit's 4 activated gates, c[t-1], and the
// per-step scalars) are NOT fetched from HBM by the chain itself: a dependent global load per step would
// put the memory latency (>= 1 us under load) on the recurrence.  A dedicated loader wave streams them one
// 8-step chunk ahead with LDS-DMA (global_load_lds_dwordx4: no VGPRs, asynchronous) into a double-buffered
// LDS stage; the chain only does LDS reads.
// One trial per workgroup: the weight gradients' operands as k-major bf16 windows of 16 macro steps (see "dW waves" below):
// [layer][hi, lo][window (m >> 4) & 1][k = m & 15][column]
constexpr int WK = 16;
constexpr int ARS = 224;            // bf16 elements per k row of the da windows: 192 + pad (448 B = 192 mod 256: the four k rows of a transposed read's block fall into four different 64-byte bank groups)
constexpr int BRS = 96;             // ... of the row windows: layer 1 {h1[t-1] | in1[t]}, layer 0 {h0[t-1] | x[t] (16 columns, C used) | -}
struct DwWin {
    unsigned short wa[2][2][2][WK][ARS];
    unsigned short wb[2][2][2][WK][BRS];
    // per-step factors of the cells, prepared one step ahead by the loader wave (see prep below): [macro step & 1][layer][unit]
    // {d c_t / d h_t, f_t, gradient from above (layer 1) or dropout multiplier (layer 0), -, i'g, c[t-1] f', i g', tanh(c_t) o'}
    float pf[2][2][H][8];
    float din1x[8][H];               // d_in1 of layer-1 macro step c at [c & 7] (the four-step hand-off)
};
struct NoWin {};
template <int NB>
struct Smem {
    float ring[2][RING][NB][G4];           // [layer][macro step % RING][trial][gate*48+unit]
    float din1[2][NB][H];
    float stage[2][NB][2][CHUNK][REC];     // [buffer][trial][layer][step in chunk][record]  (linear per trial)
    float xst[2][NB][CHUNK][8];            // (two trials per workgroup only) x[T+2-m] rows for the x1 waves (dW_ih0), same chunking
    typename std::conditional<NB == 1, DwWin, NoWin>::type win;
};

// ------------------------------------------------------------------------------------------------
// chain waves: layer = 1 (t = T-1-m) or 0 (t = T+1-m)
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// Transposed mat-vec of a chain / x1 wave: out[j] = sum_{r<192} W[r][j] * v[r] for the wave's 16 units.
// Lane (row of 16 lanes = output group og of 4 units, kk = lane & 15 = slice of 12 inputs): 12 operands from LDS
// (3 x 16 B; the previous layout -- one unit x 48 inputs per lane -- needed 12 x 16 B and made the LDS pipe the
// busiest unit of the step) against 4 x 6 weight pairs, four independent accumulator pairs (depth 6), a
// reduce-scatter over the quad and two row rotations.  Returns out[4*og + (lane & 3)], replicated over the 4 quads
// of the row -- which is where the 4 gates of that unit are evaluated.
__device__ __forceinline__ float slice_dot_t(const float *dv, const f32x2 (&wp)[4][6]) {
    f32x2 acc[4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(dv + 4 * q);
        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc[jj] = q == 0 ? wp[jj][0] * lo : pk_fma(wp[jj][2 * q], lo, acc[jj]);
            acc[jj] = pk_fma(wp[jj][2 * q + 1], hi, acc[jj]);
        }
    }
    const int lane = threadIdx.x;
    const bool odd = (lane & 1) != 0, hi2 = (lane & 2) != 0;
    const float r0 = acc[0].x + acc[0].y, r1 = acc[1].x + acc[1].y;
    const float r2 = acc[2].x + acc[2].y, r3 = acc[3].x + acc[3].y;
    const float ra = (odd ? r1 : r0) + quad_xor1(odd ? r0 : r1);
    const float rb = (odd ? r3 : r2) + quad_xor1(odd ? r2 : r3);
    float v = (hi2 ? rb : ra) + quad_xor2(hi2 ? ra : rb);     // unit (lane & 3) over this quad's 4 slices
    v += row_ror<4>(v);
    v += row_ror<8>(v);                                        // ... over all 16 slices
    return v;
}

// weights of slice_dot_t for W [192][48] row-major (nn.LSTM weight_hh / weight_ih layout)
__device__ __forceinline__ void load_wT(const float *w, const int og, const int kk, f32x2 (&wp)[4][6]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            wp[jj][p].x = w[(size_t)(12 * kk + 2 * p) * H + 4 * og + jj];
            wp[jj][p].y = w[(size_t)(12 * kk + 2 * p + 1) * H + 4 * og + jj];
        }
}

// One trial per workgroup: the per-step factors of one layer's cells for macro step mn, from the staged records -- what every gate
// lane of the recurrences used to form for itself, once per unit and one step ahead (by x1 waves 0 / 1 for layer 1 / 0; by the loader
// wave for a trial's first step).  c_t is the c[t-1] field of the record one macro step earlier (c[T-1], `cT1`, at the first step).
struct PrepIn { float4 gc; float cprev, ct, mk; float2 ad; };
__device__ __forceinline__ PrepIn prep_load(const Lstm2BwdArgs &a, Smem<1> &sm, const int mn, const int layer, const int u, const float cT1) {
    const int T = a.T;
    const int t = layer == 1 ? T - 1 - mn : T + 1 + dl0(1) - mn;
    const float *rec = &sm.stage[(mn >> 3) & 1][0][layer][mn & 7][0];
    const float *recp = &sm.stage[((mn - 1) >> 3) & 1][0][layer][(mn - 1) & 7][0];
    PrepIn in;
    in.gc = *reinterpret_cast<const float4 *>(rec + 4 * u);
    in.cprev = t > 0 ? rec[192 + u] : 0.f;
    in.ct = t == T - 1 ? cT1 : recp[192 + u];
    in.ad = make_float2(0.f, 0.f); in.mk = 1.f;
    if (layer == 1) in.ad = *reinterpret_cast<const float2 *>(rec + 240);
    else if (a.mask || a.rng.on) in.mk = rec[240 + u];
    return in;
}
__device__ __forceinline__ void prep_finish(Smem<1> &sm, const PrepIn &in, const int mn, const int layer, const int u, const float dpu, const float awu) {
    const float ig = in.gc.x, fg = in.gc.y, gg = in.gc.z, og = in.gc.w;
    const float tc = fast_tanh(in.ct);
    const float x = layer == 1 ? fmaf(in.ad.x, dpu, in.ad.y * awu) : in.mk;
    float *o = &sm.win.pf[mn & 1][layer][u][0];
    *reinterpret_cast<float4 *>(o) = make_float4(og * (1.f - tc * tc), fg, x, 0.f);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(gg * (ig * (1.f - ig)), in.cprev * (fg * (1.f - fg)), ig * (1.f - gg * gg), tc * (og * (1.f - og)));
}
__device__ __forceinline__ void prep_layer(const Lstm2BwdArgs &a, Smem<1> &sm, const int mn, const int layer, const int u,
                                           const float dpu, const float awu, const float cT1) {
    const PrepIn in = prep_load(a, sm, mn, layer, u, cT1);
    prep_finish(sm, in, mn, layer, u, dpu, awu);
}

template <int NB>
__device__ __forceinline__ void chain_role(const Lstm2BwdArgs &a, Smem<NB> &sm, const int layer, const int r,
                                           const int n_steps) {
    // lane -> mat-vec coordinates (output group og, input slice kk) and cell coordinates (unit j, gate s)
    const int og = r >> 4, kk = r & 15;
    const int j = 4 * og + (r & 3), s = (r >> 2) & 3;
    const int T = a.T, B = a.B;
    const float *cseq = layer == 0 ? a.cseq0 : a.cseq1;
    f32x2 wp[4][6];
    load_wT(layer == 0 ? a.w_hh0 : a.w_hh1, og, kk, wp);
    const float awj = a.attn_w[j];
    float db = 0.f;
    Prof prof = prof_init(a.dbg);

    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float dc[NB], dhrec[NB], ct[NB], dpj[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int b = b0 + n;
            dc[n] = 0.f; dhrec[n] = 0.f;
            dpj[n] = (NB == 2 && layer == 1 && b < B) ? a.dpooled[(size_t)b * H + j] : 0.f;
            ct[n] = (NB == 2 && b < B) ? cseq[((size_t)b * T + (T - 1)) * H + j] : 0.f;   // c[T-1] of the first step
        }
        step_barrier<false>(prof);      // chunk 0 of the stage has been written by the loader wave

        // unrolled by the ring length (== stage chunk): every LDS offset of a step is an immediate
        for (int m0 = 0; m0 < n_steps; m0 += CHUNK) {
            const int sb = (m0 / CHUNK) & 1;
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) {
                const int m = m0 + k;
                const int t = layer == 1 ? (T - 1 - m) : (T + 1 + dl0(NB) - m);
                const bool active = (t >= 0 && t < T);
                const bool prev_active = (t + 1 >= 0 && t + 1 < T);
                if constexpr (NB == 1) {
                    // One trial per workgroup: everything that does not depend on the recurrence comes READY from the loader wave's
                    // prep of the previous step (two LDS reads instead of six and ~20 instructions less per step in the six waves whose
                    // instruction streams ARE the step: the same factors as below, formed once per unit instead of once per gate lane)
                    const float *pfl = &sm.win.pf[k & 1][layer][j][0];
                    const float4 pf = *reinterpret_cast<const float4 *>(pfl);
                    const float qr = pfl[4 + s];
                    const float dout = layer == 1 ? pf.z : (NSD_B48_X1M ? sm.win.din1x[(k + 6 - dl0(NB)) & 7][j] : sm.din1[(k + 1) & 1][0][j]) * pf.z;
                    if (prev_active) dhrec[0] = slice_dot_t(&sm.ring[layer][(k + RING - 1) & (RING - 1)][0][12 * kk], wp);
                    if (active) {
                        const float dht = dout + dhrec[0];
                        const float dct = fmaf(dht, pf.x, dc[0]);
                        const float mine = (s == 3 ? dht : dct) * qr;
                        dc[0] = dct * pf.y;
                        db += mine;
                        sm.ring[layer][k][0][s * H + j] = mine;
                    }
                } else
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    // ---- everything that does not depend on the recurrence: the step's record from the LDS stage, the
                    // derivative factors of this lane's gate, the gradient arriving from above
                    const float *rec = &sm.stage[sb][n][layer][k][0];
                    const float4 gc = *reinterpret_cast<const float4 *>(rec + 4 * j);
                    const float own = rec[4 * j + s];
                    const float cprev = t > 0 ? rec[192 + j] : 0.f;
                    const float aux0 = layer == 1 ? rec[240] : ((a.mask || a.rng.on) ? rec[240 + j] : 1.f);
                    const float aux1 = rec[241];
                    const float ig = gc.x, fg = gc.y, gg = gc.z, og = gc.w;
                    const float tc = fast_tanh(ct[n]);
                    const float wq = og * (1.f - tc * tc);                     // d c_t / d h_t path
                    const float dact = s == 2 ? 1.f - own * own : own * (1.f - own);
                    const float qsel = s == 0 ? gg : s == 1 ? cprev : s == 2 ? ig : tc;
                    const float qr = qsel * dact;                               // da_s = (s == 3 ? dh : dc) * qr
                    float dout;
                    if (layer == 1) dout = fmaf(aux0, dpj[n], aux1 * awj);
                    else            dout = sm.din1[(k + 1) & 1][n][j] * aux0;
                    // ---- the recurrence
                    if (prev_active) dhrec[n] = slice_dot_t(&sm.ring[layer][(k + RING - 1) & (RING - 1)][n][12 * kk], wp);
                    if (active) {
                        float mine = 0.f;                                       // trials past B keep the ring clean
                        if (b0 + n < B) {
                            const float dht = dout + dhrec[n];
                            const float dct = fmaf(dht, wq, dc[n]);
                            mine = (s == 3 ? dht : dct) * qr;
                            dc[n] = dct * fg;
                            db += mine;
                            ct[n] = cprev;   // c[t-1] is the cell state of the next step handled
                        }
                        sm.ring[layer][k][n][s * H + j] = mine;
                    }
                }
                step_barrier<false>(prof);
            }
        }
    }
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    prof_store(a.dbg, prof);
    if (layer == 0) { slab[a.o_b_ih0 + s * H + j] = db; slab[a.o_b_hh0 + s * H + j] = db; }
    else            { slab[a.o_b_ih1 + s * H + j] = db; slab[a.o_b_hh1 + s * H + j] = db; }
}

// ------------------------------------------------------------------------------------------------
// x1 waves: d_in1[t] = W_ih1^T da1[t] (+ residual pass-through) for layer 0, and dW_ih0 (K = C <= 8)
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void x1_role(const Lstm2BwdArgs &a, Smem<NB> &sm, const int r, const int n_steps) {
    const int og = r >> 4, kk = r & 15;
    const int j = 4 * og + (r & 3), s = (r >> 2) & 3;        // s: which copy of unit j this lane is (dW_ih0 channel pair)
    const int T = a.T, B = a.B, C = a.C;
    f32x2 wp[4][6];
    load_wT(a.w_ih1, og, kk, wp);
    float dWih0[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) { dWih0[g][0] = 0.f; dWih0[g][1] = 0.f; }
    const float awj = a.attn_w[j];
    const int c0 = 2 * s, c1 = 2 * s + 1;
    Prof prof = prof_init(a.dbg);

    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float dpj[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) dpj[n] = (a.residual && b0 + n < B) ? a.dpooled[(size_t)(b0 + n) * H + j] : 0.f;
        // prep duty (one trial per workgroup): wave 0 of the role prepares layer 1's factors of the NEXT macro step, wave 1 layer 0's
        const int pw = r >> 6, pu = (r & 63) < H ? (r & 63) : (r & 63) - 16, pl = pw == 0 ? 1 : 0;
        float p_dp = 0.f, p_aw = 0.f, p_c = 0.f;
        if (NB == 1 && pw < 2) {
            p_dp = a.dpooled[(size_t)b0 * H + pu];
            p_aw = a.attn_w[pu];
            p_c = (pl == 1 ? a.cseq1 : a.cseq0)[((size_t)b0 * T + (T - 1)) * H + pu];
        }
        step_barrier<false>(prof);

        for (int m0 = 0; m0 < n_steps; m0 += CHUNK) {
            const int sb = (m0 / CHUNK) & 1;
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) {
                const int m = m0 + k;
                const int t1p = T - m;          // layer-1 step whose da1 was written at macro step m-1
                const int t0p = T + 2 - m;      // layer-0 step whose da0 was written at macro step m-1
                constexpr int PREV = RING - 1;
                const int e = (k + PREV) & (RING - 1);
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    float2 xv = {0.f, 0.f};
                    if (NB == 2) xv = *reinterpret_cast<const float2 *>(&sm.xst[sb][n][k][2 * s]);
                    PrepIn pin;
                    if constexpr (NB == 1) { if (pw < 2) pin = prep_load(a, sm, m + 1, pl, pu, p_c); }      // (requested ahead of the mat-vec: its latency hides there)
                    if (t1p >= 0 && t1p < T) {
                        float inp = slice_dot_t(&sm.ring[1][e][n][12 * kk], wp);
                        if (a.residual && b0 + n < B) {
                            // dout1[t1p] = alpha*dpooled + dscore*attn_w: the scalars sit in the record of macro step m-1
                            // (for k == 0 that is step 7 of the other stage buffer)
                            const float *recp = &sm.stage[k == 0 ? sb ^ 1 : sb][n][1][e][0];
                            inp += fmaf(recp[240], dpj[n], recp[241] * awj);
                        }
                        if (s == 0) sm.din1[k & 1][n][j] = inp;
                    }
                    if constexpr (NB == 1) { if (pw < 2) prep_finish(sm, pin, m + 1, pl, pu, p_dp, p_aw); }
                    if (NB == 2 && t0p >= 0 && t0p < T && b0 + n < B) {       // (one trial per workgroup: dW_ih0 rides in the dW waves' layer-0 window)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float d = sm.ring[0][e][n][g * H + j];
                            dWih0[g][0] = fmaf(d, xv.x, dWih0[g][0]);
                            dWih0[g][1] = fmaf(d, xv.y, dWih0[g][1]);
                        }
                    }
                }
                step_barrier<false>(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    if (NB == 2)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (c0 < C) slab[a.o_w_ih0 + (size_t)(g * H + j) * C + c0] = dWih0[g][0];
        if (c1 < C) slab[a.o_w_ih0 + (size_t)(g * H + j) * C + c1] = dWih0[g][1];
    }
}

// ------------------------------------------------------------------------------------------------
// x1 waves, one trial per workgroup: d_in1[c] = W_ih1^T da1[c] for FOUR layer-1 steps at a time on the matrix pipe -- exact fp32:
// v_mfma_f32_4x4x1_16B_f32 with the four steps as the instruction's columns (what nsd_lstm2_bwd48x4.hip does with four trials: a wave
// owns 16 output units and splits k over the four 16-lane rows, block (row ks, ub) = units 4 ub .. + 3 x k = 48 ks .. + 47 x 4 steps,
// A = W^T resident in VGPRs, B = da1[k][step] straight from the ring (ONE ds_read_b128 per four MFMAs), then a reduce-scatter of the
// four k slices over the rows with v_permlane32_swap / v_permlane16_swap).  48 MFMAs + 12 LDS reads per wave every fourth step
// instead of 24 v_pk_fma_f32 + a 14-instruction reduction EVERY step; layer 0 runs 5 macro steps behind layer 1 instead of 2.
// Waves 0 / 1 of the role also prepare the cells' factors of the next macro step (prep_load / prep_finish).
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows_reduce_scatter4(const f32x4 v) {
    const u32x2v p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
    const u32x2v p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
    const float s02 = __uint_as_float(p02[0]) + __uint_as_float(p02[1]);
    const float s13 = __uint_as_float(p13[0]) + __uint_as_float(p13[1]);
    const u32x2v q = __builtin_amdgcn_permlane16_swap(__float_as_uint(s02), __float_as_uint(s13), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ void x1m_role(const Lstm2BwdArgs &a, Smem<1> &sm, const int g, const int lane, const int n_steps) {
    const int T = a.T, B = a.B;
    const int ks = lane >> 4, ub = (lane >> 2) & 3, jc = lane & 3;  // MFMA operand coordinates: k slice, unit block, A: unit in block / B: step
    float wv[H];
#pragma unroll
    for (int sidx = 0; sidx < H; ++sidx) wv[sidx] = a.w_ih1[(size_t)(48 * ks + sidx) * H + 16 * g + 4 * ub + jc];
    const int uo = 16 * g + 4 * ub + ks;                            // after the reduce-scatter: this lane's unit, for step c0 + jc
    const float awo = a.attn_w[uo];
    // prep duty: wave 0 of the role prepares layer 1's factors of the NEXT macro step, wave 1 layer 0's
    const int pu = lane < H ? lane : lane - 16, pl = g == 0 ? 1 : 0;
    const float p_aw = a.attn_w[pu];
    Prof prof = prof_init(a.dbg);
    for (int grp = blockIdx.x; grp < B; grp += gridDim.x) {
        const int b0 = grp;
        const float dpo = a.residual ? a.dpooled[(size_t)b0 * H + uo] : 0.f;
        float p_dp = 0.f, p_c = 0.f;
        if (g < 2) {
            p_dp = a.dpooled[(size_t)b0 * H + pu];
            p_c = (pl == 1 ? a.cseq1 : a.cseq0)[((size_t)b0 * T + (T - 1)) * H + pu];
        }
#pragma unroll
        for (int s4 = 0; s4 < H; s4 += 8)
            asm volatile("" : "+v"(wv[s4]), "+v"(wv[s4 + 1]), "+v"(wv[s4 + 2]), "+v"(wv[s4 + 3]), "+v"(wv[s4 + 4]), "+v"(wv[s4 + 5]), "+v"(wv[s4 + 6]), "+v"(wv[s4 + 7]));
        step_barrier<false>(prof);
        for (int m0 = 0; m0 < n_steps; m0 += CHUNK) {
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) {
                const int m = m0 + k;
                PrepIn pin;
                if (g < 2) pin = prep_load(a, sm, m + 1, pl, pu, p_c);      // (requested ahead of the products: its latency hides there)
                if ((k & 3) == 0 && m >= 4) {
                    // layer-1 macro steps c0 .. c0 + 3 = m - 4 .. m - 1 (ring slots (k + 4 + j) & 7); a column past the trial's first step
                    // (c > T - 1) holds stale da and is never used
                    const float *vj = &sm.ring[1][(k + 4 + jc) & 7][0][48 * ks];
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma unroll
                    for (int qb = 0; qb < 12; qb += 4) {
                        f32x4 bq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(vj + 4 * (qb + q));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[4 * (qb + q) + 0], bq[q][0], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[4 * (qb + q) + 1], bq[q][1], acc1, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[4 * (qb + q) + 2], bq[q][2], acc2, 0, 0, 0);
                            acc3 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[4 * (qb + q) + 3], bq[q][3], acc3, 0, 0, 0);
                        }
                    }
                    float inp = rows_reduce_scatter4((acc0 + acc1) + (acc2 + acc3));
                    const int c = m - 4 + jc;
                    if (a.residual) {       // dout1 = alpha * dpooled + dscore * attn_w passes through: the scalars sit in the record of macro step c
                        const float *recp = &sm.stage[(c >> 3) & 1][0][1][c & 7][0];
                        inp += fmaf(recp[240], dpo, recp[241] * awo);
                    }
                    sm.win.din1x[(k + 4 + jc) & 7][uo] = inp;
                }
                if (g < 2) prep_finish(sm, pin, m + 1, pl, pu, p_dp, p_aw);
                if (g == 2 && a.da0_out && m >= 1) {
                    // the input gradient's operand (nsd_lstm_bwd with dx): da0 of macro step m - 1 leaves as it is, 768 bytes per step, from
                    // the one wave of the role without a prep duty (in the dW waves the test alone cost the training kernel 8 us)
                    const int t0 = T + 2 + dl0(1) - m;
                    if (t0 >= 0 && t0 < T && lane < H)
                        *reinterpret_cast<f32x4 *>(a.da0_out + ((size_t)b0 * T + t0) * G4 + 4 * lane) =
                            *reinterpret_cast<const f32x4 *>(&sm.ring[0][(k + 7) & 7][0][4 * lane]);
                }
                step_barrier<false>(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// dW waves: three [192,48] outer-product sums on the matrix pipe, K = 4 time steps per MFMA
// ------------------------------------------------------------------------------------------------
template <int NB>
struct DwState {
    f32x4 acc[3][2][3];      // [matrix: W_hh1, W_ih1, W_hh0][row tile][column tile]
    float bq[3][NB][3];      // prefetched B operands of the 4-step group being formed
};

// B operand of matrix Q for the 4 macro steps of group G: lane (col i = lane&15, k = lane>>4)
template <int Q, int NB>
__device__ __forceinline__ void dw_prefetch(const Lstm2BwdArgs &a, DwState<NB> &st, const int G, const int b0,
                                            const int lane) {
    const int T = a.T, B = a.B;
    const int i = lane & 15, k = lane >> 4;
    const int mm = 4 * G + k;
    // row of the saved sequence that pairs with da[t]:  W_hh1: h1[t-1], W_ih1: in1[t], W_hh0: h0[t-1]
    const int t = (Q == 2) ? (T + 1 - mm) : (T - 1 - mm);
    const int tt = (Q == 1) ? t : t - 1;
    const bool ok = (t >= 0 && t < T && tt >= 0);
    const float *src = Q == 0 ? a.hseq1 : Q == 1 ? a.in1seq : a.hseq0;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int b = b0 + n;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
            st.bq[Q][n][nt] = (ok && b < B) ? src[((size_t)b * T + tt) * H + 16 * nt + i] : 0.f;
    }
}

template <int Q, int NB>
__device__ __forceinline__ void dw_compute(const Lstm2BwdArgs &a, Smem<NB> &sm, DwState<NB> &st, const int G,
                                           const int dwid, const int lane) {
    // group G (macro steps 4G..4G+3) is complete in the ring; its B operands sit in st.bq[Q]
    const int T = a.T;
    const int i = lane & 15, k = lane >> 4;
    const int mm = 4 * G + k;
    constexpr int layer = (Q == 2) ? 0 : 1;
    const int t = layer == 1 ? (T - 1 - mm) : (T + 1 - mm);
    const bool ok = (t >= 0 && t < T);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        float av[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float v = sm.ring[layer][mm & (RING - 1)][n][32 * dwid + 16 * mt + i];
            av[mt] = ok ? v : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
                st.acc[Q][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], st.bq[Q][n][nt], st.acc[Q][mt][nt], 0, 0, 0);
    }
}

template <int NB>
__device__ __forceinline__ void dw_role(const Lstm2BwdArgs &a, Smem<NB> &sm, const int dwid, const int lane,
                                        const int n_groups) {
    DwState<NB> st;
    Prof prof = prof_init(a.dbg);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) st.acc[q][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // rng.on: the layer-0 dropout multipliers (the `aux` slot of the layer-0 records) are generated here instead of being
    // streamed from HBM: 8 steps x 48 units per chunk = one value per dW lane and chunk, a chunk ahead like the loader
    // (same values as nsd_train_masks / the forward kernel)
    const int L = dwid * 64 + lane, mk_k = L / H, mk_j = L - mk_k * H;
    auto gen_mask = [&](const int chunk, const int b0) {
        const int t = a.T + 1 - (chunk * CHUNK + mk_k);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int b = b0 + n;
            if (b < a.B && t >= 0 && t < a.T)
                sm.stage[chunk & 1][n][0][mk_k][240 + mk_j] =
                    nsd_rand_u32(a.rng.seed, a.rng.base, ((uint64_t)b * a.T + t) * H + mk_j) < a.rng.thr_lstm ? 0.f : a.rng.keep_lstm;
        }
    };
    const int ngrp = (a.B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        if (a.rng.on) gen_mask(0, b0);
        step_barrier<false>(prof);      // pairs with the stage-initialisation barrier of the other roles
        for (int G = 0; G < n_groups; ++G) {
            // macro step 4G+0 : finish matrix 0 of group G-1, start fetching matrix 0 of group G; etc.
            if (G > 0 && !ablated(a.ablate, 1)) dw_compute<0, NB>(a, sm, st, G - 1, dwid, lane);
            dw_prefetch<0, NB>(a, st, G, b0, lane);
            step_barrier<false, DW_SLEEP>(prof);
            if (a.rng.on && (G & 1) == 0) gen_mask(G / 2 + 1, b0);     // second step of a chunk: the old records are dead
            if (G > 0 && !ablated(a.ablate, 1)) dw_compute<1, NB>(a, sm, st, G - 1, dwid, lane);
            dw_prefetch<1, NB>(a, st, G, b0, lane);
            step_barrier<false, DW_SLEEP>(prof);
            if (G > 0 && !ablated(a.ablate, 1)) dw_compute<2, NB>(a, sm, st, G - 1, dwid, lane);
            dw_prefetch<2, NB>(a, st, G, b0, lane);
            step_barrier<false, DW_SLEEP>(prof);
            step_barrier<false, DW_SLEEP>(prof);
        }
        // the last group: its da sits in the ring until the chains of the next trial start (after this wave's next barrier),
        // its B operands were fetched during the group -- no extra barrier-synchronised steps for the whole workgroup
        if (!ablated(a.ablate, 1)) {
            dw_compute<0, NB>(a, sm, st, n_groups - 1, dwid, lane);
            dw_compute<1, NB>(a, sm, st, n_groups - 1, dwid, lane);
            dw_compute<2, NB>(a, sm, st, n_groups - 1, dwid, lane);
        }
    }
    prof_store(a.dbg, prof);
    // accumulator tile -> slab: lane holds rows 4*(lane>>4)+r, column lane&15 of each 16x16 tile
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    const long base[3] = {a.o_w_hh1, a.o_w_ih1, a.o_w_hh0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = 32 * dwid + 16 * mt + 4 * (lane >> 4) + rr;
                    const int col = 16 * nt + (lane & 15);
                    slab[base[q] + (size_t)row * H + col] = st.acc[q][mt][nt][rr];
                }
}

// ------------------------------------------------------------------------------------------------
// dW waves, ONE trial per workgroup (the benchmark's 256 trials): the sums over time as split-bf16 products.  Every fp32 operand is
// x = hi + lo + e (hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-18 |x|) and a product is hi.hi + lo.hi + hi.lo on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (bf16 products are exact in fp32; the dropped lo.lo is <= 2^-16 of a product):
// K = 16 macro steps per instruction, 30 tiles x 3 instructions of 32 cycles per 16 steps = 180 matrix-pipe cycles per step and CU
// against 864 for the fp32 form (27 v_mfma_f32_16x16x4_f32 of 32 cycles per step).  Wave w owns rows 32 w .. + 31 of all four
// matrices (five 32 x 32 tiles: {dW_hh1 | dW_ih1} = three column tiles, {dW_hh0 | dW_ih0 | -} = two).  Operands meet in LDS as
// k-major bf16 windows of 16 steps, two per layer (one being filled while the other is read):
//   da      converted by the wave that owns the rows, one step behind the chains, from the fp32 ring (nothing changes for the chains);
//   rows    h1[t-1], in1[t], h0[t-1], x[t]: ONE dword request per lane and step by waves 0..3 (four steps ahead), written for everyone;
//   reads   ds_read_b64_tr_b16 (nsd_bf16.h): block = 4 k rows x 16 columns -> the 8 consecutive k of a column that an MFMA lane wants.
// The window of steps 16 W .. 16 W + 15 is complete behind the barrier of step 16 W + 16; its five tiles are taken at steps
// 16 W + 17 .. + 21, the last window of a trial behind the loop (its da sits in the ring until the next trial's chains start).
// ------------------------------------------------------------------------------------------------
#ifndef NSD_DW16_NOINLINE
#define DW16_INLINE __forceinline__
#else
#define DW16_INLINE __attribute__((noinline))
#endif
template <class F, int... I>
__device__ __forceinline__ void static_for(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
__device__ __forceinline__ void split1_bf16(const float v, unsigned short &hi, unsigned short &lo) {
    const unsigned h = pack_bf16x2(v, 0.f);
    hi = (unsigned short)h;
    lo = (unsigned short)pack_bf16x2(v - bf16_lo(h), 0.f);
}
template <int RS>
__device__ __forceinline__ bf16x8 window_frag(const unsigned short *win, const int c0, const int lane) {
    // lane l (G = l >> 4, i = l & 15) gives the address of k row 8 (G >> 1) + 4 e + (i >> 2), columns c0 + 16 (G & 1) + 4 (i & 3) .. + 3
    // and receives column c0 + (l & 31), k = 8 (l >> 5) + 4 e + 0..3
    const int G = lane >> 4, i = lane & 15;
    const unsigned short *p = win + (8 * (G >> 1) + (i >> 2)) * RS + c0 + 16 * (G & 1) + 4 * (i & 3);
    const s16x4 e0 = lds_read_tr16(reinterpret_cast<const bf16_t *>(p));
    const s16x4 e1 = lds_read_tr16(reinterpret_cast<const bf16_t *>(p + 4 * RS));
    return cat_tr(e0, e1);
}

__device__ __forceinline__ void split4_bf16(const f32x4 v, u32x2 &hi, u32x2 &lo) {
    hi[0] = pack_bf16x2(v[0], v[1]);
    hi[1] = pack_bf16x2(v[2], v[3]);
    lo[0] = pack_bf16x2(v[0] - bf16_lo(hi[0]), v[1] - bf16_hi(hi[0]));
    lo[1] = pack_bf16x2(v[2] - bf16_lo(hi[1]), v[3] - bf16_hi(hi[1]));
}

// Duties beside the wave's five tiles (instruction count matters more than anything else in this kernel: every role shares its SIMD
// with a recurrence): waves 0..2 the rows h1[t-1] / in1[t] / h0[t-1] and wave 3 x[t], FOUR steps per request (lane = (step, 16-byte
// piece) or (step, channel)), split and written every fourth step; waves 4, 5 the da of layer 1 / 0, every step, four columns per lane.
__device__ DW16_INLINE void dw16_role(const Lstm2BwdArgs &a_in, Smem<1> &sm, const int w_in, const int lane, const int n_steps_in) {
    const int w = __builtin_amdgcn_readfirstlane(w_in), n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);
    const Lstm2BwdArgs a = uniform_copy(a_in);
    DwWin &win = sm.win;
    const int T = a.T, B = a.B, C = a.C;
    Prof prof = prof_init(a.dbg);
    f32x16 acc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = zero16();
    // (lanes 48..63 of a 48-piece duty repeat the pieces 32..47: same address, same value, no branch)
    const int l48 = lane < 48 ? lane : lane - 16;
    // row duty of waves 0..2: lane (step s = l48 / 12, piece c4): 16 bytes; wave 3: lane (step s = lane >> 4, channel lane & 15): 4 bytes
    const float *src = w == 0 ? a.hseq1 : w == 1 ? a.in1seq : w == 2 ? a.hseq0 : a.x;
    const int rw = w == 3 ? C : H;                                  // floats per time step of the source
    const int rs = w == 3 ? lane >> 4 : l48 / 12, rp = w == 3 ? lane & 15 : l48 - 12 * rs;
    const int rl = w < 2 ? 1 : 0;                                   // layer of the window the rows go to
    const int rc = w == 3 ? H + rp : (w == 1 ? H : 0) + 4 * rp;     // first window column of this lane
    const long sbytes = (long)B * T * rw * 4;
    const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)(sbytes > 0x7fffffffL ? 0x7fffffffL : sbytes), 0x00020000);
    constexpr unsigned DROP = 0x80000000u;
    // layer-0 dropout multipliers drawn in the kernel: one value per dW lane and chunk, a chunk ahead like the loader (as before)
    const int L = w * 64 + lane, mk_k = L / H, mk_j = L - mk_k * H;
    auto gen_mask = [&](const int chunk, const int b) {
        const int t = T + 1 + dl0(1) - (chunk * CHUNK + mk_k);
        if (t >= 0 && t < T)
            sm.stage[chunk & 1][0][0][mk_k][240 + mk_j] =
                nsd_rand_u32(a.rng.seed, a.rng.base, ((uint64_t)b * T + t) * H + mk_j) < a.rng.thr_lstm ? 0.f : a.rng.keep_lstm;
    };
    bf16x8 ah, al;                                                  // this wave's rows of a window, kept over the window's tiles of one layer
    auto a_frags = [&](const int W, const int layer) {
        ah = window_frag<ARS>(&win.wa[layer][0][W & 1][0][0], 32 * w, lane);
        al = window_frag<ARS>(&win.wa[layer][1][W & 1][0][0], 32 * w, lane);
    };
    auto tile = [&](const int W, auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int layer = q < 3 ? 1 : 0, ni = q < 3 ? q : q - 3;
        if (q == 0 || q == 3) a_frags(W, layer);
        const bf16x8 bh = window_frag<BRS>(&win.wb[layer][0][W & 1][0][0], 32 * ni, lane);
        const bf16x8 bl = window_frag<BRS>(&win.wb[layer][1][W & 1][0][0], 32 * ni, lane);
        if (!ablated(a.ablate, 1)) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[q], 0, 0, 0);
        }
    };
    // the row windows start as zeros: a row that is never written (the second half of a trial's last window when the step count is
    // 8 mod 16) meets zero da -- it has to be finite
    {
        u32x4 *z = reinterpret_cast<u32x4 *>(&win.wb[0][0][0][0][0]);
        constexpr int NZ = (int)(sizeof(win.wb) / 16);
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (int e = w * 64 + lane; e < NZ; e += 6 * 64) z[e] = zero;
    }
    const int ngrp = B;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b = grp;
        if (a.rng.on) gen_mask(0, b);
        // rows of macro steps m .. m + 3 (this lane: m + rs); out of range -> zeros (switched off at the ADDRESS)
        auto request = [&](const int m) -> f32x4 {
            const int t = (rl == 1 ? T - 1 : T + 1 + dl0(1)) - m - rs;
            const int tt = (w == 1 || w == 3) ? t : t - 1;
            const bool ok = t >= 0 && t < T && tt >= 0 && (w != 3 || rp < C);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w == 3) {
                const unsigned off = ok ? (unsigned)((((size_t)b * T + tt) * C + rp) * 4) : DROP;
                v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_s, (int)off, 0, 0));
            } else {
                const unsigned off = ok ? (unsigned)((((size_t)b * T + tt) * H + 4 * rp) * 4) : DROP;
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_s, (int)off, 0, 0));
            }
            return v;
        };
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (w < 4) bq = request(0);
        step_barrier<false>(prof);      // pairs with the stage-initialisation barrier of the other roles
        auto half = [&](const int m0, auto phc) {
            constexpr int PH = decltype(phc)::value;
            static_for([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int m = m0 + k;
                constexpr int p = PH + k;                            // m & 15
                const int wi = (m >> 4) & 1;
                if (w >= 4 && m >= 1) {                              // da of macro step m - 1 (written by the chains at that step)
                    const int mm = m - 1, cl = 5 - w;
                    const int t = cl == 1 ? T - 1 - mm : T + 1 + dl0(1) - mm;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(&sm.ring[cl][mm & (RING - 1)][0][4 * l48]);
                    if (!(t >= 0 && t < T)) v = f32x4{0.f, 0.f, 0.f, 0.f};       // (the chains do not write on inactive steps)
                    u32x2 hi, lo;
                    split4_bf16(v, hi, lo);
                    *reinterpret_cast<u32x2 *>(&win.wa[cl][0][(mm >> 4) & 1][mm & 15][4 * l48]) = hi;
                    *reinterpret_cast<u32x2 *>(&win.wa[cl][1][(mm >> 4) & 1][mm & 15][4 * l48]) = lo;
                }
                if (w < 4 && (k & 3) == 0) {                         // the rows of steps m .. m + 3, requested four steps ago
                    if (w == 3) {
                        unsigned short hi, lo;
                        split1_bf16(bq[0], hi, lo);
                        win.wb[0][0][wi][p + rs][rc] = hi;
                        win.wb[0][1][wi][p + rs][rc] = lo;
                    } else {
                        u32x2 hi, lo;
                        split4_bf16(bq, hi, lo);
                        *reinterpret_cast<u32x2 *>(&win.wb[rl][0][wi][p + rs][rc]) = hi;
                        *reinterpret_cast<u32x2 *>(&win.wb[rl][1][wi][p + rs][rc]) = lo;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bq = request(m + 4);
                }
                if (a.rng.on && k == 1) gen_mask((m >> 3) + 1, b);  // second step of a chunk: the old records are dead
                if (p >= 1 && p <= 5 && m >= 17) tile((m - 17) >> 4, std::integral_constant<int, (p >= 1 && p <= 5) ? p - 1 : 0>{});
                step_barrier<false, DW_SLEEP>(prof);
            }, std::make_integer_sequence<int, 8>{});
        };
        for (int m0 = 0; m0 < n_steps; m0 += 16) {
            half(m0, std::integral_constant<int, 0>{});
            if (m0 + 8 < n_steps) half(m0 + 8, std::integral_constant<int, 8>{});
        }
        // behind the loop: the last window's tiles.  Its row of the last macro step (always an inactive one: n_steps >= T + 3) and, when
        // the step count is 8 mod 16, its second half get zeros in THIS wave's da columns.
        {
            const int Wl = (n_steps - 1) >> 4, cl = lane >> 5, cc = 32 * w + (lane & 31);
            const int k0 = (n_steps - 1) & 15;
            for (int k = k0; k < 16; ++k) {
                win.wa[cl][0][Wl & 1][k][cc] = 0;
                win.wa[cl][1][Wl & 1][k][cc] = 0;
            }
            tile(Wl, std::integral_constant<int, 0>{});
            tile(Wl, std::integral_constant<int, 1>{});
            tile(Wl, std::integral_constant<int, 2>{});
            tile(Wl, std::integral_constant<int, 3>{});
            tile(Wl, std::integral_constant<int, 4>{});
        }
    }
    prof_store(a.dbg, prof);
    // accumulator tile -> slab: register r of lane l = dW[row 32 w + mfma32_row(r, l)][window column 32 ni + (l & 31)]
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int layer = q < 3 ? 1 : 0, c = 32 * (q < 3 ? q : q - 3) + (lane & 31);
        const long base = layer == 1 ? (c < H ? a.o_w_hh1 : a.o_w_ih1) : (c < H ? a.o_w_hh0 : a.o_w_ih0);
        const int ld = (layer == 0 && c >= H) ? C : H, col = c < H ? c : c - H;
        const bool okc = layer == 1 || c < H + C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * w + mfma32_row(r, lane);
            if (okc) slab[base + (size_t)row * ld + col] = acc[q][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// loader wave: LDS-DMA stream of the saved activations, one chunk ahead of the chain.
// A chunk image is 18 wave-wide 1 KB pieces per trial (+ one 256 B piece of x rows).  What a lane copies
// in piece q never changes except for the time index, so the address recipe is decoded ONCE per lane
// (base pointer, bytes per time step, t of chunk 0, lowest valid t) and a piece costs a handful of VALU
// instructions + the DMA issue: the loader must never be the wave the step barrier waits for.
// ------------------------------------------------------------------------------------------------
constexpr int NQ = STAGE_F4 / 64;        // 18

struct LdDesc {
    const char *base;     // address of (trial 0, t = 0) for this lane's 16 bytes
    int row_bytes;        // bytes per time step of the source array (0 = lane never copies)
    int t0;               // time index in chunk 0
};

template <int NB>
__device__ __forceinline__ void loader_decode(const Lstm2BwdArgs &a, const int lane, LdDesc (&d)[NQ]) {
    const int T = a.T;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = q * 64 + lane;
        const int layer = e / (CHUNK * REC4);
        const int rem = e - layer * (CHUNK * REC4);
        const int k = rem / REC4, w = rem - k * REC4;
        d[q].t0 = layer == 1 ? (T - 1 - k) : (T + 1 + dl0(NB) - k);
        d[q].row_bytes = 0;
        d[q].base = nullptr;
        if (w < 48) {
            d[q].base = (const char *)((layer == 0 ? a.gact0 : a.gact1) + w * 4); d[q].row_bytes = H * 16;
        } else if (w < 60) {     // c[t-1]; at t = 0 this reads the 192 bytes in front of the trial's c rows, which
                                 // lie inside the workspace (hseq precedes cseq) and are ignored by the chain
            d[q].base = (const char *)((layer == 0 ? a.cseq0 : a.cseq1) + (w - 48) * 4) - H * 4; d[q].row_bytes = H * 4;
        } else if (layer == 0) {
            if (a.mask && !a.rng.on) { d[q].base = (const char *)(a.mask + (w - 60) * 4); d[q].row_bytes = H * 4; }
        } else if (w == 60) {
            d[q].base = (const char *)a.dsc_pack; d[q].row_bytes = 16;
        }
    }
}

template <int NB, int Q0, int Q1>
__device__ __forceinline__ void loader_issue(const Lstm2BwdArgs &a, Smem<NB> &sm, const LdDesc (&d)[NQ], const int chunk,
                                             const int buf, const int b0) {
    const int T = a.T;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int b = b0 + n;
        if (b < a.B) {
#pragma unroll
            for (int q = Q0; q < Q1; ++q) {
                const int t = d[q].t0 - CHUNK * chunk;
                if (d[q].row_bytes != 0 && (unsigned)t < (unsigned)T) {
                    const char *src = d[q].base + (size_t)((unsigned)(b * T + t)) * (unsigned)d[q].row_bytes;
                    // LDS destination = wave-uniform base + lane*16: the chunk image is linear in e = q*64 + lane
                    __builtin_amdgcn_global_load_lds((const void *)src,
                                                     (__attribute__((address_space(3))) void *)(&sm.stage[buf][n][0][0][0] + q * 256),
                                                     16, 0, 0);
                }
            }
        }
    }
}

// x rows for the x1 waves (dW_ih0): lane (k = lane>>3, ch = lane&7) moves x[T+2-mm][ch] of macro step
// mm = chunk*8+k with a 4-byte LDS-DMA (no ordinary load in this wave: hipcc drains every DMA in flight
// before an ordinary VMEM load, which would stall the step barrier)
template <int NB>
__device__ __forceinline__ void loader_issue_x(const Lstm2BwdArgs &a, Smem<NB> &sm, const int chunk, const int buf,
                                               const int b0, const int lane) {
    const int T = a.T, k = lane >> 3, ch = lane & 7;
    const int tx = T + 2 - (chunk * CHUNK + k);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int b = b0 + n;
        if (b < a.B && tx >= 0 && tx < T && ch < a.C)
            __builtin_amdgcn_global_load_lds((const void *)(a.x + ((size_t)b * T + tx) * a.C + ch),
                                             (__attribute__((address_space(3))) void *)&sm.xst[buf][n][0][0], 4, 0, 0);
    }
}

template <int NB>
__device__ __forceinline__ void loader_role(const Lstm2BwdArgs &a, Smem<NB> &sm, const int lane, const int n_steps) {
    LdDesc d[NQ];
    loader_decode<NB>(a, lane, d);
    Prof prof = prof_init(a.dbg);
    const int ngrp = (a.B + NB - 1) / NB;
    const int T = a.T;
    const int u = lane < H ? lane : lane - 16;                      // prep of a trial's first step: this lane's unit (lanes 48..63 repeat units 32..47)
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float dpu = 0.f, awu = 0.f, cT1[2] = {0.f, 0.f};
        if constexpr (NB == 1) {                                    // (ordinary loads only in front of the trial's first DMA)
            dpu = a.dpooled[(size_t)b0 * H + u];
            awu = a.attn_w[u];
            cT1[1] = a.cseq1[((size_t)b0 * T + (T - 1)) * H + u];
            cT1[0] = a.cseq0[((size_t)b0 * T + (T - 1)) * H + u];
        }
        loader_issue<NB, 0, NQ>(a, sm, d, 0, 0, b0);
        if (NB == 2) loader_issue_x<NB>(a, sm, 0, 0, b0, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (NB == 1) { prep_layer(a, sm, 0, 1, u, dpu, awu, cT1[1]); prep_layer(a, sm, 0, 0, u, dpu, awu, cT1[0]); }
        step_barrier<true>(prof);
        for (int m0 = 0; m0 < n_steps; m0 += CHUNK) {
            const int chunk = m0 / CHUNK, nb = (chunk + 1) & 1;
            const bool on = !ablated(a.ablate, 16);
            if constexpr (NB == 1) {
                // next chunk: 4 pieces per step during steps 0..4; all landed before step 6 ends -- the x1 waves prepare the factors of the
                // chunk's first step during step 7
                if (on) loader_issue<NB, 0, 4>(a, sm, d, chunk + 1, nb, b0);
                step_barrier<true, LD_SLEEP>(prof);
                if (on) loader_issue<NB, 4, 8>(a, sm, d, chunk + 1, nb, b0);
                step_barrier<true, LD_SLEEP>(prof);
                if (on) loader_issue<NB, 8, 12>(a, sm, d, chunk + 1, nb, b0);
                step_barrier<true, LD_SLEEP>(prof);
                if (on) loader_issue<NB, 12, 16>(a, sm, d, chunk + 1, nb, b0);
                step_barrier<true, LD_SLEEP>(prof);
                if (on) loader_issue<NB, 16, 18>(a, sm, d, chunk + 1, nb, b0);
                step_barrier<true, LD_SLEEP>(prof);
                step_barrier<true, LD_SLEEP>(prof);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                step_barrier<true, LD_SLEEP>(prof);
                step_barrier<true, LD_SLEEP>(prof);
            } else {
            // next chunk: 3 pieces per step during steps 0..5 (+ x rows at step 0); all landed before step 7 ends
            if (on) { loader_issue<NB, 0, 3>(a, sm, d, chunk + 1, nb, b0); if (NB == 2) loader_issue_x<NB>(a, sm, chunk + 1, nb, b0, lane); }
            step_barrier<true, LD_SLEEP>(prof);
            if (on) loader_issue<NB, 3, 6>(a, sm, d, chunk + 1, nb, b0);
            step_barrier<true, LD_SLEEP>(prof);
            if (on) loader_issue<NB, 6, 9>(a, sm, d, chunk + 1, nb, b0);
            step_barrier<true, LD_SLEEP>(prof);
            if (on) loader_issue<NB, 9, 12>(a, sm, d, chunk + 1, nb, b0);
            step_barrier<true, LD_SLEEP>(prof);
            if (on) loader_issue<NB, 12, 15>(a, sm, d, chunk + 1, nb, b0);
            step_barrier<true, LD_SLEEP>(prof);
            if (on) loader_issue<NB, 15, 18>(a, sm, d, chunk + 1, nb, b0);
            step_barrier<true, LD_SLEEP>(prof);
            step_barrier<true, LD_SLEEP>(prof);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            step_barrier<true, LD_SLEEP>(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
}

template <int NB>
__global__ __launch_bounds__(NTHREADS) void lstm2_bwd48_kernel(Lstm2BwdArgs a) {
    __shared__ __align__(16) Smem<NB> sm;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // every role runs the same number of barriers: 4 per 4-step group.  The last chain step is macro step T+1 and the x1
    // waves need T+2 (group (T+2)/4); rounded up to even because the loader walks whole 8-step chunks.  (The dW waves
    // finish the last group after the loop, on their own.)
    const int n_groups = (((a.T + 2 + dl0(NB)) / 4 + 1) + 1) & ~1;
    const int n_steps = 4 * n_groups;
    // issue priority follows the critical path: the two recurrences first, then the hand-off to layer 0
    if constexpr (NB == 1) {
#ifndef NSD_B48_MAP
#define NSD_B48_MAP 2
#endif
        // Roles by SIMD (waves w and w + 4 share one; a SIMD's instructions per step are what bounds the step -- per-wave stamps:
        // the younger recurrence of a SIMD that carries two runs at 1 065 cycles of work against 840 -- so the SIMDs with TWO
        // recurrences get the four light dW waves (rows duty), the other two the x1 waves, the da converters and the loader):
        //   SIMD 0: chain1_0 chain0_1 dW0 dW2 | SIMD 1: chain1_1 chain0_2 dW1 dW3 | SIMD 2: chain1_2 x1_0 x1_2 dW5 | SIMD 3: chain0_0 x1_1 dW4 loader
#if NSD_B48_MAP == 2
        //   SIMD 0: chain1_0 chain0_1 x1_2 dW0 | SIMD 1: chain1_1 chain0_2 dW4 dW1 | SIMD 2: chain1_2 x1_0 dW2 dW5 | SIMD 3: chain0_0 x1_1 dW3 loader
        constexpr int ROLE[16] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4};
        constexpr int PART[16] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 4, 2, 3, 0, 1, 5, 0};
#else
        constexpr int ROLE[16] = {0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 2, 3, 3, 3, 3, 4};        // 0 chain1, 1 chain0, 2 x1, 3 dW, 4 loader
        constexpr int PART[16] = {0, 1, 2, 0, 1, 2, 0, 1, 0, 1, 2, 4, 2, 3, 5, 0};
#endif
        const int lane = tid & 63;
        if (NSD_B48_MAP >= 1) {
            const int role = ROLE[wave], part = PART[wave];
#ifndef NSD_B48_PC
#define NSD_B48_PC 2
#endif
#ifndef NSD_B48_PX
#define NSD_B48_PX 3
#endif
            if (role == 0)      { __builtin_amdgcn_s_setprio(NSD_B48_PC); chain_role<NB>(a, sm, 1, 64 * part + lane, n_steps); }
            else if (role == 1) { __builtin_amdgcn_s_setprio(NSD_B48_PC); chain_role<NB>(a, sm, 0, 64 * part + lane, n_steps); }
            else if (role == 2) { __builtin_amdgcn_s_setprio(NSD_B48_PX); if (NSD_B48_X1M) x1m_role(a, sm, part, lane, n_steps); else x1_role<NB>(a, sm, 64 * part + lane, n_steps); }
            else if (role == 3) dw16_role(a, sm, part, lane, n_steps);
            else                { __builtin_amdgcn_s_setprio(1); loader_role<NB>(a, sm, lane, n_steps); }
            return;
        }
    }
    if (wave < 3)       { __builtin_amdgcn_s_setprio(3); chain_role<NB>(a, sm, 1, tid, n_steps); }
    else if (wave < 6)  { __builtin_amdgcn_s_setprio(3); chain_role<NB>(a, sm, 0, tid - 192, n_steps); }
    else if (wave < 9)  { __builtin_amdgcn_s_setprio(2); x1_role<NB>(a, sm, tid - 384, n_steps); }
    else if (wave < 15) { if constexpr (NB == 1) dw16_role(a, sm, wave - 9, tid & 63, n_steps); else dw_role<NB>(a, sm, wave - 9, tid & 63, n_groups); }
    else                { __builtin_amdgcn_s_setprio(1); loader_role<NB>(a, sm, tid & 63, n_steps); }
}

}  // namespace

int nsd_lstm2_bwd48_launch(const Lstm2BwdArgs &a, int nb, int grid, hipStream_t st) {
    // the one-trial instantiation's dW waves address the saved rows through buffer descriptors with 32-bit offsets (0x80000000 = "switched
    // off"): a batch whose [B][T][H] arrays reach 2 GB takes the two-trial instantiation (64-bit addresses; any grid)
    if (nb == 1 && (long)a.B * a.T * H * 4 >= 0x7fffffffL) nb = 2;
    if (a.da0_out && nb != 1) { nsd_set_error("lstm2_bwd48: the input gradient needs the one-trial instantiation (B * T * H * 4 < 2 GB)"); return NSD_E_INVALID; }
    switch (nb) {
    case 1: hipLaunchKernelGGL((lstm2_bwd48_kernel<1>), dim3(grid), dim3(NTHREADS), 0, st, a); break;
    case 2: hipLaunchKernelGGL((lstm2_bwd48_kernel<2>), dim3(grid), dim3(NTHREADS), 0, st, a); break;
    default: nsd_set_error("lstm2_bwd48: NB=%d not built (register / LDS budget)", nb); return NSD_E_INVALID;
    }
    NSD_CHECK_LAUNCH("lstm2_bwd48");
    return NSD_OK;
}
