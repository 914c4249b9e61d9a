// nsd_lstm2_bwd48.hip -- BPTT of the two-layer H=48 LSTM, role-split workgroup (gfx950).
//
// Replaces autograd through self.lstm(x) (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:34) for the reference
// model shape (H=48, L=2, C<=8).  One 1024-thread workgroup (16 waves) owns NB trials and walks time
// backwards with ONE barrier per step; the waves have different jobs so that only the true recurrence
// is on the critical path and every register array stays small:
//
//   waves 0-2   "chain 1"  layer-1 cell backward: dh_rec = W_hh1^T da1[t+1] (48 FMA/lane as 24 v_pk_fma_f32, weights
//                          in VGPRs, 12 operands per lane from LDS, DPP reduction over 16 lanes) then the
//                          element-wise cell backward for step t -> da1[t] into an LDS ring.
//   waves 3-5   "chain 0"  the same for layer 0, two steps behind layer 1.
//   waves 6-8   "x1"       d_in1[t] = W_ih1^T da1[t] (the gradient handed to layer 0) and dW_ih0 (K=8, VALU).
//   wave  15    "loader" LDS-DMA stream of the saved activations, one 8-step chunk ahead (see below).
//   waves 9-14  "dW"       weight gradients dW_hh1, dW_ih1, dW_hh0 = sum_t da[t] (x) operand[t] as
//                          v_mfma_f32_16x16x4_f32 with K = 4 time steps per instruction: A = da tiles from
//                          the LDS ring, B = h / input rows straight from the activations saved in HBM
//                          (prefetched one 4-step group ahead).  The matrix pipe is otherwise idle, so
//                          these 27 648 MAC/step cost the chain nothing.
//
// LDS: da ring [2 layers][8 steps][NB][192] + d_in1 double buffer.  HBM traffic = saved activations read
// once (gates, c, h, in1, x) + one slab of partial gradients per workgroup at the end.
#include "nsd_args.h"
#include "nsd_prof.h"

namespace {

constexpr int H = 48;
constexpr int G4 = 192;
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
template <int NB>
struct Smem {
    float ring[2][RING][NB][G4];           // [layer][macro step % RING][trial][gate*48+unit]
    float din1[2][NB][H];
    float stage[2][NB][2][CHUNK][REC];     // [buffer][trial][layer][step in chunk][record]  (linear per trial)
    float xst[2][NB][CHUNK][8];            // x[T+2-m] rows for the x1 waves (dW_ih0), same chunking
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
            dpj[n] = (layer == 1 && b < B) ? a.dpooled[(size_t)b * H + j] : 0.f;
            ct[n] = (b < B) ? cseq[((size_t)b * T + (T - 1)) * H + j] : 0.f;   // c[T-1] of the first step
        }
        step_barrier<false>(prof);      // chunk 0 of the stage has been written by the loader wave

        // unrolled by the ring length (== stage chunk): every LDS offset of a step is an immediate
        for (int m0 = 0; m0 < n_steps; m0 += CHUNK) {
            const int sb = (m0 / CHUNK) & 1;
#pragma unroll
            for (int k = 0; k < CHUNK; ++k) {
                const int m = m0 + k;
                const int t = layer == 1 ? (T - 1 - m) : (T + 1 - m);
                const bool active = (t >= 0 && t < T);
                const bool prev_active = (t + 1 >= 0 && t + 1 < T);
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
                    const float2 xv = *reinterpret_cast<const float2 *>(&sm.xst[sb][n][k][2 * s]);
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
                    if (t0p >= 0 && t0p < T && b0 + n < B) {
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
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (c0 < C) slab[a.o_w_ih0 + (size_t)(g * H + j) * C + c0] = dWih0[g][0];
        if (c1 < C) slab[a.o_w_ih0 + (size_t)(g * H + j) * C + c1] = dWih0[g][1];
    }
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
        d[q].t0 = layer == 1 ? (T - 1 - k) : (T + 1 - k);
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
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        loader_issue<NB, 0, NQ>(a, sm, d, 0, 0, b0);
        loader_issue_x<NB>(a, sm, 0, 0, b0, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step_barrier<true>(prof);
        for (int m0 = 0; m0 < n_steps; m0 += CHUNK) {
            const int chunk = m0 / CHUNK, nb = (chunk + 1) & 1;
            const bool on = !ablated(a.ablate, 16);
            // next chunk: 3 pieces per step during steps 0..5 (+ x rows at step 0); all landed before step 7 ends
            if (on) { loader_issue<NB, 0, 3>(a, sm, d, chunk + 1, nb, b0); loader_issue_x<NB>(a, sm, chunk + 1, nb, b0, lane); }
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
    const int n_groups = (((a.T + 2) / 4 + 1) + 1) & ~1;
    const int n_steps = 4 * n_groups;
    // issue priority follows the critical path: the two recurrences first, then the hand-off to layer 0
    if (wave < 3)       { __builtin_amdgcn_s_setprio(3); chain_role<NB>(a, sm, 1, tid, n_steps); }
    else if (wave < 6)  { __builtin_amdgcn_s_setprio(3); chain_role<NB>(a, sm, 0, tid - 192, n_steps); }
    else if (wave < 9)  { __builtin_amdgcn_s_setprio(2); x1_role<NB>(a, sm, tid - 384, n_steps); }
    else if (wave < 15) dw_role<NB>(a, sm, wave - 9, tid & 63, n_groups);
    else                { __builtin_amdgcn_s_setprio(1); loader_role<NB>(a, sm, tid & 63, n_steps); }
}

}  // namespace

int nsd_lstm2_bwd48_launch(const Lstm2BwdArgs &a, int nb, int grid, hipStream_t st) {
    switch (nb) {
    case 1: hipLaunchKernelGGL((lstm2_bwd48_kernel<1>), dim3(grid), dim3(NTHREADS), 0, st, a); break;
    case 2: hipLaunchKernelGGL((lstm2_bwd48_kernel<2>), dim3(grid), dim3(NTHREADS), 0, st, a); break;
    default: nsd_set_error("lstm2_bwd48: NB=%d not built (register / LDS budget)", nb); return NSD_E_INVALID;
    }
    NSD_CHECK_LAUNCH("lstm2_bwd48");
    return NSD_OK;
}
