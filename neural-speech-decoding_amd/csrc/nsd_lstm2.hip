// nsd_lstm2.hip -- fused, register-resident two-layer LSTM for gfx950 (fast path: L==2, H in {32,48,64}, C<=8).
//
// Replaces self.lstm(x) of the reference (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34), i.e.
// torch.nn.LSTM(batch_first=True) semantics: gate order i,f,g,o, two bias vectors, zero initial state,
// inter-layer dropout on the output of layer 0.
//
// Design (see DESIGN.md "lstm2"):
//   * one workgroup owns NB trials and walks the T steps on-chip; nothing but x (in) and the saved
//     activations (out, train mode) touches HBM.  All weights of both layers live in VGPRs for the whole
//     kernel (29 184 floats spread over 8H threads).
//   * thread (layer, unit j, k-slice s): holds, for the 4 gates of unit j, the slice s of the weight rows;
//     per step it does 4 x (slice) FMAs against operands broadcast from LDS, a 2-stage DPP quad
//     reduction, and lane s of the quad evaluates gate s.  No LDS round trip inside a step.
//   * the two layers run skewed by one step in different waves (layer 1 handles t-1 while layer 0
//     handles t), so there is exactly ONE workgroup barrier per time step.
#include <stdlib.h>
#include "nsd_args.h"
#include "nsd_diag.h"


template <int H, int NB>
__global__ __launch_bounds__(8 * H) void lstm2_fwd_kernel(Lstm2FwdArgs a) {
    constexpr int KS = H / 4;        // k-slice of a hidden vector owned by one lane of the quad
    constexpr int NT = 8 * H;
    constexpr int CP = 8, CS = 2;    // x channels padded to 8: 2 per k-slice
    constexpr int XCH = 32;          // time steps per staged x chunk
    constexpr int XE = NB * XCH * CP;
    constexpr int XPT = (XE + NT - 1) / NT;
    static_assert(KS % 4 == 0, "H must be a multiple of 16");

    __shared__ __align__(16) float xs[2][NB][XCH][CP];
    __shared__ __align__(16) float ms[2][NB][XCH][H];   // inter-layer dropout multipliers, staged like x
    __shared__ __align__(16) float h0s[2][NB][H];   // h of layer 0 (recurrent operand)
    __shared__ __align__(16) float h0m[2][NB][H];   // layer-0 output after dropout (layer-1 input)
    __shared__ __align__(16) float h1s[2][NB][H];

    const int tid = threadIdx.x;
    const int layer = __builtin_amdgcn_readfirstlane(tid / (4 * H));   // wave-uniform: 4H is a multiple of 64
    const int r = tid - layer * 4 * H;
    const int j = r >> 2, s = r & 3;
    const int T = a.T, B = a.B, C = a.C;

    // ---- weights -> registers (once) -----------------------------------------------------------
    float wa[4][KS], wh[4][KS];
    float bias;
    if (layer == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = g * H + j;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int ch = s * CS + k;
                wa[g][k] = (k < CS && ch < C) ? a.w_ih0[(size_t)row * C + ch] : 0.f;
                wh[g][k] = a.w_hh0[(size_t)row * H + s * KS + k];
            }
        }
        bias = a.b_ih0[s * H + j] + a.b_hh0[s * H + j];
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = g * H + j;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                wa[g][k] = a.w_ih1[(size_t)row * H + s * KS + k];
                wh[g][k] = a.w_hh1[(size_t)row * H + s * KS + k];
            }
        }
        bias = a.b_ih1[s * H + j] + a.b_hh1[s * H + j];
    }
    const float ga = (s == 2) ? 2.f : 1.f;
    const float gb = (s == 2) ? -2.f * LOG2E_F : -LOG2E_F;
    const float gc = (s == 2) ? -1.f : 0.f;

    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float c[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) c[n] = 0.f;
        for (int e = tid; e < 2 * NB * H; e += NT) {
            (&h0s[0][0][0])[e] = 0.f;
            (&h0m[0][0][0])[e] = 0.f;
            (&h1s[0][0][0])[e] = 0.f;
        }
        // x chunk 0 (and the dropout multipliers of chunk 0) straight into LDS
        for (int e = tid; e < XE; e += NT) {
            const int n = e / (XCH * CP), tl = (e / CP) % XCH, ch = e % CP;
            const int b = b0 + n;
            xs[0][n][tl][ch] = (b < B && tl < T && ch < C) ? a.x[((size_t)b * T + tl) * C + ch] : 0.f;
        }
        // the mask chunk [XCH][H] of a trial is XCH*H/4 = NT float4: one per thread and trial
        auto mask_f4 = [&](int t0, int n) -> float4 {
            const int tl = tid / (H / 4), q = tid - tl * (H / 4);
            const int b = b0 + n, t = t0 + tl;
            if (a.mask && b < B && t < T) return *reinterpret_cast<const float4 *>(a.mask + ((size_t)b * T + t) * H + 4 * q);
            return make_float4(1.f, 1.f, 1.f, 1.f);
        };
#pragma unroll
        for (int n = 0; n < NB; ++n) *reinterpret_cast<float4 *>(&ms[0][n][0][0] + 4 * tid) = mask_f4(0, n);
        __syncthreads();

        for (int m0 = 0; m0 <= T; m0 += XCH) {
            // prefetch the next x / mask chunk into registers; written to LDS at the end of this chunk.  The
            // recurrence itself never waits for HBM: a dependent global load per step would cost a full
            // memory latency (~1 us) per step.
            float xr[XPT];
            float4 mr[NB];
#pragma unroll
            for (int q = 0; q < XPT; ++q) {
                const int e = tid + q * NT;
                const int n = e / (XCH * CP), tl = (e / CP) % XCH, ch = e % CP;
                const int b = b0 + n, t = m0 + XCH + tl;
                xr[q] = (e < XE && b < B && t < T && ch < C) ? a.x[((size_t)b * T + t) * C + ch] : 0.f;
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) mr[n] = mask_f4(m0 + XCH, n);
            const int cb = (m0 / XCH) & 1;
            for (int k = 0; k < XCH; ++k) {
                const int m = m0 + k;
                if (m > T) break;
                const int cur = m & 1, prv = cur ^ 1;
                if (layer == 0) {
                    if (m < T) {
                        const int t = m;
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            const int b = b0 + n;
                            const bool valid = b < B;
                            const size_t idx = ((size_t)(valid ? b : 0) * T + t) * H + j;
                            const float mk = ms[cb][n][k][j];
                            const float2 xv = *reinterpret_cast<const float2 *>(&xs[cb][n][k][s * CS]);
                            float hv[KS];
#pragma unroll
                            for (int q = 0; q < KS / 4; ++q) {
                                const float4 v = *reinterpret_cast<const float4 *>(&h0s[prv][n][s * KS + 4 * q]);
                                hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
                            }
                            float acc[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float v = wa[g][0] * xv.x;
                                v = fmaf(wa[g][1], xv.y, v);
#pragma unroll
                                for (int kk = 0; kk < KS; ++kk) v = fmaf(wh[g][kk], hv[kk], v);
                                acc[g] = quad_sum(v);
                            }
                            const float pre = (s == 0 ? acc[0] : s == 1 ? acc[1] : s == 2 ? acc[2] : acc[3]) + bias;
                            const float act = gate_act(pre, ga, gb, gc);
                            const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act);
                            const float gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                            c[n] = fmaf(fg, c[n], ig * gg);
                            const float h = og * fast_tanh(c[n]);
                            const float hm = h * mk;
                            if (s == 0) h0s[cur][n][j] = h;
                            if (s == 1) h0m[cur][n][j] = hm;
                            if (valid) {
                                if (a.gact0) a.gact0[idx * 4 + s] = act;
                                if (s == 0 && a.hseq0) a.hseq0[idx] = h;
                                if (s == 1 && a.cseq0) a.cseq0[idx] = c[n];
                                if (s == 2 && a.inseq) a.inseq[idx] = hm;
                            }
                        }
                    }
                } else {
                    if (m >= 1) {
                        const int t = m - 1;
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            const int b = b0 + n;
                            const bool valid = b < B;
                            const size_t idx = ((size_t)(valid ? b : 0) * T + t) * H + j;
                            float iv[KS], hv[KS];
#pragma unroll
                            for (int q = 0; q < KS / 4; ++q) {
                                const float4 u = *reinterpret_cast<const float4 *>(&h0m[prv][n][s * KS + 4 * q]);
                                iv[4 * q] = u.x; iv[4 * q + 1] = u.y; iv[4 * q + 2] = u.z; iv[4 * q + 3] = u.w;
                                const float4 v = *reinterpret_cast<const float4 *>(&h1s[cur][n][s * KS + 4 * q]);
                                hv[4 * q] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
                            }
                            float acc[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                float v = wa[g][0] * iv[0];
#pragma unroll
                                for (int kk = 1; kk < KS; ++kk) v = fmaf(wa[g][kk], iv[kk], v);
#pragma unroll
                                for (int kk = 0; kk < KS; ++kk) v = fmaf(wh[g][kk], hv[kk], v);
                                acc[g] = quad_sum(v);
                            }
                            const float pre = (s == 0 ? acc[0] : s == 1 ? acc[1] : s == 2 ? acc[2] : acc[3]) + bias;
                            const float act = gate_act(pre, ga, gb, gc);
                            const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act);
                            const float gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                            c[n] = fmaf(fg, c[n], ig * gg);
                            const float h = og * fast_tanh(c[n]);
                            if (s == 0) h1s[prv][n][j] = h;
                            if (valid) {
                                if (a.gact1) a.gact1[idx * 4 + s] = act;
                                if (s == 0 && a.hseq1) a.hseq1[idx] = h;
                                if (s == 1 && a.cseq1) a.cseq1[idx] = c[n];
                                if (s == 2 && a.top) a.top[idx] = a.residual ? h + h0m[prv][n][j] : h;
                            }
                        }
                    }
                }
                if (k == XCH - 1) {
#pragma unroll
                    for (int q = 0; q < XPT; ++q) {
                        const int e = tid + q * NT;
                        if (e < XE) (&xs[cb ^ 1][0][0][0])[e] = xr[q];
                    }
#pragma unroll
                    for (int n = 0; n < NB; ++n) *reinterpret_cast<float4 *>(&ms[cb ^ 1][n][0][0] + 4 * tid) = mr[n];
                }
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward (BPTT).  Same thread mapping.  Layer 1 handles t = T-1-m at macro step m, layer 0 lags
// by two steps (t = T+1-m) so that its incoming gradient, produced by layer 1's transposed
// mat-vec, crosses exactly one barrier.  One barrier per macro step:
//   A-part : element-wise cell backward for the own unit (registers; recurrent dh arrives by quad
//            reduction in the same quad) -> da (pre-activation grads) to LDS
//   barrier
//   B-part : transposed mat-vecs  dh_rec[k] = sum_rows W_hh[row,k] da[row],  d_in[k] = ... W_ih ...
//            (lane (k,g) owns column k of gate block g) + weight-gradient outer products in registers.
// ---------------------------------------------------------------------------------------------

template <int H, int NB>
__global__ __launch_bounds__(8 * H) void lstm2_bwd_kernel(Lstm2BwdArgs a) {
    constexpr int KS = H / 4;
    constexpr int NT = 8 * H;
    constexpr int CP = 8, CS = 2;

    __shared__ __align__(16) float dab[2][2][NB][4][H];   // [parity][layer][trial][gate][unit]
    __shared__ __align__(16) float din1[2][NB][H];        // layer-1 d_in -> layer 0 (two steps later)
    __shared__ __align__(16) float hp0[2][NB][H];         // staged operands of the outer products
    __shared__ __align__(16) float hp1[2][NB][H];
    __shared__ __align__(16) float in1[2][NB][H];
    __shared__ __align__(16) float xb[2][NB][CP];

    const int tid = threadIdx.x;
    const int layer = __builtin_amdgcn_readfirstlane(tid / (4 * H));   // wave-uniform: 4H is a multiple of 64
    const int r = tid - layer * 4 * H;
    const int j = r >> 2, s = r & 3;
    const int T = a.T, B = a.B, C = a.C;

    // whT: column j of gate block s of the own layer's W_hh (transposed mat-vec operand).
    // u  : layer 1 -> column j of gate block s of W_ih1 (read-only)
    //      layer 0 -> dW_hh1 accumulators u[g*KS+k] (rows g*H+j, columns of slice s); 4*KS == H
    // acc: layer 0 -> dW_hh0 ; layer 1 -> dW_ih1    (forward mapping: rows g*H+j, columns of slice s)
    float whT[H], u[H], acc[4][KS];
    float dWih0[4][CS];      // layer-0 threads only
    float db = 0.f;
    {
        const float *whh = layer == 0 ? a.w_hh0 : a.w_hh1;
#pragma unroll
        for (int q = 0; q < H; ++q) {
            whT[q] = whh[(size_t)(s * H + q) * H + j];
            u[q] = layer == 1 ? a.w_ih1[(size_t)(s * H + q) * H + j] : 0.f;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < KS; ++k) acc[g][k] = 0.f;
        dWih0[g][0] = 0.f; dWih0[g][1] = 0.f;
    }
    const float awj = a.attn_w[j];

    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float dc[NB], dhrec[NB], ct[NB], dpj[NB];
        float4 gcur[NB];     // activated gates of the step handled next by the A-part
        float cprev[NB];     // c[t-1] of that step
        float aux0[NB], aux1[NB];  // layer 1: alpha_t, dscore_t ; layer 0: mask_t, unused
        float da4[NB][4];

        // t handled by this layer at macro step m
        auto t_of = [&](int m) { return layer == 1 ? (T - 1 - m) : (T + 1 - m); };

        auto load_step = [&](int t, int n, float4 &g4, float &cp, float &x0, float &x1) {
            const int b = b0 + n;
            g4 = make_float4(0.f, 0.f, 0.f, 0.f); cp = 0.f; x0 = 0.f; x1 = 0.f;
            if (b < B && t >= 0 && t < T) {
                const size_t idx = ((size_t)b * T + t) * H + j;
                g4 = *reinterpret_cast<const float4 *>((layer == 0 ? a.gact0 : a.gact1) + idx * 4);
                if (t > 0) cp = (layer == 0 ? a.cseq0 : a.cseq1)[idx - H];
                if (layer == 1) { x0 = a.alpha[(size_t)b * T + t]; x1 = a.dscore[(size_t)b * T + t]; }
                else            { x0 = a.mask ? a.mask[idx] : 1.f; }
            }
        };

#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int b = b0 + n;
            dc[n] = 0.f; dhrec[n] = 0.f;
            dpj[n] = (layer == 1 && b < B) ? a.dpooled[(size_t)b * H + j] : 0.f;
            // first step of this layer is t = T-1 (layer 1 at m=0, layer 0 at m=2)
            ct[n] = (b < B) ? (layer == 0 ? a.cseq0 : a.cseq1)[((size_t)b * T + (T - 1)) * H + j] : 0.f;
            load_step(T - 1, n, gcur[n], cprev[n], aux0[n], aux1[n]);
#pragma unroll
            for (int g = 0; g < 4; ++g) da4[n][g] = 0.f;
        }
        // staging for B(0): layer-1 operands at t = T-1
        for (int e = tid; e < NB * H; e += NT) {
            const int n = e / H, q = e % H, b = b0 + n;
            const bool ok = b < B;
            hp1[0][n][q] = (ok && T >= 2) ? a.hseq1[((size_t)b * T + (T - 2)) * H + q] : 0.f;
            in1[0][n][q] = ok ? a.in1seq[((size_t)b * T + (T - 1)) * H + q] : 0.f;
            hp0[0][n][q] = 0.f;
            din1[0][n][q] = 0.f; din1[1][n][q] = 0.f;
        }
        for (int e = tid; e < NB * CP; e += NT) xb[0][e / CP][e % CP] = 0.f;
        __syncthreads();

        for (int m = 0; m <= T + 1; ++m) {
            const int par = m & 1;
            const int t = t_of(m);
            const bool active = (t >= 0 && t < T);

            // ---- issue next step's loads early (consumed after the B-part) -----------------------
            float4 gnx[NB]; float cpn[NB], a0n[NB], a1n[NB];
            float st_a[NB], st_b[NB];   // staging values for B(m+1)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                load_step(t - 1, n, gnx[n], cpn[n], a0n[n], a1n[n]);
                const int b = b0 + n;
                const int tn = t - 1;               // step handled by this layer at m+1
                st_a[n] = 0.f; st_b[n] = 0.f;
                if (b < B && tn >= 0 && tn < T) {
                    const size_t base = ((size_t)b * T + tn) * H + j;
                    if (layer == 1) {
                        if (s == 0 && tn > 0) st_a[n] = a.hseq1[base - H];       // h1[tn-1]
                        if (s == 1) st_b[n] = a.in1seq[base];                    // in1[tn]
                    } else {
                        if (s == 0 && tn > 0) st_a[n] = a.hseq0[base - H];       // h0[tn-1]
                        if (s == 1 && j < CP) st_b[n] = (j < C) ? a.x[((size_t)b * T + tn) * C + j] : 0.f;
                    }
                }
            }

            // ---- A-part ---------------------------------------------------------------------------
            if (active) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const float ig = gcur[n].x, fg = gcur[n].y, gg = gcur[n].z, og = gcur[n].w;
                    float dout;
                    if (layer == 1) dout = fmaf(aux0[n], dpj[n], aux1[n] * awj);
                    else            dout = din1[par][n][j] * aux0[n];
                    const float dht = dout + dhrec[n];
                    const float tc = fast_tanh(ct[n]);
                    const float dct = fmaf(dht * og, 1.f - tc * tc, dc[n]);
                    da4[n][0] = dct * gg * ig * (1.f - ig);
                    da4[n][1] = dct * cprev[n] * fg * (1.f - fg);
                    da4[n][2] = dct * ig * (1.f - gg * gg);
                    da4[n][3] = dht * tc * og * (1.f - og);
                    dc[n] = dct * fg;
                    const float mine = s == 0 ? da4[n][0] : s == 1 ? da4[n][1] : s == 2 ? da4[n][2] : da4[n][3];
                    dab[par][layer][n][s][j] = mine;
                    db += mine;
                    if (layer == 1 && a.residual) aux1[n] = dout;   // reuse: needed for d_in below
                }
            }
            __syncthreads();

            // ---- B-part ---------------------------------------------------------------------------
            if (active) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    float rec = 0.f, inp = 0.f;
#pragma unroll
                    for (int q = 0; q < H / 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(&dab[par][layer][n][s][4 * q]);
                        rec = fmaf(whT[4 * q], v.x, rec); rec = fmaf(whT[4 * q + 1], v.y, rec);
                        rec = fmaf(whT[4 * q + 2], v.z, rec); rec = fmaf(whT[4 * q + 3], v.w, rec);
                        if (layer == 1) {
                            inp = fmaf(u[4 * q], v.x, inp); inp = fmaf(u[4 * q + 1], v.y, inp);
                            inp = fmaf(u[4 * q + 2], v.z, inp); inp = fmaf(u[4 * q + 3], v.w, inp);
                        }
                    }
                    dhrec[n] = quad_sum(rec);
                    if (layer == 1) {
                        inp = quad_sum(inp);
                        if (a.residual) inp += aux1[n];
                        if (s == 0) din1[par][n][j] = inp;
                        // dW_ih1 += da (x) in1[t]
#pragma unroll
                        for (int q = 0; q < KS / 4; ++q) {
                            const float4 v = *reinterpret_cast<const float4 *>(&in1[par][n][s * KS + 4 * q]);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                acc[g][4 * q]     = fmaf(da4[n][g], v.x, acc[g][4 * q]);
                                acc[g][4 * q + 1] = fmaf(da4[n][g], v.y, acc[g][4 * q + 1]);
                                acc[g][4 * q + 2] = fmaf(da4[n][g], v.z, acc[g][4 * q + 2]);
                                acc[g][4 * q + 3] = fmaf(da4[n][g], v.w, acc[g][4 * q + 3]);
                            }
                        }
                    } else {
                        // dW_hh0 += da (x) h0[t-1] ; dW_ih0 += da (x) x[t]
#pragma unroll
                        for (int q = 0; q < KS / 4; ++q) {
                            const float4 v = *reinterpret_cast<const float4 *>(&hp0[par][n][s * KS + 4 * q]);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                acc[g][4 * q]     = fmaf(da4[n][g], v.x, acc[g][4 * q]);
                                acc[g][4 * q + 1] = fmaf(da4[n][g], v.y, acc[g][4 * q + 1]);
                                acc[g][4 * q + 2] = fmaf(da4[n][g], v.z, acc[g][4 * q + 2]);
                                acc[g][4 * q + 3] = fmaf(da4[n][g], v.w, acc[g][4 * q + 3]);
                            }
                        }
                        const float2 xv = *reinterpret_cast<const float2 *>(&xb[par][n][s * CS]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            dWih0[g][0] = fmaf(da4[n][g], xv.x, dWih0[g][0]);
                            dWih0[g][1] = fmaf(da4[n][g], xv.y, dWih0[g][1]);
                        }
                    }
                }
            }
            // layer-0 threads also accumulate dW_hh1 for layer 1's step of THIS macro step (t1 = T-1-m)
            if (layer == 0 && m < T) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    float d1[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) d1[g] = dab[par][1][n][g][j];
#pragma unroll
                    for (int q = 0; q < KS / 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4 *>(&hp1[par][n][s * KS + 4 * q]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            u[g * KS + 4 * q]     = fmaf(d1[g], v.x, u[g * KS + 4 * q]);
                            u[g * KS + 4 * q + 1] = fmaf(d1[g], v.y, u[g * KS + 4 * q + 1]);
                            u[g * KS + 4 * q + 2] = fmaf(d1[g], v.z, u[g * KS + 4 * q + 2]);
                            u[g * KS + 4 * q + 3] = fmaf(d1[g], v.w, u[g * KS + 4 * q + 3]);
                        }
                    }
                }
            }
            // ---- stage operands of B(m+1), rotate the prefetched step -------------------------------
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                if (layer == 1) {
                    if (s == 0) hp1[par ^ 1][n][j] = st_a[n];
                    if (s == 1) in1[par ^ 1][n][j] = st_b[n];
                } else {
                    if (s == 0) hp0[par ^ 1][n][j] = st_a[n];
                    if (s == 1 && j < CP) xb[par ^ 1][n][j] = st_b[n];
                }
                if (t <= T - 1) {         // this layer has started: advance to step t-1
                    ct[n] = cprev[n];
                    gcur[n] = gnx[n]; cprev[n] = cpn[n]; aux0[n] = a0n[n]; aux1[n] = a1n[n];
                }
            }
        }
        __syncthreads();   // all B-part reads done before the next group re-initialises LDS
    }

    // ---- partial gradients -> this workgroup's slab ------------------------------------------------
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    if (layer == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = g * H + j;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                slab[a.o_w_hh0 + (size_t)row * H + s * KS + k] = acc[g][k];
                slab[a.o_w_hh1 + (size_t)row * H + s * KS + k] = u[g * KS + k];
            }
#pragma unroll
            for (int k = 0; k < CS; ++k)
                if (s * CS + k < C) slab[a.o_w_ih0 + (size_t)row * C + s * CS + k] = dWih0[g][k];
        }
        slab[a.o_b_ih0 + s * H + j] = db;
        slab[a.o_b_hh0 + s * H + j] = db;
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = g * H + j;
#pragma unroll
            for (int k = 0; k < KS; ++k) slab[a.o_w_ih1 + (size_t)row * H + s * KS + k] = acc[g][k];
        }
        slab[a.o_b_ih1 + s * H + j] = db;
        slab[a.o_b_hh1 + s * H + j] = db;
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
constexpr int X4_MIN_B = 513;        // training batch from which the four-trial kernels are used (H = 48): more than two trials per CU (tools/x4_sweep.py; 513 .. 575 trials are two passes of the two-trial kernels otherwise)

// Which H = 48 forward instantiation a launch takes is a pure function of the launch in the product library.  The diagnostic twin
// (libnsd_hip_diag.so: this file compiled with -DNSD_DIAG=1) can pin it -- 1 / 2 / 4 trials per workgroup, 0 = automatic -- so that
// tests compare the instantiations on the same inputs; the product has neither the entry point nor an environment hook.
#if NSD_DIAG
static int g_force_fwd48 = 0, g_force_bwd48 = 0;
extern "C" int nsd_diag_force_bwd48(int32_t nb) {
    if (nb != 0 && nb != 2 && nb != 4) { nsd_set_error("nsd_diag_force_bwd48: 0, 2 (the one- / two-trial kernel) or 4"); return NSD_E_INVALID; }
    g_force_bwd48 = nb;
    return NSD_OK;
}
static int nsd_diag_forced_bwd48() { return g_force_bwd48; }
extern "C" int nsd_diag_force_fwd48(int32_t nb) {
    if (nb != 0 && nb != 1 && nb != 2 && nb != 4 && nb != 8) { nsd_set_error("nsd_diag_force_fwd48: 0, 1, 2, 4 or 8 (the experimental one-wave-per-layer kernel)"); return NSD_E_INVALID; }
    g_force_fwd48 = nb;
    return NSD_OK;
}
static int nsd_diag_forced_fwd48() { return g_force_fwd48; }
#else
static int nsd_diag_forced_fwd48() { return 0; }
static int nsd_diag_forced_bwd48() { return 0; }
#endif
static int pick_nb(int B) {
    const int cus = nsd_num_cus();
    if (B <= cus) return 1;
    if (B <= 2 * cus) return 2;
    return 4;
}

template <int H>
static int launch_fwd_h(const Lstm2FwdArgs &a, int nb, int grid, hipStream_t st) {
    switch (nb) {
    case 1: hipLaunchKernelGGL((lstm2_fwd_kernel<H, 1>), dim3(grid), dim3(8 * H), 0, st, a); break;
    case 2: hipLaunchKernelGGL((lstm2_fwd_kernel<H, 2>), dim3(grid), dim3(8 * H), 0, st, a); break;
    default: hipLaunchKernelGGL((lstm2_fwd_kernel<H, 4>), dim3(grid), dim3(8 * H), 0, st, a); break;
    }
    return 0;
}
template <int H>
static int launch_bwd_h(const Lstm2BwdArgs &a, int nb, int grid, hipStream_t st) {
    switch (nb) {
    case 1: hipLaunchKernelGGL((lstm2_bwd_kernel<H, 1>), dim3(grid), dim3(8 * H), 0, st, a); break;
    case 2: hipLaunchKernelGGL((lstm2_bwd_kernel<H, 2>), dim3(grid), dim3(8 * H), 0, st, a); break;
    default: hipLaunchKernelGGL((lstm2_bwd_kernel<H, 4>), dim3(grid), dim3(8 * H), 0, st, a); break;
    }
    return 0;
}

// the backward kernels keep at most 2 trials per workgroup (register budget); larger batches loop.  Up to two trials per CU the
// ONE-trial instantiation walks two trials one after the other: since round 4 it is the faster one per trial (H = 48: split-bf16 weight
// gradients, prepared factors, four-step hand-off -- 2 x 127 us against 268-271 us of the two-trial instantiation at 320 .. 512 trials)
static int pick_nb_bwd(int B) { const int nb = pick_nb(B); return nb > 2 ? 2 : (B <= 2 * nsd_num_cus() ? 1 : nb); }

// number of workgroups (== slabs written) the backward kernel uses for batch B
int nsd_lstm2_bwd_grid(int B) {
    const int nb = pick_nb_bwd(B);
    const int ngrp = (B + nb - 1) / nb;
    const int cus = nsd_num_cus();
    return ngrp < cus ? ngrp : cus;
}

int nsd_lstm2_fwd_launch(const Lstm2FwdArgs &a, int H, hipStream_t st) {
    const int nb = pick_nb(a.B);
    const int ngrp = (a.B + nb - 1) / nb;
    const int cap = 2 * nsd_num_cus();
    const int grid = ngrp < cap ? ngrp : cap;
    if (grid <= 0) return NSD_OK;
    switch (H) {
    case 32: launch_fwd_h<32>(a, nb, grid, st); break;
    case 48: {   // role-split kernels, one workgroup per CU, batches loop:
                 //  * one trial per workgroup (nsd_lstm2_fwd48.hip) while that leaves CUs idle or barely covers them -- latency is all
                 //    that counts -- and for inference (the pooling / head tail is built for one trial);
                 //  * FOUR trials per workgroup with the gate products on the matrix pipe (nsd_lstm2_fwd48x4.hip) for training batches
                 //    from X4_MIN_B trials on (three or more trials per CU: a four-trial step costs about what 2.3 one-trial steps cost);
                 //  * two trials per workgroup in lock step (nsd_lstm2_fwd48.hip) where the four-trial kernel does not apply (residual
                 //    extension) and every CU has at least two trials.
        const int cus = nsd_num_cus();
        const int force_nb = nsd_diag_forced_fwd48();           // 0 in the product library (diagnostic build: nsd_diag_force_fwd48)
        if (force_nb == 8 && nsd_lstm2_fwd48w_ok(a)) return nsd_lstm2_fwd48w_launch(a, a.B < cus ? a.B : cus, st);      // experiment (nsd_lstm2_fwd48w.hip)
        const bool x4 = nsd_lstm2_fwd48x4_ok(a) && (force_nb ? force_nb == 4 : a.B >= X4_MIN_B);
        if (x4) {
            // The backward pass of the same batch takes lstm2_bwd48x4_kernel under the same rule (nsd_lstm2_bwd_launch below; its domain
            // contains the forward kernel's): the fused head then leaves the per-step part of the attention's backward to it
            const int force_b = nsd_diag_forced_bwd48();
            Lstm2FwdArgs a4 = a;
            a4.defer_att = a.head_train && (force_b ? force_b == 4 : a.B >= X4_MIN_B);
            const int ngrp4 = (a.B + 3) / 4;
            return nsd_lstm2_fwd48x4_launch(a4, ngrp4 < cus ? ngrp4 : cus, st);
        }
        const bool two = force_nb ? force_nb == 2 : a.B > cus;        // (more than one trial per CU: two in lock step beat two one after the other, 204 vs 224 us at 320 trials)
        if (two && !a.logits_out) { const int ngrp2 = (a.B + 1) / 2; return nsd_lstm2_fwd48_launch(a, 2, ngrp2 < cus ? ngrp2 : cus, st); }
        return nsd_lstm2_fwd48_launch(a, 1, a.B < cus ? a.B : cus, st);
    }
    case 64: launch_fwd_h<64>(a, nb, grid, st); break;
    default: nsd_set_error("lstm2 fwd: unsupported H=%d", H); return NSD_E_INVALID;
    }
    NSD_CHECK_LAUNCH("lstm2_fwd");
    return NSD_OK;
}

int nsd_lstm2_bwd_launch(const Lstm2BwdArgs &a, int H, hipStream_t st) {
    const int nb = pick_nb_bwd(a.B);
    const int grid = nsd_lstm2_bwd_grid(a.B);
    if (grid <= 0) return NSD_OK;
    switch (H) {
    case 32: launch_bwd_h<32>(a, nb, grid, st); break;
    case 48: {   // role-split kernels: four trials per workgroup on the matrix pipe (nsd_lstm2_bwd48x4.hip) from X4_MIN_B trials on, else
                 // one / two trials per workgroup (nsd_lstm2_bwd48.hip).  Same grid either way: the workspace holds one slab per workgroup.
        const int force_nb = nsd_diag_forced_bwd48();
        if (a.da0_out) return nsd_lstm2_bwd48_launch(a, 1, grid, st);      // input gradient requested: the one-trial kernel writes da0 (any batch: it loops)
        if (nsd_lstm2_bwd48x4_ok(a) && (force_nb ? force_nb == 4 : a.B >= X4_MIN_B)) return nsd_lstm2_bwd48x4_launch(a, grid, st);
        return nsd_lstm2_bwd48_launch(a, nb, grid, st);
    }
    case 64: launch_bwd_h<64>(a, nb, grid, st); break;
    default: nsd_set_error("lstm2 bwd: unsupported H=%d", H); return NSD_E_INVALID;
    }
    NSD_CHECK_LAUNCH("lstm2_bwd");
    return NSD_OK;
}
