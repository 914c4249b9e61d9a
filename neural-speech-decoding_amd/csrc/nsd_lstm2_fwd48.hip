// nsd_lstm2_fwd48.hip -- forward of the two-layer H=48 LSTM, role-split workgroup (gfx950).
//
// Replaces self.lstm(x) (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34) for the reference model shape
// (H=48, L=2, C<=8): torch.nn.LSTM semantics, gate order i,f,g,o, two biases, zero initial state, dropout
// multipliers on the layer-0 output.
//
// Measured facts that shape it (MI355X): one wave issues at most one VALU instruction per
// 4 cycles, two or more waves on a SIMD reach one per 2 cycles; a step of the recurrence is a dependent
// chain (LDS -> FMAs -> quad reduction -> sigma/tanh -> LDS -> barrier).  So the step time is set by the
// instruction count of the slowest wave: the work of a step is spread over 9 waves in three roles, every
// mat-vec is issued as v_pk_fma_f32 (2 FMA per instruction, pairs along k), and nothing on the chain
// touches HBM:
//
//   waves 0-2  "L0"   layer 0, step t = m        : W_ih0 x_t + W_hh0 h0_{t-1}  (28 pk_fma), cell update,
//                                                  dropout multiplier, h0 / masked h0 to LDS
//   waves 3-5  "P"    layer-1 input projection of step t = m-1 : W_ih1 in1_t (24 pk_fma) -> LDS
//   waves 6-8  "L1"   layer 1, step t = m-2      : W_hh1 h1_{t-1} (24 pk_fma) + P_t, cell update
//
// Thread (unit j, k-slice s) in every role: 4 gates x 12 (or 2) weights in VGPRs, operands broadcast from
// LDS, DPP quad reduction, lane s of the quad evaluates gate s.  One barrier per step.  x and the dropout
// multipliers are staged through LDS in 32-step chunks, prefetched one chunk ahead.
#include "nsd_args.h"

namespace {

constexpr int H = 48;
constexpr int KS = 12;
constexpr int NT = 576;
constexpr int XCH = 32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NB>
struct FSmem {
    float xs[2][NB][XCH][8];
    float ms[2][NB][XCH][H];
    float h0s[2][NB][H];
    float h0m[2][NB][H];
    float h1s[2][NB][H];
    float pb[2][NB][4 * H];      // layer-1 input projection, [gate*48 + unit]
    float pin[2][NB][H];         // layer-1 input itself (residual top only)
};

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// 12 operands of a k-slice from LDS as 6 pairs
__device__ __forceinline__ void load_slice(const float *p, f32x2 (&v)[6]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float4 u = *reinterpret_cast<const float4 *>(p + 4 * q);
        v[2 * q] = (f32x2){u.x, u.y};
        v[2 * q + 1] = (f32x2){u.z, u.w};
    }
}

// gate pre-activations of unit j, reduced over the 4 k-slices of the quad; returns the one of gate s
__device__ __forceinline__ float reduce_pick(const f32x2 (&acc)[4], const int s) {
    float r[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) r[g] = quad_sum(acc[g].x + acc[g].y);
    return s == 0 ? r[0] : s == 1 ? r[1] : s == 2 ? r[2] : r[3];
}

struct GateConst { float a, b, c; };
__device__ __forceinline__ GateConst gate_const(const int s) {
    GateConst k;
    k.a = (s == 2) ? 2.f : 1.f;
    k.b = (s == 2) ? -2.f * LOG2E_F : -LOG2E_F;
    k.c = (s == 2) ? -1.f : 0.f;
    return k;
}

// ------------------------------------------------------------------------------------------------
// layer 0
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void l0_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int r, const int n_steps) {
    const int j = r >> 2, s = r & 3;
    const int T = a.T, B = a.B, C = a.C;
    f32x2 wx[4], wh[4][6];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = g * H + j;
        wx[g].x = (2 * s < C) ? a.w_ih0[(size_t)row * C + 2 * s] : 0.f;
        wx[g].y = (2 * s + 1 < C) ? a.w_ih0[(size_t)row * C + 2 * s + 1] : 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            wh[g][q].x = a.w_hh0[(size_t)row * H + s * KS + 2 * q];
            wh[g][q].y = a.w_hh0[(size_t)row * H + s * KS + 2 * q + 1];
        }
    }
    const float bias = a.b_ih0[s * H + j] + a.b_hh0[s * H + j];
    const GateConst gk = gate_const(s);
    constexpr int XE = NB * XCH * 8;                 // x floats per chunk
    constexpr int XPT = (XE + 191) / 192;
    constexpr int MPT = NB * 2;                      // mask float4 per thread: XCH*H/4 = 384 per trial / 192 threads

    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float c[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) c[n] = 0.f;
        auto x_at = [&](int e, int t0) -> float {
            const int n = e / (XCH * 8), tl = (e >> 3) & (XCH - 1), ch = e & 7;
            const int b = b0 + n, t = t0 + tl;
            return (e < XE && b < B && t < T && ch < C) ? a.x[((size_t)b * T + t) * C + ch] : 0.f;
        };
        auto mask_at = [&](int e, int t0) -> float4 {      // e: float4 index in [0, NB*384)
            const int n = e / 384, rem = e - n * 384, tl = rem / 12, q = rem - tl * 12;
            const int b = b0 + n, t = t0 + tl;
            if (a.mask && b < B && t < T) return *reinterpret_cast<const float4 *>(a.mask + ((size_t)b * T + t) * H + 4 * q);
            return make_float4(1.f, 1.f, 1.f, 1.f);
        };
        // state buffers and chunk 0
        for (int e = r; e < 2 * NB * H; e += 192) {
            (&sm.h0s[0][0][0])[e] = 0.f; (&sm.h0m[0][0][0])[e] = 0.f; (&sm.h1s[0][0][0])[e] = 0.f; (&sm.pin[0][0][0])[e] = 0.f;
        }
        for (int e = r; e < 2 * NB * 4 * H; e += 192) (&sm.pb[0][0][0])[e] = 0.f;
#pragma unroll
        for (int q = 0; q < XPT; ++q) { const int e = r + 192 * q; if (e < XE) (&sm.xs[0][0][0][0])[e] = x_at(e, 0); }
#pragma unroll
        for (int q = 0; q < MPT; ++q) *reinterpret_cast<float4 *>(&sm.ms[0][0][0][0] + 4 * (r + 192 * q)) = mask_at(r + 192 * q, 0);
        __syncthreads();

        for (int m0 = 0; m0 < n_steps; m0 += XCH) {
            float xr[XPT]; float4 mr[MPT];
#pragma unroll
            for (int q = 0; q < XPT; ++q) xr[q] = x_at(r + 192 * q, m0 + XCH);
#pragma unroll
            for (int q = 0; q < MPT; ++q) mr[q] = mask_at(r + 192 * q, m0 + XCH);
            const int cb = (m0 / XCH) & 1;
            for (int k = 0; k < XCH; ++k) {
                const int m = m0 + k;
                if (m >= n_steps) break;
                const int cur = m & 1, prv = cur ^ 1;
                if (m < T) {
                    const int t = m;
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const int b = b0 + n;
                        const bool valid = b < B;
                        const size_t idx = ((size_t)(valid ? b : 0) * T + t) * H + j;
                        const float mk = sm.ms[cb][n][k][j];
                        const float2 xq = *reinterpret_cast<const float2 *>(&sm.xs[cb][n][k][2 * s]);
                        const f32x2 xv = {xq.x, xq.y};
                        f32x2 hv[6];
                        load_slice(&sm.h0s[prv][n][s * KS], hv);
                        f32x2 acc[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc[g] = wx[g] * xv;
#pragma unroll
                            for (int q = 0; q < 6; ++q) acc[g] = pk_fma(wh[g][q], hv[q], acc[g]);
                        }
                        const float pre = reduce_pick(acc, s) + bias;
                        const float act = gate_act(pre, gk.a, gk.b, gk.c);
                        const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act);
                        const float gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                        c[n] = fmaf(fg, c[n], ig * gg);
                        const float h = og * fast_tanh(c[n]);
                        const float hm = h * mk;
                        if (s == 0) sm.h0s[cur][n][j] = h;
                        if (s == 1) sm.h0m[cur][n][j] = hm;
                        if (valid) {
                            if (a.gact0) a.gact0[idx * 4 + s] = act;
                            if (s == 0 && a.hseq0) a.hseq0[idx] = h;
                            if (s == 1 && a.cseq0) a.cseq0[idx] = c[n];
                            if (s == 2 && a.inseq) a.inseq[idx] = hm;
                        }
                    }
                }
                if (k == XCH - 1) {
#pragma unroll
                    for (int q = 0; q < XPT; ++q) { const int e = r + 192 * q; if (e < XE) (&sm.xs[cb ^ 1][0][0][0])[e] = xr[q]; }
#pragma unroll
                    for (int q = 0; q < MPT; ++q) *reinterpret_cast<float4 *>(&sm.ms[cb ^ 1][0][0][0] + 4 * (r + 192 * q)) = mr[q];
                }
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// layer-1 input projection, one step behind layer 0
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void p_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int r, const int n_steps) {
    const int j = r >> 2, s = r & 3;
    const int T = a.T;
    f32x2 wi[4][6];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            wi[g][q].x = a.w_ih1[(size_t)(g * H + j) * H + s * KS + 2 * q];
            wi[g][q].y = a.w_ih1[(size_t)(g * H + j) * H + s * KS + 2 * q + 1];
        }
    const int ngrp = (a.B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        __syncthreads();
        for (int m = 0; m < n_steps; ++m) {
            if (m >= 1 && m <= T) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x2 iv[6];
                    load_slice(&sm.h0m[(m - 1) & 1][n][s * KS], iv);
                    f32x2 acc[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = wi[g][0] * iv[0];
#pragma unroll
                        for (int q = 1; q < 6; ++q) acc[g] = pk_fma(wi[g][q], iv[q], acc[g]);
                    }
                    sm.pb[m & 1][n][s * H + j] = reduce_pick(acc, s);
                    if (a.residual && s == 0) sm.pin[m & 1][n][j] = sm.h0m[(m - 1) & 1][n][j];
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// layer 1, two steps behind layer 0
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void l1_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int r, const int n_steps) {
    const int j = r >> 2, s = r & 3;
    const int T = a.T, B = a.B;
    f32x2 wh[4][6];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            wh[g][q].x = a.w_hh1[(size_t)(g * H + j) * H + s * KS + 2 * q];
            wh[g][q].y = a.w_hh1[(size_t)(g * H + j) * H + s * KS + 2 * q + 1];
        }
    const float bias = a.b_ih1[s * H + j] + a.b_hh1[s * H + j];
    const GateConst gk = gate_const(s);
    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        float c[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) c[n] = 0.f;
        __syncthreads();
        for (int m = 0; m < n_steps; ++m) {
            const int t = m - 2;
            if (t >= 0 && t < T) {
                const int cur = m & 1, prv = cur ^ 1;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const int b = b0 + n;
                    const bool valid = b < B;
                    const size_t idx = ((size_t)(valid ? b : 0) * T + t) * H + j;
                    const float pj = sm.pb[prv][n][s * H + j];          // input projection of this step (gate s)
                    f32x2 hv[6];
                    load_slice(&sm.h1s[prv][n][s * KS], hv);
                    f32x2 acc[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = wh[g][0] * hv[0];
#pragma unroll
                        for (int q = 1; q < 6; ++q) acc[g] = pk_fma(wh[g][q], hv[q], acc[g]);
                    }
                    const float pre = reduce_pick(acc, s) + (pj + bias);
                    const float act = gate_act(pre, gk.a, gk.b, gk.c);
                    const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act);
                    const float gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
                    c[n] = fmaf(fg, c[n], ig * gg);
                    const float h = og * fast_tanh(c[n]);
                    if (s == 0) sm.h1s[cur][n][j] = h;
                    if (valid) {
                        if (a.gact1) a.gact1[idx * 4 + s] = act;
                        if (s == 0 && a.hseq1) a.hseq1[idx] = h;
                        if (s == 1 && a.cseq1) a.cseq1[idx] = c[n];
                        if (s == 2 && a.top) a.top[idx] = a.residual ? h + sm.pin[prv][n][j] : h;
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <int NB>
__global__ __launch_bounds__(NT) void lstm2_fwd48_kernel(Lstm2FwdArgs a) {
    __shared__ __align__(16) FSmem<NB> sm;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // macro steps 0..T+1, padded to whole x chunks so that every role runs the same number of barriers
    const int n_steps = ((a.T + 2 + XCH - 1) / XCH) * XCH;
    if (wave < 3)      l0_role<NB>(a, sm, tid, n_steps);
    else if (wave < 6) p_role<NB>(a, sm, tid - 192, n_steps);
    else               l1_role<NB>(a, sm, tid - 384, n_steps);
}

}  // namespace

int nsd_lstm2_fwd48_launch(const Lstm2FwdArgs &a, int nb, int grid, hipStream_t st) {
    switch (nb) {
    case 1: hipLaunchKernelGGL((lstm2_fwd48_kernel<1>), dim3(grid), dim3(NT), 0, st, a); break;
    case 2: hipLaunchKernelGGL((lstm2_fwd48_kernel<2>), dim3(grid), dim3(NT), 0, st, a); break;
    default: hipLaunchKernelGGL((lstm2_fwd48_kernel<4>), dim3(grid), dim3(NT), 0, st, a); break;
    }
    NSD_CHECK_LAUNCH("lstm2_fwd48");
    return NSD_OK;
}
