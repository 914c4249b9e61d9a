// nsd_lstm2_fwd48.hip -- forward of the two-layer H=48 LSTM, role-split workgroup (gfx950).
//
// Replaces self.lstm(x) (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34) for the reference model shape
// (H=48, L=2, C<=8): torch.nn.LSTM semantics, gate order i,f,g,o, two biases, zero initial state, dropout
// multipliers on the layer-0 output.
//
// One workgroup = one trial, one barrier per time step, all weights in VGPRs, operands through LDS, nothing on
// the recurrence touches HBM.  A step is a dependent chain (LDS -> FMAs -> quad reduction -> sigma/tanh -> LDS ->
// barrier); what the measurements say about it (MI355X): every mat-vec goes out as v_pk_fma_f32
// with four independent accumulator pairs; the chain after the FMAs is cut to 16 dependent ops; the LDS serves
// requests in arrival order, so helper waves sleep a little after each barrier; waves are placed on SIMDs by role.
//
// 12 waves, role = f(SIMD g = wave & 3, slot q = wave >> 2)   (the dispatcher deals waves round-robin over the SIMDs):
//   g 0..2, q 0  "L1"   layer 1 part g, step t = m-2 : W_hh1 h1_{t-1} (24 pk_fma) + P_t, cell update
//   g 0..2, q 1  "L0"   layer 0 part g, step t = m   : W_ih0 x_t + W_hh0 h0_{t-1} (28 pk_fma), cell update,
//                                                      dropout multiplier, h0 / masked h0 to LDS
//   g 3,    q 0..2 "P"  layer-1 input projection of step t = m-1 : W_ih1 in1_t (24 pk_fma) -> LDS
//   g 0, q 2  "saver" (training) LDS save ring -> HBM, or "pool" (inference): attention pooling + head, one launch
//   g 1, q 2  "tpool" (training with the fused head) attention pooling along the recurrence + the head's
//                     forward / loss / backward in the kernel tail; otherwise spare
//   g 2, q 2  "spare"  generates the dropout multipliers when the random streams are drawn in the kernel
//
// Thread (unit j, k-slice s) in every chain role: 4 gates x 12 (or 2) weights in VGPRs, operands broadcast from
// LDS, DPP quad reduction, lane s of the quad evaluates gate s.  x and the dropout multipliers are staged through
// LDS in 32-step chunks, prefetched one chunk ahead.
#include "nsd_args.h"
#include "nsd_prof.h"

namespace {

constexpr int H = 48;
constexpr int KS = 12;
constexpr int NT = 768;      // 12 waves, 3 per SIMD (see the role table in the kernel)
constexpr int XCH = 32;
typedef float f32x2 __attribute__((ext_vector_type(2)));
// ---- activations for the backward pass leave through LDS: a per-step record written by the chain lanes
// (cheap ds_write_b32) and streamed to HBM by a dedicated wave with 16-byte stores, 8 steps at a time.
// Per-step dword stores from the 6 chain waves (~22 VMEM instructions per step) would sit on the recurrence.
constexpr int SREC = 384;                 // floats per (layer, step): gates[192] | h[48] | c[48] | in1 or top[48] | spare[48]
constexpr int SREC4 = SREC / 4;           // 96
constexpr int SRING = 16;                 // steps kept in LDS (two 8-step chunks)
constexpr int SCH = 8;
#ifndef NSD_P_SLEEP
#define NSD_P_SLEEP 2
#endif
#ifndef NSD_S_SLEEP
#define NSD_S_SLEEP 3
#endif
constexpr int P_SLEEP = NSD_P_SLEEP, S_SLEEP = NSD_S_SLEEP;   // see step_barrier
// fused train head limits (larger shapes use the separate head kernel)
constexpr int TT_TMAX = 1024, TT_KMAX = 8, TT_W0S = 49;
constexpr int NT_TRAIN = NT;
constexpr int TT_PARTS = NT_TRAIN / 48;   // 16

template <int NB>
struct FSmem {
    float xs[2][NB][XCH][8];
    float ms[2][NB][XCH][H];
    float pb[2][NB][4 * H];      // layer-1 input projection, [gate*48 + unit]
    // save ring, indexed by macro step % SRING.  It is also the STATE of the recurrence: the next step reads h (and the
    // masked h that feeds layer 1) straight from the previous step's record, so a chain lane issues exactly two
    // unconditional ds_write_b32 per step (its gate, and slot s of {h, c, in1/top, spare})
    float sv[SRING][NB][2][SREC];
    float pk[NB][8];             // inference tail: softmax numerators of the chunk being pooled
    float vec[NB][64];           // inference tail: LayerNorm output / activated fc.0 output
    // fused train head (head_train): raw attention scores of the trial, staged head weights, small vectors
    float sc[NB][TT_TMAX];
    float w0[64 * TT_W0S];       // fc.0 weight rows, stride 49 (odd: lane f reads row f without bank conflicts)
    float w3[TT_KMAX * 64];
    float vln[64], vx[64], vz[64], vdz[64], vdl[64], dp[64];
    float md[4];                 // {running max, 1/denominator}
    float red[16];
    float part[TT_PARTS][H];
};

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int NB>
__device__ __forceinline__ void train_tail(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int tid, const int b, const int n);
template <int NB>
__device__ __forceinline__ void tail_all(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int tid, const int b0);

// 12 operands of a k-slice from LDS as 6 pairs
__device__ __forceinline__ void load_slice(const float *p, f32x2 (&v)[6]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float4 u = *reinterpret_cast<const float4 *>(p + 4 * q);
        v[2 * q] = (f32x2){u.x, u.y};
        v[2 * q + 1] = (f32x2){u.z, u.w};
    }
}

// gate pre-activations of unit j, reduced over the 4 k-slices of the quad; lane s returns the sum of gate s.
// Reduce-scatter (3 DPP adds, depth 4) instead of 4 full quad sums (8 DPP adds) + a 3-deep select.
__device__ __forceinline__ float reduce_pick(const f32x2 (&acc)[4], const int s) {
    const float r0 = acc[0].x + acc[0].y, r1 = acc[1].x + acc[1].y;
    const float r2 = acc[2].x + acc[2].y, r3 = acc[3].x + acc[3].y;
    const bool odd = (s & 1) != 0, hi = (s & 2) != 0;
    const float ra = (odd ? r1 : r0) + quad_xor1(odd ? r0 : r1);   // gate (odd ? 1 : 0) over lanes {s, s^1}
    const float rb = (odd ? r3 : r2) + quad_xor1(odd ? r2 : r3);   // gate (odd ? 3 : 2) over lanes {s, s^1}
    return (hi ? rb : ra) + quad_xor2(hi ? ra : rb);
}


// ---- the step's dependent chain, kept as short as the arithmetic allows (every op on it costs ~10 cycles of every
// step; instructions off it are free) ----
//  * weights and biases are pre-multiplied by the exp2 argument scale of their gate (-log2e for sigma rows, -2 log2e
//    for the tanh row), and the input projection + bias is the INITIAL value of the accumulator of the lane's own
//    gate: the reduced sum is the exp2 argument itself;
//  * the tanh lane (s == 2) returns KC * tanh with KC = -2 log2e, and the cell state is carried as cK = KC * c, so
//    that exp2(cK) = e^(-2c) needs no multiply;  c and g are un-scaled off the chain for the saved activations;
//  * i*g exists only in lane 2 (v_mul_f32_dpp with its own value), f*cK in all lanes, their sum is one
//    v_add_f32_dpp;  h = o * (2r - 1) is one FMA with 2o and -o prepared while the cell waits.
constexpr float KC = -2.f * LOG2E_F;
constexpr float INV_KC = 1.f / KC;
__host__ __device__ constexpr float gate_scale(const int g) { return g == 2 ? -2.f * LOG2E_F : -LOG2E_F; }

struct CellOut { float gate, c, h; };     // gate: this lane's activation (un-scaled), c: new cell state, h: o * tanh(c) * hmul
// arg: exp2 argument of this lane's gate; cK: scaled cell state (updated); hmul: multiplier folded into h (dropout)
__device__ __forceinline__ CellOut cell_step(const float arg, float &cK, const int s, const float hmul) {
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg));
    const float act = s == 2 ? fmaf(2.f * KC, r, -KC) : r;                 // lane 2: KC * tanh(pre)
    const float o = quad_bcast<3>(act);
    const float o2 = (2.f * hmul) * o, on = -hmul * o;                      // off the chain (the cell is still being formed)
    const float igK = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), QP(0, 0, 0, 0), 0xF, 0xF, true)) * act;   // lane 2: i * KC g
    const float fcK = quad_bcast<1>(act) * cK;
    cK = quad_bcast<2>(igK) + fcK;
    const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cK));
    CellOut out;
    out.h = fmaf(o2, rc, on);
    out.c = cK * INV_KC;
    out.gate = s == 2 ? act * INV_KC : act;
    return out;
}

// x and the explicit dropout multipliers of a 32-step chunk, element e of the chunk image (trial-major): zero / one padding
template <int NB>
__device__ __forceinline__ float chunk_x_at(const Lstm2FwdArgs &a, const int b0, const int e, const int t0) {
    const int n = e / (XCH * 8), tl = (e >> 3) & (XCH - 1), ch = e & 7;
    const int b = b0 + n, t = t0 + tl;
    return (e < NB * XCH * 8 && b < a.B && t < a.T && ch < a.C) ? a.x[((size_t)b * a.T + t) * a.C + ch] : 0.f;
}
__device__ __forceinline__ float4 chunk_mask_at(const Lstm2FwdArgs &a, const int b0, const int e, const int t0) {      // e: float4 index in [0, NB*384)
    const int n = e / 384, rem = e - n * 384, tl = rem / 12, q = rem - tl * 12;
    const int b = b0 + n, t = t0 + tl;
    if (a.mask && b < a.B && t < a.T) return *reinterpret_cast<const float4 *>(a.mask + ((size_t)b * a.T + t) * H + 4 * q);
    return make_float4(1.f, 1.f, 1.f, 1.f);
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
        const float gs = gate_scale(g);
        wx[g].x = (2 * s < C) ? gs * a.w_ih0[(size_t)row * C + 2 * s] : 0.f;
        wx[g].y = (2 * s + 1 < C) ? gs * a.w_ih0[(size_t)row * C + 2 * s + 1] : 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            wh[g][q].x = gs * a.w_hh0[(size_t)row * H + s * KS + 2 * q];
            wh[g][q].y = gs * a.w_hh0[(size_t)row * H + s * KS + 2 * q + 1];
        }
    }
    const float biasK = (s == 2 ? gate_scale(2) : gate_scale(0)) * (a.b_ih0[s * H + j] + a.b_hh0[s * H + j]);
    const int hslot = (s == 1) ? 0 : s;          // record slot this lane writes its h to (lanes 0 and 1: the same h twice)
    constexpr int XE = NB * XCH * 8;                 // x floats per chunk
    constexpr int XPT = (XE + 191) / 192;
    constexpr int MPT = NB * 2;                      // mask float4 per thread: XCH*H/4 = 384 per trial / 192 threads

    Prof prof = prof_init(a.dbg);
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
            if (a.rng.on) {                                    // chunk 0 only: later chunks are generated by the P waves
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = (b < B && t < T && nsd_rand_u32(a.rng.seed, a.rng.base, ((uint64_t)b * T + t) * H + 4 * q + u) < a.rng.thr_lstm)
                               ? 0.f : a.rng.keep_lstm;
                return (b < B && t < T) ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(1.f, 1.f, 1.f, 1.f);
            }
            if (a.mask && b < B && t < T) return *reinterpret_cast<const float4 *>(a.mask + ((size_t)b * T + t) * H + 4 * q);
            return make_float4(1.f, 1.f, 1.f, 1.f);
        };
        // state buffers and chunk 0
        for (int e = r; e < SRING * NB * 2 * SREC; e += 192) (&sm.sv[0][0][0][0])[e] = 0.f;    // h(-1) = 0 lives in slot 15
        for (int e = r; e < 2 * NB * 4 * H; e += 192) (&sm.pb[0][0][0])[e] = 0.f;
#pragma unroll
        for (int q = 0; q < XPT; ++q) { const int e = r + 192 * q; if (e < XE) (&sm.xs[0][0][0][0])[e] = x_at(e, 0); }
#pragma unroll
        for (int q = 0; q < MPT; ++q) *reinterpret_cast<float4 *>(&sm.ms[0][0][0][0] + 4 * (r + 192 * q)) = mask_at(r + 192 * q, 0);
        step_barrier<false>(prof);

        for (int m0 = 0; m0 < n_steps; m0 += XCH) {
            // (the NEXT chunk of x and of the explicit multipliers is fetched and written to the other LDS half by the spare wave: 19
            // registers held across the chunk by these chain waves were what pushed the two-trial instantiation into scratch)
            const int cb = (m0 / XCH) & 1;
            for (int kh = 0; kh < XCH; kh += SRING) {
#pragma unroll
              for (int kr = 0; kr < SRING; ++kr) {
                const int k = kh + kr;
                const int m = m0 + k;
                if (m < T && !ablated(a.ablate, 64)) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const float mk = sm.ms[cb][n][k][j];
                        const float2 xq = *reinterpret_cast<const float2 *>(&sm.xs[cb][n][k][2 * s]);
                        const f32x2 xv = {xq.x, xq.y};
                        f32x2 hv[6];
                        load_slice(&sm.sv[(kr + SRING - 1) & (SRING - 1)][n][0][192 + s * KS], hv);
                        f32x2 acc[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            acc[g] = pk_fma(wx[g], xv, (f32x2){g == s ? biasK : 0.f, 0.f});
#pragma unroll
                            for (int q = 0; q < 6; ++q) acc[g] = pk_fma(wh[g][q], hv[q], acc[g]);
                        }
                        const CellOut o = cell_step(reduce_pick(acc, s), c[n], s, s == 2 ? mk : 1.f);   // lane 2: h * dropout multiplier
                        float *sr = &sm.sv[kr][n][0][0];
                        sr[192 + 48 * hslot + j] = o.h;                 // slots: h | c | masked h | spare (h again)
                        sr[192 + 48 + j] = o.c;                         // the quad's four lanes hold the same c
                        sr[4 * j + s] = o.gate;
                    }
                }
                step_barrier<false>(prof);
              }
            }
        }
        if (a.head_train) tail_all<NB>(a, sm, threadIdx.x, b0);   // (its first barrier: the save ring of this trial group is drained)
        else step_barrier<false>(prof);      // save ring drained
    }
    prof_store(a.dbg, prof);
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
            wi[g][q].x = gate_scale(g) * a.w_ih1[(size_t)(g * H + j) * H + s * KS + 2 * q];
            wi[g][q].y = gate_scale(g) * a.w_ih1[(size_t)(g * H + j) * H + s * KS + 2 * q + 1];
        }
    Prof prof = prof_init(a.dbg);
    const int ngrp = (a.B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        step_barrier<false>(prof);
        for (int m0 = 0; m0 < n_steps; m0 += SRING) {
#pragma unroll
          for (int k = 0; k < SRING; ++k) {
            const int m = m0 + k;
            if (m >= 1 && m <= T && !ablated(a.ablate, 128)) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    f32x2 iv[6];
                    load_slice(&sm.sv[(k + SRING - 1) & (SRING - 1)][n][0][288 + s * KS], iv);
                    f32x2 acc[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = wi[g][0] * iv[0];
#pragma unroll
                        for (int q = 1; q < 6; ++q) acc[g] = pk_fma(wi[g][q], iv[q], acc[g]);
                    }
                    sm.pb[k & 1][n][s * H + j] = reduce_pick(acc, s);
                }
            }
            step_barrier<false, P_SLEEP>(prof);
          }
        }
        if (a.head_train) tail_all<NB>(a, sm, threadIdx.x, grp * NB);   // (its first barrier: the save ring of this trial group is drained)
        else step_barrier<false>(prof);      // save ring drained
    }
    prof_store(a.dbg, prof);
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
            wh[g][q].x = gate_scale(g) * a.w_hh1[(size_t)(g * H + j) * H + s * KS + 2 * q];
            wh[g][q].y = gate_scale(g) * a.w_hh1[(size_t)(g * H + j) * H + s * KS + 2 * q + 1];
        }
    const float biasK = (s == 2 ? gate_scale(2) : gate_scale(0)) * (a.b_ih1[s * H + j] + a.b_hh1[s * H + j]);
    const int hslot = (s == 1) ? 0 : s;
    Prof prof = prof_init(a.dbg);
    const int ngrp = (B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        float c[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) c[n] = 0.f;
        step_barrier<false>(prof);
        // The time loop is unrolled by the ring length: ring slots, buffer parities and every LDS offset become
        // immediates.  (A wave issues about one instruction per 5 cycles whatever its ILP -- tools/isa_loops.py --
        // so scalar index arithmetic in the step body costs as much as the FMAs.)
        for (int m0 = 0; m0 < n_steps; m0 += SRING) {
#pragma unroll
          for (int k = 0; k < SRING; ++k) {
            const int m = m0 + k;
            const int t = m - 2;
            prof_mark<-1, false>(prof);
            if (t >= 0 && t < T && !ablated(a.ablate, 1024)) {
                const int prv = (k & 1) ^ 1;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const float pj = sm.pb[prv][n][s * H + j];          // input projection of this step (gate s)
                    f32x2 hv[6];
                    load_slice(&sm.sv[(k + SRING - 1) & (SRING - 1)][n][1][192 + s * KS], hv);
                    prof_mark<0, true>(prof);        // seg0: LDS operands arrived
                    const float pjb = pj + biasK;
                    f32x2 acc[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[g] = pk_fma(wh[g][0], hv[0], (f32x2){g == s ? pjb : 0.f, 0.f});
#pragma unroll
                        for (int q = 1; q < 6; ++q) acc[g] = pk_fma(wh[g][q], hv[q], acc[g]);
                    }
                    prof_mark<1, false>(prof);       // seg1: 24 pk_fma issued (not necessarily retired)
                    const float arg = reduce_pick(acc, s);
                    prof_mark<2, false>(prof);       // seg2: quad reduction + select
                    CellOut o = cell_step(arg, c[n], s, 1.f);
                    prof_mark<4, true>(prof);        // seg4: gates, cell update, tanh
                    float *sr = &sm.sv[k][n][1][0];
                    // layer-1 input of step t was written two macro steps ago
                    if (a.residual && s == 2) o.h += sm.sv[(k + SRING - 2) & (SRING - 1)][n][0][288 + j];
                    sr[192 + 48 * hslot + j] = o.h;               // slots: h | c | top | spare (h again)
                    sr[192 + 48 + j] = o.c;
                    sr[4 * j + s] = o.gate;
                    prof_mark<5, false>(prof);       // seg5: record for the saver wave
                }
            }
            step_barrier<false>(prof);
          }
        }
        if (a.head_train) tail_all<NB>(a, sm, threadIdx.x, grp * NB);   // (its first barrier: the save ring of this trial group is drained)
        else step_barrier<false>(prof);      // save ring drained
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// saver wave: LDS save ring -> HBM with 16-byte stores, one 8-step chunk behind the chain
// ------------------------------------------------------------------------------------------------
constexpr int SPIECES = 2 * SCH * SREC4 / 64;     // 24 wave-wide pieces (1 KB) per trial and chunk

// One 16-byte piece of the chunk image per lane and q: where it lies in the ring and where it goes.  Two registers per piece (24
// pieces): the destination as a 32-bit float offset from the workspace's hseq region (all saved arrays live in ONE workspace buffer,
// a few hundred MB), and {ring offset, time index in chunk 0, row length} packed -- five registers per piece (a 64-bit pointer and
// three ints) left no room for two trials' worth of pieces in flight.
struct SvDesc {
    int dst;             // float offset of (trial 0, t = 0) for this lane's 16 bytes, relative to sv_base; < 0: not saved
    unsigned meta;       // bits 0..15 float offset inside sv[0][0] (ring slot 0, trial 0) | bits 16..23 (t0 + 2) | bit 24 row = 4H floats (else H)
};

template <int NB>
__device__ __forceinline__ void saver_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int lane, const int n_steps) {
    const int T = a.T;
    SvDesc d[SPIECES];
    float *const sv_base = a.hseq0;                            // (training launches always carry the whole workspace)
#pragma unroll
    for (int q = 0; q < SPIECES; ++q) {
        const int e = q * 64 + lane;                       // float4 index in the chunk image [k][layer][84]
        const int k = e / (2 * SREC4), rem = e - k * (2 * SREC4);
        const int layer = rem / SREC4, w = rem - layer * SREC4;
        const int t0 = k - (layer == 1 ? 2 : 0);
        const int lds_off = (k * NB * 2 + layer) * SREC + 4 * w;
        float *dst = nullptr; int wide = 0;
        if (w < 48)      { dst = layer == 0 ? a.gact0 : a.gact1; wide = 1; if (dst) dst += 4 * w; }
        else if (w < 60) { dst = layer == 0 ? a.hseq0 : a.hseq1; if (dst) dst += 4 * (w - 48); }
        else if (w < 72) { dst = layer == 0 ? a.cseq0 : a.cseq1; if (dst) dst += 4 * (w - 60); }
        else if (w < 84) { dst = layer == 0 ? a.inseq : a.top;   if (dst) dst += 4 * (w - 72); }
        d[q].dst = (dst && sv_base) ? (int)(dst - sv_base) : -1;
        d[q].meta = (unsigned)lds_off | ((unsigned)(t0 + 2) << 16) | ((unsigned)wide << 24);
    }
    Prof prof = prof_init(a.dbg);
    // The LDS reads of a batch of pieces are all issued before the first store: one LDS latency per call instead of one
    // per piece (the lane conditions are divergent, so the compiler would otherwise serialise read -> wait -> store).
    auto flush = [&](const int chunk, const int b0, const int q0, const int q1) {
        const float *ring = &sm.sv[(chunk & 1) * SCH][0][0][0];
#pragma unroll
        for (int qb = 0; qb < SPIECES; qb += 3) {
            if (qb < q0 || qb >= q1) continue;
            float4 v[3][NB];
            bool ok[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = qb + u;
                const int t = (int)((d[q].meta >> 16) & 0xffu) - 2 + SCH * chunk;
                ok[u] = d[q].dst >= 0 && (unsigned)t < (unsigned)T;
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    v[u][n] = ok[u] ? *reinterpret_cast<const float4 *>(ring + (d[q].meta & 0xffffu) + n * 2 * SREC) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = qb + u;
                const int t = (int)((d[q].meta >> 16) & 0xffu) - 2 + SCH * chunk;
                const unsigned row = (d[q].meta >> 24) & 1u ? 4u * H : (unsigned)H;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const int b = b0 + n;
                    if (ok[u] && b < a.B)
                        *reinterpret_cast<float4 *>(sv_base + d[q].dst + (size_t)((unsigned)(b * T + t)) * row) = v[u][n];
                }
            }
        }
    };
    const int ngrp = (a.B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        step_barrier<false>(prof);
        for (int m0 = 0; m0 < n_steps; m0 += SCH) {
            const int done = m0 / SCH - 1;                 // chunk completed before this one started
            // 24 pieces, 3 per step
            if (done >= 0) flush(done, b0, 0, 3);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 3, 6);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 6, 9);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 9, 12);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 12, 15);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 15, 18);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 18, 21);
            step_barrier<false, S_SLEEP>(prof);
            if (done >= 0) flush(done, b0, 21, 24);
            step_barrier<false, S_SLEEP>(prof);
        }
        flush(n_steps / SCH - 1, b0, 0, SPIECES);          // last chunk (its LDS image is complete: barrier above)
        if (a.head_train) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail reads the top rows back
        if (a.head_train) tail_all<NB>(a, sm, threadIdx.x, b0);
        else step_barrier<false>(prof);                     // keep the ring intact until it has been read
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// attention pooling over time as an ONLINE softmax over the layer-1 records of the save ring, one 8-step chunk behind
// the chain (running max, denominator, weighted sum in lane j).  The update of a chunk is a long dependent sequence
// (dot products, three reductions, two exponentials, the weighted sum), and this wave stands at the same step
// barriers as the chains: done in one piece it stretched one step in eight.  So it is cut into four stages, one per
// step, and every reduction is DPP / v_readlane (no LDS crossbar).  Lane = (kq = lane >> 3: step of the chunk,
// part = lane & 7: 6 of the 48 units).
// ------------------------------------------------------------------------------------------------
struct PoolRun {
    float awp[6], ab;
    float mrun, den, pooled;        // running max, denominator, weighted sum (lane j < 48)
    float sc, pkv, scale;           // chunk in flight
    bool ok, skip;
};
template <int NB>
__device__ __forceinline__ void pool_init(PoolRun &p, const Lstm2FwdArgs &a, const int lane) {
#pragma unroll
    for (int u = 0; u < 6; ++u) p.awp[u] = a.attn_w[6 * (lane & 7) + u];
    p.ab = a.attn_b[0];
}
__device__ __forceinline__ void pool_reset(PoolRun &p) { p.mrun = -INFINITY; p.den = 0.f; p.pooled = 0.f; p.skip = true; }

template <int STAGE, int NB>
__device__ __forceinline__ void pool_stage(PoolRun &p, FSmem<NB> &sm, const int chunk, const int lane, const int T, float *sc_out, const int n = 0) {
    const int kq = lane >> 3, part = lane & 7;
    if (STAGE == 0) {            // scores of the chunk's 8 steps
        const int t = SCH * chunk + kq - 2;               // layer-1 time index of ring slot kq of this chunk
        const float *rec = &sm.sv[(chunk & 1) * SCH + kq][n][1][288];
        float sc = 0.f;
#pragma unroll
        for (int u = 0; u < 6; ++u) sc = fmaf(p.awp[u], rec[6 * part + u], sc);
        sc = oct_sum(sc);
        p.ok = (unsigned)t < (unsigned)T;
        p.sc = p.ok ? sc + p.ab : -INFINITY;
        if (sc_out && p.ok && part == 0) sc_out[t] = p.sc;
    } else if (STAGE == 1) {     // new running max, numerators
        const float cm = rows_combine_max(fmaxf(p.sc, row_ror<8>(p.sc)));
        const float mnew = fmaxf(p.mrun, cm);
        p.skip = (mnew == -INFINITY);                     // nothing valid yet (wave-uniform)
        if (p.skip) return;
        p.pkv = p.ok ? __expf(p.sc - mnew) : 0.f;
        p.scale = __expf(p.mrun - mnew);                  // exp(-inf) = 0 on the first valid chunk
        p.mrun = mnew;
    } else if (STAGE == 2) {     // denominator
        if (p.skip) return;
        const float ps = rows_combine_sum(p.pkv + row_ror<8>(p.pkv));
        p.den = fmaf(p.den, p.scale, ps);
        if (part == 0) sm.pk[n][kq] = p.pkv;
    } else {                     // weighted sum of the 8 rows (pk was published by the barrier after stage 2)
        if (p.skip) return;
        float acc = p.pooled * p.scale;
        if (lane < H) {
#pragma unroll
            for (int k = 0; k < SCH; ++k) acc = fmaf(sm.pk[n][k], sm.sv[(chunk & 1) * SCH + k][n][1][288 + lane], acc);
        }
        p.pooled = acc;
    }
}

// the time loop of a pooling wave: one stage of chunk (m0/8 - 1) per step -- trial 0 in steps 0..3 of the 8, trial 1 (two trials
// per workgroup) in steps 4..7: the chunk's ring half is not rewritten before the next 8 steps -- then the last chunk in one go
template <int NB>
__device__ __forceinline__ void pool_loop(PoolRun (&p)[NB], FSmem<NB> &sm, const int lane, const int T, const int n_steps, const bool keep_scores,
                                          Prof &prof) {
    static_assert(NB <= 2, "one pooling stage per step: 4 stages x NB trials per 8-step chunk");
    for (int m0 = 0; m0 < n_steps; m0 += SCH) {
        const int done = m0 / SCH - 1;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (n < NB) {
                float *sc_out = keep_scores ? sm.sc[n < NB ? n : 0] : nullptr;
                if (done >= 0) pool_stage<0, NB>(p[n < NB ? n : 0], sm, done, lane, T, sc_out, n);
                step_barrier<false, S_SLEEP>(prof);
                if (done >= 0) pool_stage<1, NB>(p[n < NB ? n : 0], sm, done, lane, T, sc_out, n);
                step_barrier<false, S_SLEEP>(prof);
                if (done >= 0) pool_stage<2, NB>(p[n < NB ? n : 0], sm, done, lane, T, sc_out, n);
                step_barrier<false, S_SLEEP>(prof);
                if (done >= 0) pool_stage<3, NB>(p[n < NB ? n : 0], sm, done, lane, T, sc_out, n);
                step_barrier<false, S_SLEEP>(prof);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) step_barrier<false, S_SLEEP>(prof);
            }
        }
    }
    const int last = n_steps / SCH - 1;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        float *sc_out = keep_scores ? sm.sc[n] : nullptr;
        pool_stage<0, NB>(p[n], sm, last, lane, T, sc_out, n);
        pool_stage<1, NB>(p[n], sm, last, lane, T, sc_out, n);
        pool_stage<2, NB>(p[n], sm, last, lane, T, sc_out, n);
        pool_stage<3, NB>(p[n], sm, last, lane, T, sc_out, n);     // same wave: the LDS queue is in order, the reads see pk
    }
}

// ------------------------------------------------------------------------------------------------
// inference tail (wave 9 when logits_out != null): lstm_eeg_model.py:35-39 and the class softmax of :97 fused into
// the LSTM kernel.  Attention pooling over time runs as an ONLINE softmax over the layer-1 records of the save ring,
// one 8-step chunk behind the chain (running max m, denominator, weighted sum in lane j); the [T,H] sequence is
// never written to HBM, so the kernel's traffic is the algorithmic 8 000 B in + K*4 B out per trial.
// ------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void pool_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int lane, const int n_steps) {
    static_assert(NB == 1, "inference tail is built for one trial per workgroup");
    const int T = a.T, K = a.K, F = a.F;
    PoolRun pr[NB];
    pool_init<NB>(pr[0], a, lane);
    Prof prof = prof_init(a.dbg);
    const int ngrp = (a.B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b = grp;
        pool_reset(pr[0]);
        step_barrier<false>(prof);
        pool_loop<NB>(pr, sm, lane, T, n_steps, false, prof);
        const float pooled = pr[0].pooled, den = pr[0].den;
        // ---- LayerNorm (biased variance, eps in the sqrt), fc.0 -> RReLU(eval) -> fc.3, softmax over classes ----
        const float p = lane < H ? pooled / den : 0.f;
        const float mu = wave_sum(p) * (1.0f / H);
        const float dlt = lane < H ? p - mu : 0.f;
        const float rstd = 1.0f / sqrtf(wave_sum(dlt * dlt) * (1.0f / H) + 1e-5f);
        if (lane < H) sm.vec[0][lane] = dlt * rstd * a.ln_w[lane] + a.ln_b[lane];
        float z = 0.f;
        if (lane < F) {
            float acc = a.fc0_b[lane];
            const float *w = a.fc0_w + (size_t)lane * H;
#pragma unroll 8
            for (int j = 0; j < H; ++j) acc = fmaf(w[j], sm.vec[0][j], acc);
            z = acc >= 0.f ? acc : acc * a.eval_slope;
        }
        if (lane < F) sm.vec[0][lane] = z;                    // all reads of the LayerNorm vector are done (same wave, in order)
        float lg = -INFINITY;
        if (lane < K) {
            float acc = a.fc3_b[lane];
            const float *w = a.fc3_w + (size_t)lane * F;
            for (int f = 0; f < F; ++f) acc = fmaf(w[f], sm.vec[0][f], acc);
            lg = acc;
            if (b < a.B) a.logits_out[(size_t)b * K + lane] = acc;
        }
        if (a.probs_out) {
            const float mx = wave_max(lg);
            const float e = lane < K ? __expf(lg - mx) : 0.f;
            const float d = wave_sum(e);
            if (lane < K && b < a.B) a.probs_out[(size_t)b * K + lane] = e / d;
        }
        step_barrier<false>(prof);
    }
    prof_store(a.dbg, prof);
}


// ------------------------------------------------------------------------------------------------
// fused TRAIN head (head_train): what nsd_head.hip's head_train_kernel does in a second launch
// (lstm_eeg_model.py:35-39 forward, mean cross-entropy, and the backward of both down to dL/dscore_t and dL/dpooled).
//   * wave 10 ("tpool") pools along the recurrence like the inference tail (online softmax over 8-step chunks) and
//     keeps the raw scores of the trial in LDS; after the last step it runs LayerNorm, fc.0 -> RReLU -> dropout ->
//     fc.3, the loss and the dense backward by itself (vectors of <= 64: one wave, weights staged in LDS);
//   * then ALL waves compute alpha_t, dL/dscore_t = alpha_t (dpooled . top_t - sum_s alpha_s dpooled . top_s) and
//     d attn.weight = sum_t dscore_t top_t from the top rows the saver wave has just written (L2-resident).
// Per-trial gradient slab and outputs: exactly those of HeadArgs (nsd_args.h).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tail_block_sum(float v, float *red, const int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NT_TRAIN / 64; ++w) s += red[w];
    __syncthreads();
    return s;
}

// every wave of the workgroup calls this after the ring-drain barrier of trial b (the barrier also published the
// tpool wave's LDS vectors and, with the saver's vmcnt(0), the top rows in memory)
template <int NB>
__device__ __forceinline__ void train_tail(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int tid, const int b, const int n) {
    const int T = a.T;
    float *sc = sm.sc[n];
    if (ablated(a.ablate, 256)) return;
    // without the residual extension the attention input IS the layer-1 output: that sequence is not saved twice
    const float *top = (a.top ? a.top : a.hseq1) + (size_t)b * T * H;
    const float mx = sm.md[0], rden = sm.md[1];
    float al[2], dd[2];
    float lsd = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int t = tid + q * NT_TRAIN;
        al[q] = 0.f; dd[q] = 0.f;
        if (t < T) {
            al[q] = __expf(sc[t] - mx) * rden;
            const float4 *row = reinterpret_cast<const float4 *>(top + (size_t)t * H);
            float d0 = 0.f, d1 = 0.f;
#pragma unroll 3                                               // (the role's weights stay live across the tail: keep it lean)
            for (int j4 = 0; j4 < H / 4; ++j4) {
                const float4 v = row[j4];
                const float4 p = *reinterpret_cast<const float4 *>(&sm.dp[4 * j4]);
                d0 = fmaf(v.x, p.x, d0); d1 = fmaf(v.y, p.y, d1); d0 = fmaf(v.z, p.z, d0); d1 = fmaf(v.w, p.w, d1);
            }
            dd[q] = d0 + d1;
            lsd = fmaf(al[q], dd[q], lsd);
        }
    }
    const float sdot = tail_block_sum(lsd, sm.red, tid);
    float lb = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int t = tid + q * NT_TRAIN;
        if (t < T) {
            const float ds = al[q] * (dd[q] - sdot);
            a.alpha[(size_t)b * T + t] = al[q];
            a.dscore[(size_t)b * T + t] = ds;
            *reinterpret_cast<float4 *>(a.adpack + ((size_t)b * T + t) * 4) = make_float4(al[q], ds, 0.f, 0.f);
            sc[t] = ds;                                      // the raw score is no longer needed
            lb += ds;
        }
    }
    const float dab = tail_block_sum(lb, sm.red, tid);       // (its barriers also publish sc[] = dscore)
    float *slab = a.hslabs + (size_t)b * a.Ph;
    if (tid == 0) slab[a.o_attn_b] = dab;
    // d attn.weight[j] = sum_t dscore_t top_t[j]: thread (j, part) sums the steps t == part (mod 14)
    if (tid < TT_PARTS * H) {
        const int part = tid / H, j = tid - part * H;
        float s0 = 0.f, s1 = 0.f;
        int t = part;
        for (; t + TT_PARTS < T; t += 2 * TT_PARTS) {
            s0 = fmaf(sc[t], top[(size_t)t * H + j], s0);
            s1 = fmaf(sc[t + TT_PARTS], top[(size_t)(t + TT_PARTS) * H + j], s1);
        }
        if (t < T) s0 = fmaf(sc[t], top[(size_t)t * H + j], s0);
        sm.part[part][j] = s0 + s1;
    }
    __syncthreads();
    if (tid < H) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < TT_PARTS; ++p) s += sm.part[p][tid];
        slab[a.o_attn_w + tid] = s;
    }
    __syncthreads();                                          // sc / part / red are reused by the next trial
}

// What every wave but the train-pooling wave does after the last step of a trial group: for each trial of the group, meet the
// pooling wave (which has just run that trial's dense head alone and left dpooled / the softmax statistics in LDS), then take
// part in the trial's tail.  The first of these barriers is also the "save ring drained" barrier of the group.
template <int NB>
__device__ __forceinline__ void tail_all(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int tid, const int b0) {
#pragma unroll 1
    for (int n = 0; n < NB; ++n) {
        __syncthreads();
        if (b0 + n < a.B) train_tail<NB>(a, sm, tid, b0 + n, n);   // (workgroup-uniform: the padding trial of an odd batch's last group has no tail)
    }
}

template <int NB>
__device__ __forceinline__ void tpool_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int lane, const int n_steps) {
    const int T = a.T, K = a.K, F = a.F;
    PoolRun pr[NB];
    pool_init<NB>(pr[0], a, lane);
#pragma unroll
    for (int n = 1; n < NB; ++n) pr[n] = pr[0];
    // head weights: staged once per workgroup (LDS), per-lane vectors in registers
    stage_head_weights<H, TT_W0S>(a.fc0_w, a.fc3_w, F, K, sm.w0, sm.w3, lane);
    const float lnw = lane < H ? a.ln_w[lane] : 0.f, lnb = lane < H ? a.ln_b[lane] : 0.f;
    const float b0v = lane < F ? a.fc0_b[lane] : 0.f, b3v = lane < K ? a.fc3_b[lane] : 0.f;
    Prof prof = prof_init(a.dbg);
    const int ngrp = (a.B + NB - 1) / NB;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        // per-trial scalars, fetched while the recurrence runs
        float sl_fv[NB], mk_fv[NB];
        int labelv[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int b = b0 + n < a.B ? b0 + n : a.B - 1;      // (a padding trial of the last group: computed, never written)
            float sl_f = (lane < F && a.rrelu_slope) ? a.rrelu_slope[(size_t)b * F + lane] : a.eval_slope;
            float mk_f = (lane < F && a.drop_head) ? a.drop_head[(size_t)b * F + lane] : 1.f;
            if (a.rng.on && lane < F) {                           // same values as nsd_train_masks streams base+1 / base+2
                const uint64_t idx = (uint64_t)b * F + lane;
                const float u = (float)(nsd_rand_u32(a.rng.seed, a.rng.base + 1u, idx) >> 8) * (1.0f / 16777216.0f);
                sl_f = 0.125f + ((float)(1.0 / 3.0) - 0.125f) * u;
                mk_f = nsd_rand_u32(a.rng.seed, a.rng.base + 2u, idx) >= a.rng.thr_head ? a.rng.keep_head : 0.f;
            }
            sl_fv[n] = sl_f; mk_fv[n] = mk_f; labelv[n] = a.labels[b];
            pool_reset(pr[n]);
        }
        step_barrier<false>(prof);
        pool_loop<NB>(pr, sm, lane, T, n_steps, true, prof);
        // ---- per trial: the dense head by this wave alone, then the workgroup's tail (the first barrier is also "ring drained") ----
#pragma unroll 1
        for (int n = 0; n < NB; ++n) {
            const int b = b0 + n;
            const float sl_f = n == 0 ? sl_fv[0] : sl_fv[NB - 1], mk_f = n == 0 ? mk_fv[0] : mk_fv[NB - 1];
            const int label = n == 0 ? labelv[0] : labelv[NB - 1];
            const float pooled = n == 0 ? pr[0].pooled : pr[NB - 1].pooled, den = n == 0 ? pr[0].den : pr[NB - 1].den;
            const float mrun = n == 0 ? pr[0].mrun : pr[NB - 1].mrun;
            // ---- forward of the dense head (same formulas as head_train_kernel) ----
            const bool vb = b < a.B && !ablated(a.ablate, 512);
            const float rden = 1.0f / den;
            const float p = lane < H ? pooled * rden : 0.f;
            if (lane < H && vb) a.pooled[(size_t)b * H + lane] = p;
            const float mu = wave_sum(p) * (1.0f / H);
            const float dlt = lane < H ? p - mu : 0.f;
            const float rstd = 1.0f / sqrtf(wave_sum(dlt * dlt) * (1.0f / H) + 1e-5f);
            const float xh = dlt * rstd;
            const float ln = fmaf(xh, lnw, lnb);
            if (lane < H) { sm.vx[lane] = xh; sm.vln[lane] = ln; }
            float pre = 0.f, z = 0.f;                             // (same wave: LDS queue in order, no barrier needed)
            if (lane < F) {
                float acc = b0v;
                const float *w = &sm.w0[lane * TT_W0S];
#pragma unroll 8
                for (int j = 0; j < H; ++j) acc = fmaf(w[j], sm.vln[j], acc);
                pre = acc;
                if (vb) a.fc0_pre[(size_t)b * F + lane] = acc;
                z = (acc >= 0.f ? acc : acc * sl_f) * mk_f;
                sm.vz[lane] = z;
            }
            float lg = -INFINITY;
            if (lane < K) {
                float acc = b3v;
                for (int f = 0; f < F; ++f) acc = fmaf(sm.w3[lane * F + f], sm.vz[f], acc);
                lg = acc;
                if (vb) a.logits[(size_t)b * K + lane] = acc;
            }
            // ---- mean cross-entropy: dlogits = (softmax - onehot) * scale, without cancellation for the label ----
            const float m2 = wave_max(lg);
            const float e = lane < K ? expf(lg - m2) : 0.f;
            const float d = wave_sum(e);
            const float rest = wave_sum(lane == label ? 0.f : e);
            const float dl = (lane == label ? -rest / d : e / d) * a.scale;
            if (lane < K) sm.vdl[lane] = dl;
            if (lane == label && vb) a.loss[b] = -((lg - m2) - logf(d));
            // ---- backward of the dense head ----
            float *slab = a.hslabs + (size_t)(vb ? b : 0) * a.Ph;
            float dz = 0.f;
            if (lane < F) {
                for (int k = 0; k < K; ++k) dz = fmaf(sm.w3[k * F + lane], sm.vdl[k], dz);
                dz *= mk_f;
                dz = pre >= 0.f ? dz : dz * sl_f;
                sm.vdz[lane] = dz;
                if (vb) slab[a.o_fc0_b + lane] = dz;
            }
            if (vb) {
                for (int e2 = lane; e2 < K * F; e2 += 64) slab[a.o_fc3_w + e2] = sm.vdl[e2 / F] * sm.vz[e2 % F];
                if (lane < K) slab[a.o_fc3_b + lane] = dl;
                for (int e2 = lane; e2 < F * H; e2 += 64) { const int f = e2 / H; slab[a.o_fc0_w + e2] = sm.vdz[f] * sm.vln[e2 - f * H]; }
            }
            float dxh = 0.f;
            if (lane < H) {
                float dv = 0.f;
                for (int f = 0; f < F; ++f) dv = fmaf(sm.w0[f * TT_W0S + lane], sm.vdz[f], dv);
                if (vb) { slab[a.o_ln_w + lane] = dv * xh; slab[a.o_ln_b + lane] = dv; }
                dxh = dv * lnw;
            }
            const float m1 = wave_sum(dxh) * (1.0f / H);
            const float m2b = wave_sum(dxh * xh) * (1.0f / H);
            const float dpl = lane < H ? rstd * (dxh - m1 - xh * m2b) : 0.f;
            sm.dp[lane] = dpl;
            if (lane < H && vb) a.dpooled[(size_t)b * H + lane] = dpl;
            if (lane == 0) { sm.md[0] = mrun; sm.md[1] = rden; }
            __syncthreads();                                       // n == 0: ring drained; publishes dp / md / sc of trial n
            if (b < a.B) train_tail<NB>(a, sm, threadIdx.x, b, n);
        }
    }
    prof_store(a.dbg, prof);
}

// The spare wave.  Without in-kernel random streams it only keeps the barrier count of the workgroup; with rng.on it
// generates the inter-layer dropout multipliers (one row of 48 per step, a whole x chunk ahead of the layer-0 waves that
// read them from sm.ms; chunk 0 is written by the layer-0 waves themselves).  Same values as nsd_train_masks: the
// stream index of (b, t, j) is (b*T + t)*48 + j, advanced by additions only.
template <int NB>
__device__ __forceinline__ void spare_role(const Lstm2FwdArgs &a, FSmem<NB> &sm, const int lane, const int n_steps) {
    Prof prof = prof_init(a.dbg);
    const int T = a.T;
    const int ngrp = (a.B + NB - 1) / NB;
    constexpr int XPL = NB * XCH * 8 / 64;                        // x floats per lane and chunk (4 per trial)
    constexpr int MPL = NB * 384 / 64;                            // explicit-multiplier float4 per lane and chunk (6 per trial)
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NB;
        uint64_t idx[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) idx[n] = ((uint64_t)(b0 + n) * T + XCH) * H + lane;      // row t = XCH
        step_barrier<false>(prof);
        for (int m0 = 0; m0 < n_steps; m0 += XCH) {
            // the next chunk of x (and of the explicit multipliers when the streams are not drawn here): requested now, written to
            // the other half of the staging buffers before the last barrier of this chunk
            float xr[XPL]; float4 mr[MPL];
#pragma unroll
            for (int q = 0; q < XPL; ++q) xr[q] = chunk_x_at<NB>(a, b0, lane + 64 * q, m0 + XCH);
            if (!a.rng.on) {
#pragma unroll
                for (int q = 0; q < MPL; ++q) mr[q] = chunk_mask_at(a, b0, lane + 64 * q, m0 + XCH);
            }
            const int cb = (m0 / XCH) & 1;
            for (int k = 0; k < XCH; ++k) {
                const int m = m0 + k;
                if (a.rng.on && lane < H) {
                    const int t = m + XCH;
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        float v = 1.f;
                        if (b0 + n < a.B && t < T) v = nsd_rand_u32(a.rng.seed, a.rng.base, idx[n]) < a.rng.thr_lstm ? 0.f : a.rng.keep_lstm;
                        sm.ms[cb ^ 1][n][k][lane] = v;
                        idx[n] += H;
                    }
                }
                if (k == XCH - 1) {
#pragma unroll
                    for (int q = 0; q < XPL; ++q) (&sm.xs[cb ^ 1][0][0][0])[lane + 64 * q] = xr[q];
                    if (!a.rng.on) {
#pragma unroll
                        for (int q = 0; q < MPL; ++q) *reinterpret_cast<float4 *>(&sm.ms[cb ^ 1][0][0][0] + 4 * (lane + 64 * q)) = mr[q];
                    }
                }
                step_barrier<false, S_SLEEP>(prof);
            }
        }
        if (a.head_train) tail_all<NB>(a, sm, threadIdx.x, b0);
        else step_barrier<false>(prof);
    }
}

template <int NB>
__global__ __launch_bounds__(NT) void lstm2_fwd48_kernel(Lstm2FwdArgs a) {
    __shared__ __align__(16) FSmem<NB> sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // macro steps 0..T+1, padded to whole x chunks so that every role runs the same number of barriers
    const int n_steps = ((a.T + 2 + XCH - 1) / XCH) * XCH;
    // Role table.  The step time is set by the VALU issue load of the busiest SIMD, and the dispatcher deals the waves
    // of a workgroup round-robin over the 4 SIMDs (waves w and w+4 share one; checked with HW_REG_HW_ID).  So roles
    // are placed by g = wave & 3 (the SIMD) and q = wave >> 2 (the slot on it):
    //     g = 0..2 :  L1 part g | L0 part g | {saver or inference pool, train pool, spare (dropout stream)}[g]
    //     g = 3    :  P part 0  | P part 1  | P part 2
    // i.e. ~155 / 150 / 135 / 135 VALU instructions per step and SIMD, instead of one SIMD carrying an L1, an L0 and
    // a P wave (~180).  s_setprio follows the critical path: L1 > L0 > P > the rest.
    const int g = wave & 3, q = wave >> 2;
#ifdef NSD_FWD48_ONLY_ROLE                                     // resource probe (never built into the library): one role alone
    if (NSD_FWD48_ONLY_ROLE == 1) p_role<NB>(a, sm, q * 64 + lane, n_steps);
    else if (NSD_FWD48_ONLY_ROLE == 2) l1_role<NB>(a, sm, g * 64 + lane, n_steps);
    else if (NSD_FWD48_ONLY_ROLE == 3) l0_role<NB>(a, sm, g * 64 + lane, n_steps);
    else if (NSD_FWD48_ONLY_ROLE == 4) saver_role<NB>(a, sm, lane, n_steps);
    else if (NSD_FWD48_ONLY_ROLE == 5) tpool_role<NB>(a, sm, lane, n_steps);
    else spare_role<NB>(a, sm, lane, n_steps);
    return;
#endif
    if (g == 3)      { __builtin_amdgcn_s_setprio(1); p_role<NB>(a, sm, q * 64 + lane, n_steps); }
    else if (q == 0) { __builtin_amdgcn_s_setprio(3); l1_role<NB>(a, sm, g * 64 + lane, n_steps); }
    else if (q == 1) { __builtin_amdgcn_s_setprio(2); l0_role<NB>(a, sm, g * 64 + lane, n_steps); }
    else if (g == 0) {
        if constexpr (NB == 1) {
            if (a.logits_out) pool_role<NB>(a, sm, lane, n_steps);
            else              saver_role<NB>(a, sm, lane, n_steps);
        } else {
            saver_role<NB>(a, sm, lane, n_steps);               // (inference runs one trial per workgroup: nothing but its latency matters there)
        }
    }
    else if (g == 1 && a.head_train) tpool_role<NB>(a, sm, lane, n_steps);
    else spare_role<NB>(a, sm, lane, n_steps);
}

}  // namespace

int nsd_lstm2_fwd48_launch(const Lstm2FwdArgs &a, int nb, int grid, hipStream_t st) {
    // one trial per workgroup (latency: batches up to one trial per CU, and inference), or two trials per workgroup advancing in
    // lock step (throughput: larger training batches -- two independent dependent chains per wave fill each other's issue gaps)
    if (nb != 1 && nb != 2) { nsd_set_error("lstm2_fwd48: NB=%d not built", nb); return NSD_E_INVALID; }
    if (nb == 2 && a.logits_out) { nsd_set_error("lstm2_fwd48: the inference tail runs one trial per workgroup"); return NSD_E_INVALID; }
    if (a.head_train) {
        if (!nsd_lstm2_fwd48_head_train_fits(a.T, a.F, a.K) || !(a.top || a.hseq1) || a.logits_out) {
            nsd_set_error("lstm2_fwd48: fused train head needs T<=%d, F<=64, K<=%d and the training workspace", TT_TMAX, TT_KMAX);
            return NSD_E_INVALID;
        }
    }
    if (nb == 2) hipLaunchKernelGGL((lstm2_fwd48_kernel<2>), dim3(grid), dim3(NT), 0, st, a);
    else         hipLaunchKernelGGL((lstm2_fwd48_kernel<1>), dim3(grid), dim3(NT), 0, st, a);
    NSD_CHECK_LAUNCH("lstm2_fwd48");
    return NSD_OK;
}

bool nsd_lstm2_fwd48_head_train_fits(int T, int F, int K) { return T <= TT_TMAX && F <= 64 && K <= TT_KMAX && F >= 1 && K >= 1; }
