// nsd_lstm2_fwd48x4.hip -- TRAINING forward of the two-layer H=48 LSTM for batches of several trials per CU (BASELINE configs[3]:
// 1024 trials per GPU): FOUR trials per workgroup, the gate products on the matrix pipe (gfx950).
//
// Replaces self.lstm(x) + the head (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34-39) like nsd_lstm2_fwd48.hip, whose
// one-trial-per-workgroup kernel is a latency design: every mat-vec is VALU work, ~155 instructions per trial-step and SIMD, and
// a second trial per workgroup only fills its issue gaps.  With four trials the product is no longer a mat-vec:
//
//   v_mfma_f32_4x4x1_16B_f32: 16 blocks per instruction, block b = lanes 4b..4b+3, D_b[4x4] += A_b[4x1] * B_b[1x4]
//     A lane (b, i) = W[row = gate i of unit b][k]            -- the weights, resident in VGPRs for the whole launch
//     B lane (b, j) = v[k][trial j]                            -- the same for all 16 blocks: BLGP = 4 + r broadcasts the 16-lane row r
//                                                                of the register, so ONE ds_read_b128 per lane feeds 16 MFMAs
//     D register i of lane (b, j) = gate i of unit b, trial j  -- the four gates of a cell in ONE lane: the cell update is in-lane,
//                                                                no cross-lane traffic, every lane useful
//   (layout and broadcast modifiers probed on the hardware: tools/micro/mfma4x4.hip; exact fp32, an fmaf chain over k.)
//   One instruction = 16 units x 4 gates x 4 trials x 1 k = 512 FLOP in 8 cycles of the SIMD's matrix pipe = the nominal fp32 rate
//   (64 FLOP/clk/SIMD; the pipe takes one per 8 cycles however many waves feed it: SQ_VALU_MFMA_BUSY_CYCLES = 8 per instruction), for
//   ONE issue slot -- 456 per 4-trial step against ~2 900 VALU instructions in the one-trial kernel -- and a VALU wave on the same SIMD
//   runs beside the MFMA stream at its own speed (same probe), so the cells hide behind the other layer's products.
//
// 12 waves, role = f(SIMD g = wave & 3, slot q = wave >> 2); one barrier per macro step m; layer 1 two steps behind layer 0:
//   g 0..2, q 1  "L0"  layer 0, units 16g..16g+15, t = m    : W_ih0 x_t + W_hh0 h0_{t-1} (56 MFMAs), cells, h0 / masked h0 -> LDS
//   g 3,    q 0..2 "P" layer-1 input projection of t = m-1  : rows of group q, W_ih1 in1_t (48 MFMAs) + both biases -> LDS tiles
//   g 0..2, q 0  "L1"  layer 1, units 16g.., t = m-2        : P tile + W_hh1 h1_{t-1} (48), cells
//   g 0,    q 2  "stage"  x and the dropout multipliers of the next 16-step chunk (explicit tensor or the counter stream) -> LDS
//   g 1, 2, q 2  "pool"   fused train head: attention pooling of two trials each as an online softmax along the recurrence,
//                         then LayerNorm / fc / CE / dense backward; otherwise spare
//   (104 / 104 / 104 / 144 MFMAs per step and SIMD: the recurrences' SIMDs also carry the cells and the helper waves.)  The saved activations leave from the lanes that own them: the four gates of a
//   cell are 16 contiguous bytes of gact[b][t][unit][4] -- buffer stores with a scalar time offset, issued one at a time between the
//   MFMAs of the NEXT step (no saver wave, no LDS ring: see "Pending").
// After the last step: the tail of nsd_lstm2_fwd48.hip's fused train head (alpha, dL/dscore, d attn.weight), one trial at a time.
#include "nsd_args.h"
#include "nsd_prof.h"

namespace {

constexpr int H = 48;
constexpr int NTR = 4;            // trials per workgroup = columns of an MFMA block
constexpr int NTHR = 768;
constexpr int VS = 80;            // floats per trial in the operand vectors: bank = 16 j + unit -> conflict-free writes and b128 reads
constexpr int XCH = 16;           // steps per staged chunk of x / multipliers
constexpr int HR = 16;            // h1 ring (two 8-step pooling chunks)
#ifndef NSD_X4_KP
#define NSD_X4_KP 48
#endif
constexpr int KP = NSD_X4_KP;     // k-columns of W_ih1 the P waves take (the rest would ride in the L1 waves); 32 <= KP <= 48, even.
                                  // All 48: the SIMDs of the recurrences also carry the cells and the helper waves -- measured (same box, ablation
                                  // build): KP = 32 / 36 / 40 / 44 / 48 -> 332 / 327 / 323 / 314 / 309 us per launch
constexpr int SCH = 8;
constexpr int TT_TMAX = 1024, TT_KMAX = 8, TT_W0S = 49;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct XSmem {
    float xs[2][NTR][XCH][8];
    float ms[2][NTR][XCH][H];
    float v0[4][NTR][VS];         // h0_t, slot m & 3
    float vm[4][NTR][VS];         // in1_t = h0_t * multiplier, slot m & 3
    float h1[HR][NTR][VS];        // h1_t, slot t & 15 (recurrence operand AND the pooling waves' input)
    float pacc[2][3][64][4];      // P tiles: accumulator registers of lane l of group g, slot m & 1
    float pk[NTR][8];
    float sc[NTR][TT_TMAX];       // raw attention scores of the trial, later dL/dscore
    float w0[64 * TT_W0S];
    float w3[TT_KMAX * 64];
    float vln[NTR][64], vx[NTR][64], vz[NTR][64], vdz[NTR][64], vdl[NTR][64], dp[NTR][64];
    float md[NTR][4];
    float pst[NTR][64];           // pooling state handed to the trial's head wave: [0..47] weighted sum, [48] denominator, [49] running max
    float red[16];
    float part[NTR * 12][H];      // d attn.weight partials of (trial, wave of the team, 16-lane row)
};

// (file scope: the helper roles and the tail are real function calls -- inlined into one body with the compute roles, the kernel
// arguments they keep live push ~330 scalar registers into spills -- and a callee sees this object as LDS, not as a generic pointer)
__shared__ __align__(16) XSmem g_sm;

// The helper roles and the tail are real function calls and get a reference to the kernel's argument block (hipcc keeps a copy of
// the by-value parameter in scratch for that).  Each of them copies the block into a LOCAL first: the step barrier is an asm
// statement with a memory clobber, and a helper that re-read one field per step from scratch arrived ~1 000 cycles late at EVERY
// step barrier (measured: the stage wave -- the whole workgroup waited for it).
constexpr float KC = -2.f * LOG2E_F;
constexpr float INV_KC = 1.f / KC;
__host__ __device__ constexpr float gate_scale(const int g) { return g == 2 ? -2.f * LOG2E_F : -LOG2E_F; }

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, const long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}

// D += A(lane: weight) x B(row r of the register, broadcast to all rows)
template <int BL> __device__ __forceinline__ f32x4 mfma4(const float a, const float b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, BL);
}
__device__ __forceinline__ f32x4 mfma_row(const int r, const float a, const float b, const f32x4 c) {
    switch (r) {                                                   // (r is a constant after unrolling)
    case 0: return mfma4<4>(a, b, c);
    case 1: return mfma4<5>(a, b, c);
    case 2: return mfma4<6>(a, b, c);
    default: return mfma4<7>(a, b, c);
    }
}
// k-columns [K0, K1) of a 48-wide operand whose 16-column pieces sit in bv[0..2] (lane (r, ., j) holds columns 16c + 4r .. + 3 of
// trial j); two accumulator chains (a dependent 4x4x1 chain issues every ~12.8 cycles, two chains every ~9: tools/micro/mfma4x4.hip)
template <int K0, int K1>
__device__ __forceinline__ void mfma_cols(const float (&w)[H], const f32x4 (&bv)[3], f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const int c = k >> 4, r = (k >> 2) & 3, q = k & 3;
        if (k & 1) acc1 = mfma_row(r, w[k], bv[c][q], acc1); else acc0 = mfma_row(r, w[k], bv[c][q], acc0);
    }
}

struct CellOut { float i, f, g, o, c, h; };
// acc: exp2 arguments of the four gates (weights and biases are pre-multiplied by the gate's exp2 scale); cK = KC * c (updated)
__device__ __forceinline__ CellOut cell4_abl(const f32x4 acc, float &cK) {      // timing experiments only: no transcendentals
    CellOut o;
    o.i = acc[0]; o.f = acc[1]; o.g = acc[2]; o.o = acc[3];
    cK = fmaf(o.f, cK, o.i);
    o.c = cK; o.h = o.o * 1e-3f;
    return o;
}
__device__ __forceinline__ CellOut cell4(const f32x4 acc, float &cK) {
    const float ri = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[0]));
    const float rf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[1]));
    const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[2]));
    const float ro = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[3]));
    const float gK = fmaf(2.f * KC, rg, -KC);                      // KC * tanh(pre_g)
    cK = fmaf(rf, cK, ri * gK);
    const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cK));
    CellOut o;
    o.i = ri; o.f = rf; o.g = gK * INV_KC; o.o = ro;
    o.c = cK * INV_KC;
    o.h = fmaf(2.f * ro, rc, -ro);                                 // o * tanh(c)
    return o;
}

#ifndef NSD_X4_ST_AUX
#define NSD_X4_ST_AUX 2          // nt: streamed out, not kept in the L2 (measured: -33 us of 349 per launch at B = 1024)
#endif
__device__ __forceinline__ void st_b128(const rsrc_t r, const unsigned voff, const unsigned soff, const float a, const float b, const float c, const float d) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)}, r, (int)voff, (int)soff, NSD_X4_ST_AUX);
}
__device__ __forceinline__ void st_b32(const rsrc_t r, const unsigned voff, const unsigned soff, const float a) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), r, (int)voff, (int)soff, NSD_X4_ST_AUX);
}
constexpr unsigned VOFF_DROP = 0x80000000u;                        // beyond every descriptor's range: the store of a padding trial is dropped

__device__ __attribute__((noinline)) void tail_all(const Lstm2FwdArgs &a_in, const int tid, const int b0);

// One barrier per macro step.  Raw s_barrier behind lgkmcnt(0): the step's LDS writes are complete, the asynchronous buffer stores of
// the saved activations are NOT waited for (__syncthreads() would drain vmcnt every step).
template <bool RAW_UNUSED = false>
__device__ __forceinline__ void xstep_barrier(Prof &p) {
    if (kProfile && p.on) {
        const long long t = clock64();
        p.work += t - p.last;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const long long t2 = clock64();
        p.wait += t2 - t;
        p.last = t2;
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------
// What a cell lane leaves for the backward pass goes out from the lane itself (the four gates of a cell are 16 contiguous bytes of
// gact[b][t][unit][4]) -- but not where it is formed.  Issued behind the cell, the step's 21 store instructions of the six cell waves hit
// the CU's one vector-memory path in a burst and every wave stood ~130 cycles per instruction IN FRONT of the step barrier (0.25 us
// of a 1.04-us step; records through LDS + saver waves cost the same in LDS traffic: both measured, profiles/r04_x4_kernels.md).
// So a step's values wait in registers and leave one instruction at a time between the MFMAs of the NEXT step, where the wave's issue
// slots are idle anyway; non-temporal: the backward pass reads them after the whole forward, nothing of them should stay in the L2.
// ------------------------------------------------------------------------------------------------
struct Pending { float i, f, g, o, h, c, x; unsigned so4, so16; bool on; };   // x: in1 (layer 0) / unused

// ------------------------------------------------------------------------------------------------
// layer 0
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l0_role(const Lstm2FwdArgs &a, XSmem &sm, const int g, const int lane, const int n_steps, const int grp) {
    const int blk = lane >> 2, j = lane & 3, r = lane >> 4;
    const int unit = 16 * g + blk, T = a.T, C = a.C;
    const int row = (lane & 3) * H + unit;                          // A operand: gate (lane & 3) of this block's unit
    const float gs = gate_scale(lane & 3);
    float wx[8], wh[H];
#pragma unroll
    for (int k = 0; k < 8; ++k) wx[k] = k < C ? gs * a.w_ih0[(size_t)row * C + k] : 0.f;
#pragma unroll
    for (int k = 0; k < H; ++k) wh[k] = gs * a.w_hh0[(size_t)row * H + k];
    f32x4 bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = gate_scale(i) * (a.b_ih0[i * H + unit] + a.b_hh0[i * H + unit]);
    const long bth4 = (long)a.B * T * H * 4;
    const rsrc_t r_g = make_rsrc(a.gact0, bth4 * 4), r_h = make_rsrc(a.hseq0, bth4), r_c = make_rsrc(a.cseq0, bth4), r_i = make_rsrc(a.inseq, bth4);
    Prof prof = prof_init(a.dbg);
    {
        float cK = 0.f;
        const int bt = grp * NTR + j;
        const unsigned vo4 = bt < a.B ? (unsigned)(((size_t)bt * T * H + unit) * 4) : VOFF_DROP;
        const unsigned vo16 = bt < a.B ? vo4 * 4u : VOFF_DROP;
        Pending pd;
        pd.on = false; pd.i = pd.f = pd.g = pd.o = pd.h = pd.c = pd.x = 0.f; pd.so4 = pd.so16 = 0u;
        // zero initial states: h0_{-1} lives in slot 3 of v0, h1_{-1} in slot 15 of the h1 ring (the whole ring: the pooling waves
        // multiply rows beyond T by zero weights, which must not meet stale NaNs)
        for (int e = g * 64 + lane; e < NTR * VS; e += 192) (&sm.v0[3][0][0])[e] = 0.f;
        for (int e = g * 64 + lane; e < HR * NTR * VS; e += 192) (&sm.h1[0][0][0])[e] = 0.f;
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + k;
                prof_mark<-1, false>(prof);
                const bool st_on = pd.on && !ablated(a.ablate, 2048);
                if (m < T && !ablated(a.ablate, 1048576)) {
                    const int tl = m & (XCH - 1), cb = (m >> 4) & 1;
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(&sm.xs[cb][j][tl][4 * (r & 1)]);
                    const float mk = sm.ms[cb][j][tl][unit];
                    f32x4 hv[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) hv[c] = ablated(a.ablate, 524288) ? f32x4{mk, mk, mk, mk} : *reinterpret_cast<const f32x4 *>(&sm.v0[(k + 3) & 3][j][16 * c + 4 * r]);
                    prof_mark<0, true>(prof);        // seg0: LDS operands arrived
                    f32x4 acc0 = bias, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        if (kk & 1) acc1 = mfma_row(kk >> 2, wx[kk], xv[kk & 3], acc1); else acc0 = mfma_row(kk >> 2, wx[kk], xv[kk & 3], acc0);
                    }
                    if (!ablated(a.ablate, 4096)) {
                        mfma_cols<0, 10>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b128(r_g, vo16, pd.so16, pd.i, pd.f, pd.g, pd.o);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<10, 22>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b32(r_h, vo4, pd.so4, pd.h);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<22, 34>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b32(r_c, vo4, pd.so4, pd.c);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<34, 46>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b32(r_i, vo4, pd.so4, pd.x);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<46, H>(wh, hv, acc0, acc1);
                    }
                    prof_mark<1, false>(prof);       // seg1: 56 MFMAs issued
                    const CellOut o = ablated(a.ablate, 16384) ? cell4_abl(acc0 + acc1, cK) : cell4(acc0 + acc1, cK);
                    const float in1 = o.h * mk;
                    sm.v0[k][j][unit] = o.h;
                    sm.vm[k][j][unit] = in1;
                    prof_mark<2, true>(prof);        // seg2: MFMAs retired, cell, h in LDS
                    pd.i = o.i; pd.f = o.f; pd.g = o.g; pd.o = o.o; pd.h = o.h; pd.c = o.c; pd.x = in1;
                    pd.so4 = (unsigned)m * (H * 4); pd.so16 = (unsigned)m * (H * 16); pd.on = true;
                } else {
                    if (st_on) {                     // (the last step's values)
                        st_b128(r_g, vo16, pd.so16, pd.i, pd.f, pd.g, pd.o);
                        st_b32(r_h, vo4, pd.so4, pd.h); st_b32(r_c, vo4, pd.so4, pd.c); st_b32(r_i, vo4, pd.so4, pd.x);
                    }
                    pd.on = false;
                }
                xstep_barrier(prof);
            }
        }
        if (pd.on && !ablated(a.ablate, 2048)) {    // (T a multiple of the step padding: the last step's values are still waiting)
            st_b128(r_g, vo16, pd.so16, pd.i, pd.f, pd.g, pd.o);
            st_b32(r_h, vo4, pd.so4, pd.h); st_b32(r_c, vo4, pd.so4, pd.c); st_b32(r_i, vo4, pd.so4, pd.x);
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// layer-1 input projection (columns k < KP), one step behind layer 0: accumulator tiles for the L1 waves
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void p_role(const Lstm2FwdArgs &a, XSmem &sm, const int g, const int lane, const int n_steps, const int grp) {
    const int blk = lane >> 2, j = lane & 3, r = lane >> 4;
    const int unit = 16 * g + blk, T = a.T;
    const int row = (lane & 3) * H + unit;
    const float gs = gate_scale(lane & 3);
    float wi[H];
#pragma unroll
    for (int k = 0; k < H; ++k) wi[k] = k < KP ? gs * a.w_ih1[(size_t)row * H + k] : 0.f;
    f32x4 bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = gate_scale(i) * (a.b_ih1[i * H + unit] + a.b_hh1[i * H + unit]);
    Prof prof = prof_init(a.dbg);
    {
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + k;
                prof_mark<-1, false>(prof);
                if (m >= 1 && m <= T && !ablated(a.ablate, 1048576)) {
                    f32x4 iv[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) iv[c] = ablated(a.ablate, 524288) ? bias : *reinterpret_cast<const f32x4 *>(&sm.vm[(k + 3) & 3][j][16 * c + 4 * r]);
                    prof_mark<0, true>(prof);
                    f32x4 acc0 = bias, acc1 = {0.f, 0.f, 0.f, 0.f};
                    if (!ablated(a.ablate, 4096)) mfma_cols<0, KP>(wi, iv, acc0, acc1);
                    prof_mark<1, false>(prof);
                    *reinterpret_cast<f32x4 *>(&sm.pacc[k & 1][g][lane][0]) = acc0 + acc1;
                    prof_mark<2, true>(prof);
                }
                xstep_barrier(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// layer 1, two steps behind layer 0
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l1_role(const Lstm2FwdArgs &a, XSmem &sm, const int g, const int lane, const int n_steps, const int grp) {
    const int blk = lane >> 2, j = lane & 3, r = lane >> 4;
    const int unit = 16 * g + blk, T = a.T;
    const int row = (lane & 3) * H + unit;
    const float gs = gate_scale(lane & 3);
    float wt[H], wh[H];                                             // wt: columns KP.. of W_ih1 (the others are the P waves')
#pragma unroll
    for (int k = 0; k < H; ++k) { wt[k] = k >= KP ? gs * a.w_ih1[(size_t)row * H + k] : 0.f; wh[k] = gs * a.w_hh1[(size_t)row * H + k]; }
    const float *h1b = &sm.h1[0][j][4 * r];
    const long bth4 = (long)a.B * T * H * 4;
    const rsrc_t r_g = make_rsrc(a.gact1, bth4 * 4), r_h = make_rsrc(a.hseq1, bth4), r_c = make_rsrc(a.cseq1, bth4);
    const rsrc_t r_t = make_rsrc(a.top ? a.top : a.hseq1, bth4);
    const bool save_top = a.top != nullptr;
    Prof prof = prof_init(a.dbg);
    {
        float cK = 0.f;
        const int bt = grp * NTR + j;
        const unsigned vo4 = bt < a.B ? (unsigned)(((size_t)bt * T * H + unit) * 4) : VOFF_DROP;
        const unsigned vo16 = bt < a.B ? vo4 * 4u : VOFF_DROP;
        Pending pd;
        pd.on = false; pd.i = pd.f = pd.g = pd.o = pd.h = pd.c = pd.x = 0.f; pd.so4 = pd.so16 = 0u;
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + k, t = m - 2;
                prof_mark<-1, false>(prof);
                const bool st_on = pd.on && !ablated(a.ablate, 2048);
                if (t >= 0 && t < T && !ablated(a.ablate, 1048576)) {
                    f32x4 acc0 = *reinterpret_cast<const f32x4 *>(&sm.pacc[(k + 1) & 1][g][lane][0]);     // P tile of step m - 1
                    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
                    f32x4 tv[3], hv[3];
                    tv[2] = *reinterpret_cast<const f32x4 *>(&sm.vm[(k + 2) & 3][j][32 + 4 * r]);         // in1_t, columns 32..47
                    tv[0] = tv[2]; tv[1] = tv[2];
                    const float *hp = h1b + ((t + HR - 1) & (HR - 1)) * (NTR * VS);
#pragma unroll
                    for (int c = 0; c < 3; ++c) hv[c] = ablated(a.ablate, 524288) ? tv[2] : *reinterpret_cast<const f32x4 *>(hp + 16 * c);
                    prof_mark<0, true>(prof);
                    mfma_cols<KP, H>(wt, tv, acc0, acc1);
                    if (!ablated(a.ablate, 4096)) {
                        mfma_cols<0, 8>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b128(r_g, vo16, pd.so16, pd.i, pd.f, pd.g, pd.o);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<8, 22>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b32(r_h, vo4, pd.so4, pd.h);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<22, 36>(wh, hv, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st_on) st_b32(r_c, vo4, pd.so4, pd.c);
                        if (st_on && save_top) st_b32(r_t, vo4, pd.so4, pd.h);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_cols<36, H>(wh, hv, acc0, acc1);
                    }
                    prof_mark<1, false>(prof);
                    const CellOut o = ablated(a.ablate, 16384) ? cell4_abl(acc0 + acc1, cK) : cell4(acc0 + acc1, cK);
                    (&sm.h1[0][0][0])[((t & (HR - 1)) * NTR + j) * VS + unit] = o.h;
                    prof_mark<2, true>(prof);
                    pd.i = o.i; pd.f = o.f; pd.g = o.g; pd.o = o.o; pd.h = o.h; pd.c = o.c;
                    pd.so4 = (unsigned)t * (H * 4); pd.so16 = (unsigned)t * (H * 16); pd.on = true;
                } else {
                    if (st_on) {
                        st_b128(r_g, vo16, pd.so16, pd.i, pd.f, pd.g, pd.o);
                        st_b32(r_h, vo4, pd.so4, pd.h); st_b32(r_c, vo4, pd.so4, pd.c);
                        if (save_top) st_b32(r_t, vo4, pd.so4, pd.h);
                    }
                    pd.on = false;
                }
                xstep_barrier(prof);
            }
        }
        if (pd.on && !ablated(a.ablate, 2048)) {    // (T + 2 a multiple of the step padding: the last step's values are still waiting)
            st_b128(r_g, vo16, pd.so16, pd.i, pd.f, pd.g, pd.o);
            st_b32(r_h, vo4, pd.so4, pd.h); st_b32(r_c, vo4, pd.so4, pd.c);
            if (save_top) st_b32(r_t, vo4, pd.so4, pd.h);
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// stage wave: x and the dropout multipliers, one 16-step chunk ahead
// ------------------------------------------------------------------------------------------------
// (pointers that reach a called function through the argument block are generic to hipcc: a flat load counts in lgkmcnt as well, and
// the step barrier's lgkmcnt(0) would sit out its HBM latency -- say what they are)
typedef const __attribute__((address_space(1))) float *gfloat_p;
typedef const __attribute__((address_space(1))) f32x4 *gf32x4_p;
__device__ __forceinline__ float x_at(const Lstm2FwdArgs &a, const int b0, const int e, const int t0) {     // e in [0, NTR*XCH*8)
    const int n = e / (XCH * 8), tl = (e >> 3) & (XCH - 1), ch = e & 7;
    const int b = b0 + n, t = t0 + tl;
    return (b < a.B && t < a.T && ch < a.C) ? ((gfloat_p)a.x)[((size_t)b * a.T + t) * a.C + ch] : 0.f;
}
__device__ __forceinline__ float4 mask_at(const Lstm2FwdArgs &a, const int b0, const int e, const int t0) {  // e: float4 index in [0, NTR*XCH*12)
    const int n = e / (XCH * 12), rem = e - n * (XCH * 12), tl = rem / 12, q = rem - tl * 12;
    const int b = b0 + n, t = t0 + tl;
    if (a.mask && b < a.B && t < a.T) { const f32x4 v = *(gf32x4_p)(a.mask + ((size_t)b * a.T + t) * H + 4 * q); return make_float4(v[0], v[1], v[2], v[3]); }
    return make_float4(1.f, 1.f, 1.f, 1.f);
}
// the multipliers of time step t for the four trials: 192 values, 3 per lane (same stream as nsd_train_masks: index (b*T + t)*48 + unit)
__device__ __forceinline__ void rng_row(const Lstm2FwdArgs &a, XSmem &sm, const int b0, const int t, const int buf, const int tl, const int lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int v = lane + 64 * i, n = v / H, u = v - n * H;
        const int b = b0 + n;
        float mk = 1.f;
        if (b < a.B && t < a.T) mk = nsd_rand_u32(a.rng.seed, a.rng.base, ((uint64_t)b * a.T + t) * H + u) < a.rng.thr_lstm ? 0.f : a.rng.keep_lstm;
        sm.ms[buf][n][tl][u] = mk;
    }
}
__device__ __attribute__((noinline)) void stage_role(const Lstm2FwdArgs &a_in, const int lane, const int n_steps_in, const int grp_in) {
    XSmem &sm = g_sm;
    const int n_steps = __builtin_amdgcn_readfirstlane(n_steps_in), grp = __builtin_amdgcn_readfirstlane(grp_in);   // (arguments arrive in VGPRs)
    const Lstm2FwdArgs a = uniform_copy(a_in);
    constexpr int XPL = NTR * XCH * 8 / 64;                          // 8 x floats per lane and chunk
    constexpr int MPL = NTR * XCH * 12 / 64;                         // 12 multiplier float4 per lane and chunk
    Prof prof = prof_init(a.dbg);
    {
        const int b0 = grp * NTR;
        // chunk 0
#pragma unroll
        for (int q = 0; q < XPL; ++q) (&sm.xs[0][0][0][0])[lane + 64 * q] = x_at(a, b0, lane + 64 * q, 0);
        if (a.rng.on) {
#pragma unroll 1
            for (int tl = 0; tl < XCH; ++tl) rng_row(a, sm, b0, tl, 0, tl, lane);
        } else {
#pragma unroll
            for (int q = 0; q < MPL; ++q) *reinterpret_cast<float4 *>(&sm.ms[0][0][0][0] + 4 * (lane + 64 * q)) = mask_at(a, b0, lane + 64 * q, 0);
        }
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += XCH) {
            const int cb = (m0 >> 4) & 1;
            float xr[XPL]; float4 mr[MPL];
#pragma unroll
            for (int q = 0; q < XPL; ++q) xr[q] = x_at(a, b0, lane + 64 * q, m0 + XCH);
            if (!a.rng.on) {
#pragma unroll
                for (int q = 0; q < MPL; ++q) mr[q] = mask_at(a, b0, lane + 64 * q, m0 + XCH);
            }
#pragma unroll 1
            for (int k = 0; k < XCH; ++k) {
                if (a.rng.on && !ablated(a.ablate, 8192)) rng_row(a, sm, b0, m0 + XCH + k, cb ^ 1, k, lane);
                if (k == XCH - 1) {
#pragma unroll
                    for (int q = 0; q < XPL; ++q) (&sm.xs[cb ^ 1][0][0][0])[lane + 64 * q] = xr[q];
                    if (!a.rng.on) {
#pragma unroll
                        for (int q = 0; q < MPL; ++q) *reinterpret_cast<float4 *>(&sm.ms[cb ^ 1][0][0][0] + 4 * (lane + 64 * q)) = mr[q];
                    }
                }
                xstep_barrier(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// attention pooling over time as an online softmax over 8-step chunks of the h1 ring (the scheme of nsd_lstm2_fwd48.hip: four
// stages per chunk, one per macro step, DPP / v_readlane reductions); lane = (kq = lane >> 3: step of the chunk, part = lane & 7)
// ------------------------------------------------------------------------------------------------
struct PoolRun {
    float mrun, den, pooled;        // running max, denominator, weighted sum (lane j < 48)
    float sc, pkv, scale;           // chunk in flight
    bool ok, skip;
};
__device__ __forceinline__ void pool_reset(PoolRun &p) { p.mrun = -INFINITY; p.den = 0.f; p.pooled = 0.f; p.skip = true; p.sc = 0.f; p.pkv = 0.f; p.scale = 0.f; p.ok = false; }

__device__ __forceinline__ void pool_stage(const int stage, PoolRun &p, XSmem &sm, const float (&awp)[6], const float ab, const int chunk, const int n,
                                           const int lane, const int T) {
    const int kq = lane >> 3, part = lane & 7;
    const int s0 = SCH * (chunk & 1);
    if (stage == 0) {
        const int t = SCH * chunk + kq;
        const float *rec = &sm.h1[s0 + kq][n][6 * part];
        float sc = 0.f;
#pragma unroll
        for (int u = 0; u < 6; ++u) sc = fmaf(awp[u], rec[u], sc);
        sc = oct_sum(sc);
        p.ok = t < T;
        p.sc = p.ok ? sc + ab : -INFINITY;
        if (p.ok && part == 0) sm.sc[n][t] = p.sc;
    } else if (stage == 1) {
        const float cm = rows_combine_max(fmaxf(p.sc, row_ror<8>(p.sc)));
        const float mnew = fmaxf(p.mrun, cm);
        p.skip = (mnew == -INFINITY);
        if (p.skip) return;
        p.pkv = p.ok ? __expf(p.sc - mnew) : 0.f;
        p.scale = __expf(p.mrun - mnew);
        p.mrun = mnew;
    } else if (stage == 2) {
        if (p.skip) return;
        const float ps = rows_combine_sum(p.pkv + row_ror<8>(p.pkv));
        p.den = fmaf(p.den, p.scale, ps);
        if (part == 0) sm.pk[n][kq] = p.pkv;
    } else {
        if (p.skip) return;
        float acc = p.pooled * p.scale;
        if (lane < H) {
#pragma unroll
            for (int k = 0; k < SCH; ++k) acc = fmaf(sm.pk[n][k], sm.h1[s0 + k][n][lane], acc);
        }
        p.pooled = acc;
    }
}

// the dense head of trial n by one wave alone: lstm_eeg_model.py:38-39 forward, mean CE, and their backward down to dL/dpooled.
// (The gradients of the two weight matrices are outer products of vectors it leaves in LDS: the whole team writes them afterwards.)
struct HeadPre { float lnw, lnb, b0v, b3v, sl_f, mk_f; int label; };
// (requested BEFORE the wave's share of the top rows, so that the head can start while those are still in flight)
__device__ __forceinline__ HeadPre dense_head_pre(const Lstm2FwdArgs &a, const int lane, const int b) {
    const int K = a.K, F = a.F;
    const int bs = b < a.B ? b : a.B - 1;
    HeadPre h;
    h.lnw = lane < H ? a.ln_w[lane] : 0.f; h.lnb = lane < H ? a.ln_b[lane] : 0.f;
    h.b0v = lane < F ? a.fc0_b[lane] : 0.f; h.b3v = lane < K ? a.fc3_b[lane] : 0.f;
    h.sl_f = (lane < F && a.rrelu_slope) ? a.rrelu_slope[(size_t)bs * F + lane] : a.eval_slope;
    h.mk_f = (lane < F && a.drop_head) ? a.drop_head[(size_t)bs * F + lane] : 1.f;
    if (a.rng.on && lane < F) {                                     // same values as nsd_train_masks streams base+1 / base+2
        const uint64_t idx = (uint64_t)bs * F + lane;
        const float u = (float)(nsd_rand_u32(a.rng.seed, a.rng.base + 1u, idx) >> 8) * (1.0f / 16777216.0f);
        h.sl_f = 0.125f + ((float)(1.0 / 3.0) - 0.125f) * u;
        h.mk_f = nsd_rand_u32(a.rng.seed, a.rng.base + 2u, idx) >= a.rng.thr_head ? a.rng.keep_head : 0.f;
    }
    h.label = a.labels[bs];
    return h;
}
__device__ __forceinline__ void dense_head(const Lstm2FwdArgs &a, XSmem &sm, const int lane, const int b, const int n, const HeadPre &hp) {
    const int K = a.K, F = a.F;
    const bool vb = b < a.B;
    const int bs = vb ? b : a.B - 1;
    const float lnw = hp.lnw, lnb = hp.lnb, b0v = hp.b0v, b3v = hp.b3v, sl_f = hp.sl_f, mk_f = hp.mk_f;
    const int label = hp.label;
    const float rden = 1.0f / sm.pst[n][48];
    const float p = lane < H ? sm.pst[n][lane] * rden : 0.f;
    if (lane < H && vb) a.pooled[(size_t)b * H + lane] = p;
    const float mu = wave_sum(p) * (1.0f / H);
    const float dlt = lane < H ? p - mu : 0.f;
    const float rstd = 1.0f / sqrtf(wave_sum(dlt * dlt) * (1.0f / H) + 1e-5f);
    const float xh = dlt * rstd;
    const float ln = fmaf(xh, lnw, lnb);
    if (lane < H) { sm.vx[n][lane] = xh; sm.vln[n][lane] = ln; }
    float pre = 0.f, z = 0.f;                                       // (same wave: the LDS queue is in order, no barrier needed)
    if (lane < F) {
        float acc = b0v;
        const float *w = &sm.w0[lane * TT_W0S];
#pragma unroll 8
        for (int jj = 0; jj < H; ++jj) acc = fmaf(w[jj], sm.vln[n][jj], acc);
        pre = acc;
        if (vb) a.fc0_pre[(size_t)b * F + lane] = acc;
        z = (acc >= 0.f ? acc : acc * sl_f) * mk_f;
        sm.vz[n][lane] = z;
    }
    float lg = -INFINITY;
    if (lane < K) {
        float acc = b3v;
        for (int f = 0; f < F; ++f) acc = fmaf(sm.w3[lane * F + f], sm.vz[n][f], acc);
        lg = acc;
        if (vb) a.logits[(size_t)b * K + lane] = acc;
    }
    const float m2 = wave_max(lg);
    const float e = lane < K ? expf(lg - m2) : 0.f;
    const float d = wave_sum(e);
    const float rest = wave_sum(lane == label ? 0.f : e);
    const float dl = (lane == label ? -rest / d : e / d) * a.scale;
    if (lane < K) sm.vdl[n][lane] = dl;
    if (lane == label && vb) a.loss[b] = -((lg - m2) - logf(d));
    float *slab = a.hslabs + (size_t)bs * a.Ph;
    float dz = 0.f;
    if (lane < F) {
        for (int k = 0; k < K; ++k) dz = fmaf(sm.w3[k * F + lane], sm.vdl[n][k], dz);
        dz *= mk_f;
        dz = pre >= 0.f ? dz : dz * sl_f;
        sm.vdz[n][lane] = dz;
        if (vb) slab[a.o_fc0_b + lane] = dz;
    }
    if (vb && lane < K) slab[a.o_fc3_b + lane] = dl;
    float dxh = 0.f;
    if (lane < H) {
        float dv = 0.f;
        for (int f = 0; f < F; ++f) dv = fmaf(sm.w0[f * TT_W0S + lane], sm.vdz[n][f], dv);
        if (vb) { slab[a.o_ln_w + lane] = dv * xh; slab[a.o_ln_b + lane] = dv; }
        dxh = dv * lnw;
    }
    const float m1 = wave_sum(dxh) * (1.0f / H);
    const float m2b = wave_sum(dxh * xh) * (1.0f / H);
    const float dpl = lane < H ? rstd * (dxh - m1 - xh * m2b) : 0.f;
    sm.dp[n][lane] = dpl;
    if (lane < H && vb) a.dpooled[(size_t)b * H + lane] = dpl;
    if (lane == 0) { sm.md[n][0] = sm.pst[n][49]; sm.md[n][1] = rden; }
}

// ------------------------------------------------------------------------------------------------
// The fused train head after the last step, ALL FOUR TRIALS AT ONCE: team n = waves 3n .. 3n + 2 takes trial n.
//   1. every wave requests its share of the trial's top rows (h1, which the layer-1 waves streamed out): lane (r = lane >> 4, c = lane & 15
//      < 12) of wave w takes the 16-byte piece c of row 12 k + 4 w + r, k = 0 .. TB-1 -- one coalesced 768-byte request per instruction,
//      all TB of a batch in flight; T <= 264 is ONE batch, whose rows then stay in registers for both passes over them
//   2. wave 0 of the team runs the dense head of its trial (LayerNorm .. CE and back to dL/dpooled) while the rows arrive
//   3. pass 1: alpha_t, dd_t = dpooled . top_t (16-lane DPP sums), sdot = sum_t alpha_t dd_t over the team
//   4. dL/dscore_t = alpha_t (dd_t - sdot): one thread per t, coalesced stores
//   5. pass 2: d attn.weight = sum_t dL/dscore_t top_t from the same registers (a second sweep when T > 264), d attn.bias; the outer
//      products of the dense weights' gradients are spread over the whole team
// (Before: the dense heads two per pooling wave, 16.5 us; one row per thread with three requests in flight, then a second sweep with
// eight rows per thread in flight, 25 us of a 292-us launch -- latency, not work.)
// ------------------------------------------------------------------------------------------------
constexpr int TEAM = NTHR / NTR;                                   // 192 threads = 3 waves per trial
constexpr int TB = 16;                                             // requests per batch and lane
constexpr int TBROWS = 12 * TB;                                    // 264 rows per batch and team
__device__ __forceinline__ float row16_sum(float v) { v = oct_sum(v); v += dpp_quad<0x140>(v); return v; }

// Sum over the 192 threads (3 waves) of a TEAM; every team of the workgroup calls it at the same time (the barriers are the workgroup's)
__device__ __forceinline__ float team_sum(float v, float *red /* [NTHR / 64] */, const int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const int w0 = 3 * (tid / TEAM);
    const float s = (red[w0] + red[w0 + 1]) + red[w0 + 2];
    __syncthreads();
    return s;
}

__device__ __forceinline__ void train_tail4(const Lstm2FwdArgs &a, XSmem &sm, const int tid, const int b0) {
    static_assert(sizeof(sm.ms) >= sizeof(float) * NTR * TT_TMAX, "dd_t / dL/dscore_t live in the (dead) multiplier staging area");
    const int T = a.T, K = a.K, F = a.F;
    const int n = __builtin_amdgcn_readfirstlane(tid / TEAM), tt = tid - n * TEAM;    // team = trial, thread of the team
    const int w = __builtin_amdgcn_readfirstlane(tt >> 6), lane = tid & 63, r = lane >> 4, c = lane & 15;
    const int b = b0 + n;
    const bool vb = b < a.B;
    float *sc = sm.sc[n];                                           // raw scores -> alpha_t
    float *ddv = &sm.ms[0][0][0][0] + n * TT_TMAX;                  // dd_t -> dL/dscore_t
    // (range check on the vector offset: rows beyond T and the pieces c >= 12 read as zero, a padding trial has no rows at all)
    const rsrc_t top = make_rsrc(a.hseq1 + (size_t)(vb ? b : 0) * T * H, vb ? (long)T * H * 4 : 0);
    const unsigned vo = c < 12 ? (unsigned)(((4 * w + r) * H + 4 * c) * 4) : VOFF_DROP;
    if (a.defer_att) {
        // The backward pass of this batch runs lstm2_bwd48x4_kernel, which walks the top rows anyway and forms dL/dscore_t, d attn.weight
        // and d attn.bias on its way (sdot = dpooled . pooled needs no pass over the rows).  Left here: the dense head, alpha_t, and the
        // records marked as open.  (Reading the rows back costs this kernel 49 MB per pass at B = 1 024: ~30 us of a 292-us launch.)
        HeadPre hp = {};
        if (w == 0) hp = dense_head_pre(a, lane, b);
        if (w == 0 && !ablated(a.ablate, 4194304)) dense_head(a, sm, lane, b, n, hp);
        __syncthreads();
        const float mx = sm.md[n][0], rden = sm.md[n][1];
        if (vb) {
            for (int t = tt; t < T; t += TEAM) {
                const float al = __expf(sc[t] - mx) * rden;
                a.alpha[(size_t)b * T + t] = al;
                *reinterpret_cast<float4 *>(a.adpack + ((size_t)b * T + t) * 4) = make_float4(al, 0.f, 1.f, 0.f);
            }
            float *slab = a.hslabs + (size_t)b * a.Ph;
            for (int e2 = tt; e2 < K * F; e2 += TEAM) slab[a.o_fc3_w + e2] = sm.vdl[n][e2 / F] * sm.vz[n][e2 % F];
            for (int e2 = tt; e2 < F * H; e2 += TEAM) { const int f = e2 / H; slab[a.o_fc0_w + e2] = sm.vdz[n][f] * sm.vln[n][e2 - f * H]; }
        }
        __syncthreads();
        return;
    }
    const int nb = (T + TBROWS - 1) / TBROWS;
    f32x4 v[TB];
    auto load = [&](const int kb) {
        const unsigned vk = vo + (unsigned)kb * (unsigned)(TBROWS * H * 4);
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(top, (int)(vk + (unsigned)(k * 12 * H * 4)), 0, 0);
            v[k] = f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
        }
    };
    HeadPre hp = {};
    if (w == 0) hp = dense_head_pre(a, lane, b);
    load(0);
    if (w == 0 && !ablated(a.ablate, 4194304)) dense_head(a, sm, lane, b, n, hp);
    __syncthreads();
    const float mx = sm.md[n][0], rden = sm.md[n][1];
    f32x4 dp4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < 12) dp4 = *reinterpret_cast<const f32x4 *>(&sm.dp[n][4 * c]);
    float lsd = 0.f;
    auto pass1 = [&](const int kb) {
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const int t = kb * TBROWS + 12 * k + 4 * w + r;
            const bool ok = t < T;
            const float d = row16_sum(fmaf(v[k][0], dp4[0], v[k][1] * dp4[1]) + fmaf(v[k][2], dp4[2], v[k][3] * dp4[3]));
            const float al = ok ? __expf(sc[ok ? t : 0] - mx) * rden : 0.f;
            if (c == 0) {
                lsd = fmaf(al, d, lsd);
                if (ok) { sc[t] = al; ddv[t] = d; }
            }
        }
    };
    pass1(0);
    for (int kb = 1; kb < nb; ++kb) { load(kb); pass1(kb); }
    const float sdot = team_sum(lsd, sm.red, tid);
    float lb = 0.f;
    for (int t = tt; t < T; t += TEAM) {
        const float al = sc[t], ds = al * (ddv[t] - sdot);
        if (vb) {
            a.alpha[(size_t)b * T + t] = al;
            a.dscore[(size_t)b * T + t] = ds;
            *reinterpret_cast<float4 *>(a.adpack + ((size_t)b * T + t) * 4) = make_float4(al, ds, 0.f, 0.f);
        }
        ddv[t] = ds;
        lb += ds;
    }
    const float dab = team_sum(lb, sm.red, tid);                    // (its barriers also publish ddv[] = dL/dscore)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pass2 = [&](const int kb) {
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const int t = kb * TBROWS + 12 * k + 4 * w + r;
            const bool ok = t < T;
            const float ds = ok ? ddv[ok ? t : 0] : 0.f;
            acc[0] = fmaf(ds, v[k][0], acc[0]); acc[1] = fmaf(ds, v[k][1], acc[1]);
            acc[2] = fmaf(ds, v[k][2], acc[2]); acc[3] = fmaf(ds, v[k][3], acc[3]);
        }
    };
    if (nb == 1) pass2(0);
    else for (int kb = 0; kb < nb; ++kb) { load(kb); pass2(kb); }
    if (c < 12) *reinterpret_cast<f32x4 *>(&sm.part[n * 12 + 4 * w + r][4 * c]) = acc;
    float *slab = a.hslabs + (size_t)(vb ? b : 0) * a.Ph;
    if (vb) {
        if (tt == 0) slab[a.o_attn_b] = dab;
        for (int e2 = tt; e2 < K * F; e2 += TEAM) slab[a.o_fc3_w + e2] = sm.vdl[n][e2 / F] * sm.vz[n][e2 % F];
        for (int e2 = tt; e2 < F * H; e2 += TEAM) { const int f = e2 / H; slab[a.o_fc0_w + e2] = sm.vdz[n][f] * sm.vln[n][e2 - f * H]; }
    }
    __syncthreads();
    if (tt < H && vb) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int p = 0; p < 12; p += 2) { s0 += sm.part[n * 12 + p][tt]; s1 += sm.part[n * 12 + p + 1][tt]; }
        slab[a.o_attn_w + tt] = s0 + s1;
    }
    __syncthreads();
}

// what every wave does after the last step of a trial group when the head is fused
__device__ __attribute__((noinline)) void tail_all(const Lstm2FwdArgs &a_in, const int tid, const int b0_in) {
    XSmem &sm = g_sm;
    const int b0 = __builtin_amdgcn_readfirstlane(b0_in);
    const Lstm2FwdArgs a = uniform_copy(a_in);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's h rows are in memory before the workgroup reads them back
    __syncthreads();                                                // ... and the pooling waves have left the four trials' pooling state in LDS
    if (!ablated(a.ablate, 8388608)) train_tail4(a, sm, tid, b0);
}

// pooling wave `pw` (0 / 1) takes trials 2 pw and 2 pw + 1 of the group.  Chunk c of a trial (t = 8c .. 8c + 7) is complete in the
// ring when macro step 8c + 9 has ended and is overwritten from macro step 8c + 18 on: its eight stage-steps (4 stages x 2 trials)
// run in the macro steps 8c + 10 .. 8c + 17.
__device__ __attribute__((noinline)) void pool_role(const Lstm2FwdArgs &a_in, const int pw_in, const int lane, const int n_steps_in, const int grp_in) {
    XSmem &sm = g_sm;
    const int pw = __builtin_amdgcn_readfirstlane(pw_in), n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);
    (void)grp_in;
    const Lstm2FwdArgs a = uniform_copy(a_in);
    const int T = a.T, K = a.K, F = a.F;
    float awp[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) awp[u] = a.attn_w[6 * (lane & 7) + u];
    const float ab = a.attn_b[0];
    if (pw == 0) {                                                  // head weights: staged once per workgroup
        stage_head_weights<H, TT_W0S>(a.fc0_w, a.fc3_w, F, K, sm.w0, sm.w3, lane);
    }
    Prof prof = prof_init(a.dbg);
    const int last_q = SCH * ((T - 1) / SCH) + SCH - 1;             // last stage-step: stage 3 of the second trial of the last chunk
    {
        PoolRun pr[2];
        pool_reset(pr[0]); pool_reset(pr[1]);
        auto stage_step = [&](const int q) {
            const int chunk = q >> 3, idx = q & 7;
            if (idx < 4) pool_stage(idx, pr[0], sm, awp, ab, chunk, 2 * pw, lane, T);
            else         pool_stage(idx - 4, pr[1], sm, awp, ab, chunk, 2 * pw + 1, lane, T);
        };
        xstep_barrier(prof);
        for (int m = 0; m < n_steps; ++m) {
            const int q = m - 10;
            if (q >= 0 && q <= last_q && !ablated(a.ablate, 2097152)) stage_step(q);
            xstep_barrier(prof);
        }
        for (int q = n_steps - 10 > 0 ? n_steps - 10 : 0; q <= last_q; ++q) stage_step(q);     // (the ring is complete and stable now)
        // the pooling state goes to the head waves of the tail (tail_all: one wave per trial, the four of them at once)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (lane < H) sm.pst[2 * pw + i][lane] = pr[i].pooled;
            if (lane == 0) { sm.pst[2 * pw + i][48] = pr[i].den; sm.pst[2 * pw + i][49] = pr[i].mrun; }
        }
    }
    prof_store(a.dbg, prof);
}

__device__ __forceinline__ void idle_role(const Lstm2FwdArgs &a, XSmem &sm, const int n_steps, const int grp) {
    Prof prof = prof_init(a.dbg);
    {
        xstep_barrier(prof);
        for (int m = 0; m < n_steps; ++m) xstep_barrier(prof);
    }
}

__global__ __launch_bounds__(NTHR) void lstm2_fwd48x4_kernel(Lstm2FwdArgs a) {
    XSmem &sm = g_sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // macro steps 0 .. T+1, padded to whole 16-step chunks: every role runs the same number of barriers
    const int n_steps = ((a.T + 2 + XCH - 1) / XCH) * XCH;
    const int g = wave & 3, q = wave >> 2;                          // SIMD, slot (the dispatcher deals the waves round-robin over the SIMDs)
    // One trial group at a time (a launch with at most four trials per CU has one group per workgroup): the roles reload their
    // weights per group, so that nothing of a role is live across the fused head's tail
    const int ngrp = (a.B + NTR - 1) / NTR;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
#ifdef NSD_X4_ONLY_ROLE                                             // resource probe (never built into the library): one role alone
        if (NSD_X4_ONLY_ROLE == 1) p_role(a, sm, q, lane, n_steps, grp);
        else if (NSD_X4_ONLY_ROLE == 2) l1_role(a, sm, g, lane, n_steps, grp);
        else if (NSD_X4_ONLY_ROLE == 3) l0_role(a, sm, g, lane, n_steps, grp);
        else if (NSD_X4_ONLY_ROLE == 4) stage_role(a, lane, n_steps, grp);
        else if (NSD_X4_ONLY_ROLE == 5) pool_role(a, g - 1, lane, n_steps, grp);
        else tail_all(a, tid, grp * NTR);
        continue;
#endif
        if (g == 3)      { __builtin_amdgcn_s_setprio(1); p_role(a, sm, q, lane, n_steps, grp); }
        else if (q == 0) { __builtin_amdgcn_s_setprio(3); l1_role(a, sm, g, lane, n_steps, grp); }
        else if (q == 1) { __builtin_amdgcn_s_setprio(2); l0_role(a, sm, g, lane, n_steps, grp); }
        else if (g == 0) { __builtin_amdgcn_s_setprio(0); stage_role(a, lane, n_steps, grp); }
        else if (a.head_train) { __builtin_amdgcn_s_setprio(0); pool_role(a, g - 1, lane, n_steps, grp); }
        else idle_role(a, sm, n_steps, grp);
        __builtin_amdgcn_s_setprio(0);
        if (a.head_train) tail_all(a, tid, grp * NTR);
    }
}

}  // namespace

bool nsd_lstm2_fwd48x4_ok(const Lstm2FwdArgs &a) {
    // training launches of the plain two-layer stack; offsets of the saved activations are 32-bit byte offsets
    // (the saved arrays are addressed with 32-bit byte offsets)
    if ((long)a.B * a.T * H * 16 >= 0x7fffffffL) return false;
    return a.hseq0 != nullptr && a.hseq1 && a.cseq0 && a.cseq1 && a.gact0 && a.gact1 && a.inseq && !a.logits_out && !a.residual && a.C <= 8 &&
           (!a.head_train || (a.T <= TT_TMAX && a.F <= 64 && a.K <= TT_KMAX && a.F >= 1 && a.K >= 1));
}

int nsd_lstm2_fwd48x4_launch(const Lstm2FwdArgs &a, int grid, hipStream_t st) {
    if (!nsd_lstm2_fwd48x4_ok(a)) { nsd_set_error("lstm2_fwd48x4: launch outside the kernel's domain"); return NSD_E_INVALID; }
    hipLaunchKernelGGL(lstm2_fwd48x4_kernel, dim3(grid), dim3(NTHR), 0, st, a);
    NSD_CHECK_LAUNCH("lstm2_fwd48x4");
    return NSD_OK;
}
