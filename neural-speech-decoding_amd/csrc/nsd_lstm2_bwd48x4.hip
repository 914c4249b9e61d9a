// nsd_lstm2_bwd48x4.hip -- BPTT of the two-layer H=48 LSTM for batches of several trials per CU (BASELINE configs[3]: 1024 trials
// per GPU): FOUR trials per workgroup, every product on the matrix pipe (gfx950).
//
// Replaces autograd through self.lstm(x) (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:34) like nsd_lstm2_bwd48.hip (one or two
// trials per workgroup: transposed mat-vecs as v_pk_fma_f32 with DPP reductions, ~1 300 cycles per trial-step and issue-bound).
// With four trials every product of a step is a small GEMM with N = 4 and runs as v_mfma_f32_4x4x1_16B_f32 (16 blocks per
// instruction, D_b[4x4] += A_b[4x1] * B_b[1x4]; operand layout, broadcast modifiers and rate: tools/micro/mfma4x4.hip -- the SIMD's
// matrix pipe takes one per 8 cycles = the nominal fp32 rate, and the 912 of a step make this kernel matrix-pipe-bound):
//
//   * the transposed products dh[u][trial] = sum_k W[k][u] da[k][trial] (W_hh1, W_ih1, W_hh0: 48 outputs x 192 k each).  A wave owns
//     16 output units and splits k over the four 16-lane ROWS of the instruction: block (row ks, ub) = units 4ub..4ub+3 x the
//     k-slice 48ks..48ks+47 x 4 trials, A lane (ks, ub, i) = W[k][unit 4ub + i] resident in VGPRs, B lane (ks, ub, j) = da[k][trial j]
//     from ONE ds_read_b128 per four MFMAs; 48 MFMAs per wave and step, then a reduce-scatter of the four slices over the rows with
//     v_permlane32_swap / v_permlane16_swap (3 swaps + 3 adds; semantics probed in tools/micro/permlane_probe.hip): lane (r, ub, j)
//     ends up with dh of unit 4ub + r for trial j -- one cell per lane, every lane useful, the cell's backward in the lane;
//   * the weight gradients dW[192 x 48] += da[.][trial] (x) operand[trial][.] are a GEMM whose K runs over (step, trial): as 4x4x1 outer
//     products they were half of the kernel's matrix-pipe time; they now run as SPLIT-bf16 products on v_mfma_f32_32x32x16_bf16 (x = hi
//     + lo, hi.hi + lo.hi + hi.lo, fp32 accumulation, K = 16 = four steps x four trials; "weight gradients" below): 22.5 instructions of
//     32 cycles per step instead of 480 of 8.
//
// 16 waves, role = f(SIMD g = wave & 3, slot q = wave >> 2); ONE barrier per macro step m:
//   g 0..2, q 0  "C1"  layer-1 recurrence, t = T-1-m, units 16g..: W_hh1^T da1[t+1] -> dh -> cell backward -> da1[t] -> LDS (fp32 vector
//                      for the products + the bf16 halves into the weight gradients' window)
//   g 0..2, q 1  "C0"  layer-0 recurrence, t = T+1-m: W_hh0^T da0[t+1] + multiplier * d_in1[t] -> da0[t] -> LDS
//   g 3,  q 0..2 "X1"  d_in1[t] = W_ih1^T da1[t], t = T-m (what layer 0 receives from layer 1)
//   g 0..2, q 2  "dW"  layer 1: rows 64 g .. + 63 of {dW_hh1 | dW_ih1} (six 32 x 32 tiles);  g 0..1, q 3: layer 0, rows 96 g .. + 95 of
//                      {dW_hh0 | dW_ih0 | -} (six tiles): one column tile of the last complete window per step
//   g 2,  q 3  "rows"  the saved rows h1[t-1], in1[t], h0[t-1] (three 16-byte requests per lane and step, four steps ahead) and x[t]
//                      (staged by the aux wave) -> bf16 halves -> the row windows
//   g 3,  q 3  "aux"   {alpha, dscore} of the layer-1 steps, the x rows and the layer-0 dropout multipliers (explicit tensor or the
//                      counter stream), one 16-step chunk ahead -> LDS; dL/dscore_t itself where the forward kernel left it open
//   (matrix pipe per step and SIMD: 96 x 8 + ~2 x 144 cycles on SIMDs 0..2, 144 x 8 on SIMD 3.)  The saved activations of a cell (16
//   bytes of gates, c[t-1]) are prefetched by the lane that owns the cell, two steps ahead, with buffer loads whose time offset is
//   scalar -- no staging through LDS.
// HBM traffic = saved activations read once + one slab of partial gradients per workgroup at the end.
#include "nsd_args.h"
#include "nsd_prof.h"
#include "nsd_bf16.h"

namespace {

constexpr int H = 48;
constexpr int NTR = 4;
constexpr int NTHR = 1024;
#ifndef NSD_BX4_VSD
#define NSD_BX4_VSD 196
#endif
constexpr int VSD = NSD_BX4_VSD;  // floats per trial of a da vector (k' = 4 unit + gate).  196: the 16 (k-slice, trial) pieces of a B-operand read fall into 16 different 4-bank groups (208: four groups, 4-way conflicts)
constexpr int VS1 = 64;
constexpr int XCH = 16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#ifndef NSD_BX4_DWD
#define NSD_BX4_DWD 4
#endif
constexpr int DWD = NSD_BX4_DWD;
#ifndef NSD_BX4_DW_SLEEP
#define NSD_BX4_DW_SLEEP 0
#endif
#ifndef NSD_BX4_X1_SLEEP
#define NSD_BX4_X1_SLEEP 0
#endif
#ifndef NSD_BX4_B96
#define NSD_BX4_B96 1
#endif
#ifndef NSD_BX4_CHD
#define NSD_BX4_CHD 2
#endif
#ifndef NSD_BX4_CONV
#define NSD_BX4_CONV 0            // who splits da into the bf16 windows: 0 the cell lanes (from their registers, at the step), 1 the rows wave (from the fp32 vectors in LDS, one step behind)
#endif
#ifndef NSD_BX4_VAR
#define NSD_BX4_VAR 0             // timing experiments (never in the library): 1 no d attn.weight in C1, 2 records never open, 4 open without the row requests
#endif
constexpr int WK = 16;            // k rows of a weight-gradient window: 4 macro steps x 4 trials = the K of one v_mfma_f32_32x32x16_bf16
constexpr int ARS = 224;          // bf16 elements per k row of the da windows: 192 + pad (448 B = 192 mod 256: the four k rows of a transposed read's block fall into four different 64-byte bank groups)
constexpr int BRS = 96;           // ... of the row windows (192 B)
constexpr int CHD = NSD_BX4_CHD;   // steps a cell lane requests its saved activations ahead (= unroll of the recurrences' step loop; even, 16 % CHD == 0)   // steps the dW waves' B rows are requested ahead (= unroll of their step loop; 16 % DWD == 0)

struct BSmem {
    float da[2][2][NTR][VSD];     // [slot m & 1][layer][trial][4 unit + gate]
    float din1[2][NTR][VS1];      // [slot m & 1][trial][unit]: W_ih1^T da1 of t = T - m
    float mk[2][NTR][XCH][H];     // layer-0 dropout multipliers of t = T + 1 - m, 16 macro steps per chunk
    float sc[2][NTR][XCH][4];     // {alpha, dscore, open, -} of t = T - 1 - m (open = 1: the record came without dscore, see the aux wave)
    float dpv[NTR][H];            // dL/dpooled of the group's trials (the aux wave's own copy, open records only)
    float xs[2][NTR][XCH][16];    // x[t = T + 1 - m][channel] (channels >= C: zeros), the layer-0 window's columns 48..63
    // split-bf16 operand windows of the weight gradients (see "weight gradients" below): [layer][hi, lo][window m >> 2 & 1][k = 4 (m & 3) + trial][..]
    unsigned short wa[2][2][2][WK][ARS];   // da, k-major: column = k' = 4 unit + gate (written by the cell lanes)
    unsigned short wb[2][2][2][WK][BRS];   // saved rows, k-major: layer 1 {h1[t-1] | in1[t]}, layer 0 {h0[t-1] | x[t] (16 columns) | -} (written by the rows wave)
};
__shared__ __align__(16) BSmem g_bsm;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, const long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
constexpr unsigned VOFF_DROP = 0x80000000u;                        // beyond every descriptor's range: a load returns zeros

// one barrier per macro step: raw s_barrier behind lgkmcnt(0) -- the prefetches in flight (vmcnt) are not waited for
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

__device__ __forceinline__ f32x4 mfma_plain(const float a, const float b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
// outer-product form: A of block `TR` of each group of four blocks, B of the 16-lane row `TR`
template <int TR> __device__ __forceinline__ f32x4 mfma_outer(const float a, const float b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, TR, 4 + TR);
}

// sum of the four 16-lane rows' partial sums, scattered: row r keeps register r (v_permlane32_swap: {x.lo, y.lo} / {x.hi, y.hi};
// v_permlane16_swap: {x.r0, y.r0, x.r2, y.r2} / {x.r1, y.r1, x.r3, y.r3})
__device__ __forceinline__ float rows_reduce_scatter(const f32x4 a) {
    const u32x2 p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[0]), __float_as_uint(a[2]), false, false);
    const u32x2 p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[1]), __float_as_uint(a[3]), false, false);
    const float s02 = __uint_as_float(p02[0]) + __uint_as_float(p02[1]);
    const float s13 = __uint_as_float(p13[0]) + __uint_as_float(p13[1]);
    const u32x2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(s02), __float_as_uint(s13), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// ------------------------------------------------------------------------------------------------
// weight gradients: dW[192 x 48] = sum over (trial, step) of da (x) row -- a GEMM whose K runs over time, fp32 in, fp32 out.  As
// v_mfma_f32_4x4x1 outer products they were half of the kernel's matrix-pipe time (432 + 48 of a step's 912 instructions, 8 cycles each).
// Now: every fp32 operand is split into two bf16 (x = hi + lo + e, hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-18 |x|) and a product is
// THREE bf16 MFMAs with fp32 accumulation, hi.hi + hi.lo + lo.hi (the dropped lo.lo is <= 2^-16 of the product, bf16 products are exact
// in fp32): v_mfma_f32_32x32x16_bf16 with K = 16 = four macro steps x four trials, 3 x 30 tiles per four steps = 22.5 instructions
// of 32 cycles per step instead of 480 of 8.  The operands meet in LDS as k-major bf16 windows (two windows per layer, one being
// filled while the other is read): the cell lanes write the da of their cell (one ds_write_b64 per half: the four gates of a unit
// are adjacent columns), the rows wave the saved rows; the dW waves take both with ds_read_b64_tr_b16 (block = 4 k rows x 16 columns
// -> the 8 consecutive k of one column an MFMA lane wants; nsd_bf16.h).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_bf16(const f32x4 v, u32x2 &hi, u32x2 &lo) {
    hi[0] = pack_bf16x2(v[0], v[1]);
    hi[1] = pack_bf16x2(v[2], v[3]);
    lo[0] = pack_bf16x2(v[0] - bf16_lo(hi[0]), v[1] - bf16_hi(hi[0]));
    lo[1] = pack_bf16x2(v[2] - bf16_lo(hi[1]), v[3] - bf16_hi(hi[1]));
}
// window row of macro step m, trial j; `col` = first of four adjacent columns
__device__ __forceinline__ void put_da(BSmem &sm, const int layer, const int m, const int j, const int col, const f32x4 v) {
    u32x2 hi, lo;
    split_bf16(v, hi, lo);
    const int w = (m >> 2) & 1, k = 4 * (m & 3) + j;
    *reinterpret_cast<u32x2 *>(&sm.wa[layer][0][w][k][col]) = hi;
    *reinterpret_cast<u32x2 *>(&sm.wa[layer][1][w][k][col]) = lo;
}
__device__ __forceinline__ void put_row(BSmem &sm, const int layer, const int m, const int j, const int col, const f32x4 v) {
    u32x2 hi, lo;
    split_bf16(v, hi, lo);
    const int w = (m >> 2) & 1, k = 4 * (m & 3) + j;
    *reinterpret_cast<u32x2 *>(&sm.wb[layer][0][w][k][col]) = hi;
    *reinterpret_cast<u32x2 *>(&sm.wb[layer][1][w][k][col]) = lo;
}
// operand fragment of v_mfma_f32_32x32x16_bf16 from a k-major window: lane l (G = l >> 4, i = l & 15) gives the address of k row
// 8 (G >> 1) + 4 e + (i >> 2), columns c0 + 16 (G & 1) + 4 (i & 3) .. + 3 and receives column c0 + (l & 31), k = 8 (l >> 5) + 4 e + 0..3
template <int RS>
__device__ __forceinline__ bf16x8 window_frag(const unsigned short *win, const int c0, const int lane) {
    const int G = lane >> 4, i = lane & 15;
    const unsigned short *p = win + (8 * (G >> 1) + (i >> 2)) * RS + c0 + 16 * (G & 1) + 4 * (i & 3);
    const s16x4 e0 = lds_read_tr16(reinterpret_cast<const bf16_t *>(p));
    const s16x4 e1 = lds_read_tr16(reinterpret_cast<const bf16_t *>(p + 4 * RS));
    return cat_tr(e0, e1);
}

// W^T rows of a transposed product for lane (ks = lane >> 4, ub = (lane >> 2) & 3, i = lane & 3): w[s] = W[row(k' = 48 ks + s)][16 g + 4 ub + i],
// k' = 4 unit + gate (the order of the da vectors in LDS) -> row of nn.LSTM's [4H][H] weight = gate * 48 + unit
__device__ __forceinline__ void load_wT(const float *w, const int g, const int lane, float (&wv)[H]) {
    const int ks = lane >> 4, col = 16 * g + 4 * ((lane >> 2) & 3) + (lane & 3);
#pragma unroll
    for (int s = 0; s < H; ++s) {
        const int kp = 48 * ks + s;
        wv[s] = w[(size_t)((kp & 3) * H + (kp >> 2)) * H + col];
    }
}

// dh of this lane's cell (unit 16 g + 4 ub + r, trial j) = sum over k' of W[k'][unit] * v[k'][trial]; v = one trial-major da vector set
__device__ __forceinline__ float transposed_product(const float (&wv)[H], const float *vj /* &v[j][48 * ks] */, const int abl = 0) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
    if (ablated(abl, 2)) return wv[0];                              // timing experiments: no product at all
    if (ablated(abl, 4)) {                                          // ... the MFMAs without the LDS reads
        const f32x4 c4 = {wv[1], wv[2], wv[3], wv[4]};
#pragma unroll
        for (int s = 0; s < H; s += 2) { acc0 = mfma_plain(wv[s], c4[s & 3], acc0); acc1 = mfma_plain(wv[s + 1], c4[(s + 1) & 3], acc1); }
        return rows_reduce_scatter(acc0 + acc1);
    }
#pragma unroll
    for (int qb = 0; qb < 12; qb += 4) {                              // three batches of four reads: 16 B-operand registers live at a time
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4 *>(vj + 4 * (qb + q));
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                   // four accumulator chains (a dependent 4x4x1 issues every ~12.8 cycles,
            acc0 = mfma_plain(wv[4 * (qb + q) + 0], bq[q][0], acc0);    // independent ones every ~8.8)
            acc1 = mfma_plain(wv[4 * (qb + q) + 1], bq[q][1], acc1);
            acc2 = mfma_plain(wv[4 * (qb + q) + 2], bq[q][2], acc2);
            acc3 = mfma_plain(wv[4 * (qb + q) + 3], bq[q][3], acc3);
        }
    }
    return rows_reduce_scatter((acc0 + acc1) + (acc2 + acc3));
}

// ------------------------------------------------------------------------------------------------
// recurrences: layer = 1 (t = T-1-m) or 0 (t = T+1-m)
// ------------------------------------------------------------------------------------------------
// (every role is a real function call with its own register allocation -- inlined into one body, hipcc spilled the dW accumulators
// inside the step loop -- and works on a LOCAL copy of the argument block: the step barrier is an asm statement with a memory clobber)
template <int LAYER>
__device__ __attribute__((noinline)) void chain_role(const Lstm2BwdArgs &a_in, const int g_in, const int lane, const int n_steps_in) {
    BSmem &sm = g_bsm;
    const int g = __builtin_amdgcn_readfirstlane(g_in), n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);    // (arguments arrive in VGPRs: uniform_copy's comment)
    const Lstm2BwdArgs a = uniform_copy(a_in);
    const int r = lane >> 4, ub = (lane >> 2) & 3, j = lane & 3;
    const int u = 16 * g + 4 * ub + r;                              // this lane's cell after the reduce-scatter
    const int T = a.T, B = a.B;
    float wv[H];
    load_wT(LAYER == 0 ? a.w_hh0 : a.w_hh1, g, lane, wv);
    const float awj = a.attn_w[u];
    const long bth4 = (long)B * T * H * 4;
    const rsrc_t r_g = make_rsrc(LAYER == 0 ? a.gact0 : a.gact1, bth4 * 4), r_c = make_rsrc(LAYER == 0 ? a.cseq0 : a.cseq1, bth4);
    const bool masked = LAYER == 0 && (a.mask != nullptr || a.rng.on);
    float db[4] = {0.f, 0.f, 0.f, 0.f};
    Prof prof = prof_init(a.dbg);
    const int ngrp = (B + NTR - 1) / NTR;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b = grp * NTR + j;
        const bool vb = b < B;
        const unsigned vo4 = vb ? (unsigned)(((size_t)b * T * H + u) * 4) : VOFF_DROP;
        const unsigned vo16 = vb ? vo4 * 4u : VOFF_DROP;
        const float dpj = (LAYER == 1 && vb) ? a.dpooled[(size_t)b * H + u] : 0.f;
        float dc = 0.f;
        float attw = 0.f, open_rec = 0.f;                           // layer 1, open records: d attn.weight[u] of trial j = sum_t dscore_t h1_t[u]
        // time index of macro step m, clamped for the prefetches (values of inactive steps are never used)
        auto t_of = [&](const int m) { return LAYER == 1 ? T - 1 - m : T + 1 - m; };
        auto clampt = [&](const int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };
        f32x4 gq[CHD];
        float cq[CHD];                                             // c[t-1] of the step
        auto prefetch = [&](const int m, f32x4 &gv, float &cv) {
            const int t = clampt(t_of(m)), tp = clampt(t_of(m) - 1);
            gv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_g, (int)vo16, t * (H * 16), 0));
            cv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_c, (int)vo4, tp * (H * 4), 0));
        };
        float ct = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_c, (int)vo4, (T - 1) * (H * 4), 0));     // c[T-1] of the first step
#pragma unroll
        for (int k = 0; k < CHD; ++k) prefetch(k, gq[k], cq[k]);
        // (weights in registers before the loop: otherwise hipcc carries "loads pending" into the loop and every wait inside it becomes vmcnt(0))
#pragma unroll
        for (int s4 = 0; s4 < H; s4 += 8)
            asm volatile("" : "+v"(wv[s4]), "+v"(wv[s4 + 1]), "+v"(wv[s4 + 2]), "+v"(wv[s4 + 3]), "+v"(wv[s4 + 4]), "+v"(wv[s4 + 5]), "+v"(wv[s4 + 6]), "+v"(wv[s4 + 7]));
        if (LAYER == 0 && g == 0) {                                 // zero the da slots and d_in1 of the group (one wave: 3 328 + 512 floats)
            for (int e = lane; e < 2 * 2 * NTR * VSD; e += 64) (&sm.da[0][0][0][0])[e] = 0.f;
            for (int e = lane; e < 2 * NTR * VS1; e += 64) (&sm.din1[0][0][0])[e] = 0.f;
        }
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += CHD) {
#pragma unroll
            for (int k = 0; k < CHD; ++k) {
                const int m = m0 + k;
                const int t = t_of(m);
                const bool active = t >= 0 && t < T, prev_active = t + 1 >= 0 && t + 1 < T;
                prof_mark<-1, false>(prof);
                // ---- what does not depend on the recurrence: the derivative factors of this cell, the gradient arriving from above
                const float cprev = t > 0 ? cq[k] : 0.f;
                const float ig = gq[k][0], fg = gq[k][1], gg = gq[k][2], og = gq[k][3];
                const float tc = fast_tanh(ct);
                float wq = og * (1.f - tc * tc);                    // d c_t / d h_t
                float ht = tc * og;                                 // h_t of this cell (o * tanh(c): what the forward pass saved, to rounding)
                float Fi = gg * ig * (1.f - ig), Ff = cprev * fg * (1.f - fg), Fg = ig * (1.f - gg * gg), Fo = ht * (1.f - og);
                float fgk = fg, cpk = cprev;
                // (the step's saved values are consumed: pinned here, so that the loads below may land in the SAME registers -- with the
                // old values still live hipcc rotates the prefetch registers with copies in the loop latch and waits for the loads there)
                if (LAYER == 1) asm volatile("" : "+v"(wq), "+v"(Fi), "+v"(Ff), "+v"(Fg), "+v"(Fo), "+v"(fgk), "+v"(cpk), "+v"(ht));
                else            asm volatile("" : "+v"(wq), "+v"(Fi), "+v"(Ff), "+v"(Fg), "+v"(Fo), "+v"(fgk), "+v"(cpk));
                prof_mark<0, false>(prof);                          // seg0: derivative factors
                if (!ablated(a.ablate, 16)) prefetch(m + CHD, gq[k], cq[k]);     // CHD steps ahead
                prof_mark<1, false>(prof);                          // seg1: prefetch issued
                float dout;
                if (LAYER == 1) {
                    const f32x4 ad = *reinterpret_cast<const f32x4 *>(&sm.sc[(m >> 4) & 1][j][m & (XCH - 1)][0]);
                    dout = fmaf(ad[0], dpj, ad[1] * awj);
                    if (!(NSD_BX4_VAR & 1)) {
                    attw = fmaf(ad[1], ht, attw);                   // (dscore is zero on inactive steps and padding trials)
                    open_rec = fmaxf(open_rec, ad[2]);
                    }
                } else {
                    const float mkv = masked ? sm.mk[(m >> 4) & 1][j][m & (XCH - 1)][u] : 1.f;
                    dout = sm.din1[(k + 1) & 1][j][u] * mkv;        // written by the X1 waves at macro step m - 1
                }
                // ---- the recurrence
                float rec = 0.f;
                prof_mark<2, true>(prof);                           // seg2: dout operands (LDS) arrived
                if (prev_active) rec = transposed_product(wv, &sm.da[(k + 1) & 1][LAYER][j][48 * r], a.ablate);
                prof_mark<3, false>(prof);                          // seg3: product + reduce-scatter
                f32x4 dav = {0.f, 0.f, 0.f, 0.f};
                if (active && vb) {
                    const float dht = dout + rec;
                    const float dct = fmaf(dht, wq, dc);
                    dav = f32x4{dct * Fi, dct * Ff, dct * Fg, dht * Fo};
                    dc = dct * fgk;
                    db[0] += dav[0]; db[1] += dav[1]; db[2] += dav[2]; db[3] += dav[3];
                }
                if (active) ct = cpk;                               // c[t-1] is the cell state of the next step handled
                *reinterpret_cast<f32x4 *>(&sm.da[k & 1][LAYER][j][4 * u]) = dav;      // (zeros for inactive steps / padding trials: dW and X1 add nothing)
                if (!NSD_BX4_CONV) put_da(sm, LAYER, m, j, 4 * u, dav);     // the weight gradients' operand: row (step, trial) of the window being filled
                prof_mark<4, true>(prof);                           // seg4: cell backward, da in LDS
                xstep_barrier(prof);
            }
        }
        if (LAYER == 1 && vb && open_rec != 0.f) a.hslabs[(size_t)b * a.Ph + a.o_attn_w + u] = attw;
    }
    prof_store(a.dbg, prof);
    // bias gradients: sum over the four trials of the quad, lane j == 0 writes (b_ih and b_hh get the same sum)
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float s = quad_sum(db[q]);
        if (j == 0) {
            slab[(LAYER == 0 ? a.o_b_ih0 : a.o_b_ih1) + q * H + u] = s;
            slab[(LAYER == 0 ? a.o_b_hh0 : a.o_b_hh1) + q * H + u] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// X1: d_in1[t] = W_ih1^T da1[t], t = T - m (da1 written at macro step m - 1) -> LDS for the layer-0 recurrence of macro step m + 1
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) void x1_role(const Lstm2BwdArgs &a_in, const int g_in, const int lane, const int n_steps_in) {
    BSmem &sm = g_bsm;
    const int g = __builtin_amdgcn_readfirstlane(g_in), n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);
    const Lstm2BwdArgs a = uniform_copy(a_in);
    const int r = lane >> 4, ub = (lane >> 2) & 3, j = lane & 3;
    const int u = 16 * g + 4 * ub + r;
    const int T = a.T, B = a.B;
    float wv[H];
    load_wT(a.w_ih1, g, lane, wv);
    Prof prof = prof_init(a.dbg);
    const int ngrp = (B + NTR - 1) / NTR;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
#pragma unroll
        for (int s4 = 0; s4 < H; s4 += 8)
            asm volatile("" : "+v"(wv[s4]), "+v"(wv[s4 + 1]), "+v"(wv[s4 + 2]), "+v"(wv[s4 + 3]), "+v"(wv[s4 + 4]), "+v"(wv[s4 + 5]), "+v"(wv[s4 + 6]), "+v"(wv[s4 + 7]));
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + k, t1p = T - m;
                if (NSD_BX4_X1_SLEEP) __builtin_amdgcn_s_sleep(NSD_BX4_X1_SLEEP);
                if (t1p >= 0 && t1p < T) sm.din1[k & 1][j][u] = transposed_product(wv, &sm.da[(k + 1) & 1][1][j][48 * r], a.ablate);
                xstep_barrier(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// rows wave: the saved rows of macro step m -- layer 1 (t = T-1-m): h1[t-1] | in1[t]; layer 0 (t = T+1-m): h0[t-1] | x[t] -- requested
// DWD steps ahead (three 16-byte buffer loads per lane, lanes 0..47 = (trial, 4 columns); x comes staged from the aux wave, lanes
// 48..63), split into bf16 halves and written into the row windows at the k row of (step, trial): what the cell lanes do for da.
// Rows of inactive steps, of t - 1 < 0 and of padding trials read as zeros (switched off at the ADDRESS).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) void rows_role(const Lstm2BwdArgs &a_in, const int lane, const int n_steps_in) {
    BSmem &sm = g_bsm;
    const int n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);
    const Lstm2BwdArgs a = uniform_copy(a_in);
    const int T = a.T, B = a.B;
    const long bth4 = (long)B * T * H * 4;
    const rsrc_t r_h1 = make_rsrc(a.hseq1, bth4), r_in1 = make_rsrc(a.in1seq, bth4), r_h0 = make_rsrc(a.hseq0, bth4);
    const bool hb = lane < 48;                                      // lanes with a piece of the HBM rows
    const int j = hb ? lane / 12 : (lane - 48) >> 2;                // trial
    const int c4 = hb ? lane - 12 * j : lane & 3;                   // piece (four columns) of the 48-float row / of the 16 staged x floats
    const int col0 = hb ? 4 * c4 : 48 + 4 * c4;                     // third piece: layer-0 window column (h0 | x)
    Prof prof = prof_init(a.dbg);
    const int ngrp = (B + NTR - 1) / NTR;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b = grp * NTR + j;
        const unsigned vo = (hb && b < B) ? (unsigned)(((size_t)b * T * H + 4 * c4) * 4) : VOFF_DROP;
        auto prefetch = [&](const int m, f32x4 (&bv)[3]) {
            const int t1 = T - 1 - m, t0 = T + 1 - m;
            const bool ok1 = t1 >= 0 && t1 < T, ok0 = t0 >= 0 && t0 < T;
            auto cl = [&](const int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };
            bv[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_h1, (int)((ok1 && t1 >= 1) ? vo : VOFF_DROP), cl(t1 - 1) * (H * 4), 0));
            bv[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_in1, (int)(ok1 ? vo : VOFF_DROP), cl(t1) * (H * 4), 0));
            bv[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_h0, (int)((ok0 && t0 >= 1) ? vo : VOFF_DROP), cl(t0 - 1) * (H * 4), 0));
        };
        f32x4 bq[DWD][3];
#pragma unroll
        for (int k = 0; k < DWD; ++k) prefetch(k, bq[k]);
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += DWD) {
#pragma unroll
            for (int k = 0; k < DWD; ++k) {
                const int m = m0 + k;
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(&sm.xs[(m >> 4) & 1][j][m & (XCH - 1)][hb ? 0 : 4 * c4]);      // (lanes >= 48 use it)
                if (hb) {
                    put_row(sm, 1, m, j, 4 * c4, bq[k][0]);
                    put_row(sm, 1, m, j, 48 + 4 * c4, bq[k][1]);
                }
                put_row(sm, 0, m, j, col0, hb ? bq[k][2] : xv);
                if (NSD_BX4_CONV) {                                  // da of macro step m - 1, both layers: 384 pieces of four columns, six per lane
                    if (m >= 1) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const int rem = lane + 64 * (i % 3), jj = rem / 48, pc = rem - 48 * jj;
                            put_da(sm, i / 3, m - 1, jj, 4 * pc, *reinterpret_cast<const f32x4 *>(&sm.da[(m + 1) & 1][i / 3][jj][4 * pc]));
                        }
                    }
                    if (m == n_steps - 1) {                          // (the last macro step is an inactive one: zeros, nobody converts it)
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const int rem = lane + 64 * (i % 3), jj = rem / 48, pc = rem - 48 * jj;
                            put_da(sm, i / 3, m, jj, 4 * pc, f32x4{0.f, 0.f, 0.f, 0.f});
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!ablated(a.ablate, 8)) prefetch(m + DWD, bq[k]);     // (behind the last use of these registers: the loads land in place)
                xstep_barrier(prof);
            }
        }
    }
    prof_store(a.dbg, prof);
}

// ------------------------------------------------------------------------------------------------
// dW waves.  Five waves of six 32 x 32 accumulator tiles: d = 0..2 layer 1, rows k' = 64 d .. + 63 (two row tiles) x the 96 columns
// {dW_hh1 | dW_ih1} (three column tiles); d = 3, 4 layer 0, rows 96 (d - 3) .. + 95 (three row tiles) x the 64 columns {dW_hh0 | dW_ih0
// (16, C used) | -} (two column tiles).  The window of macro steps 4w .. 4w + 3 is complete behind the barrier of step 4w + 3 and is
// overwritten from step 4w + 8 on: its products are spread over steps 4w + 4 .. 4w + 7, one column tile per step (three MFMAs per tile:
// hi.hi, lo.hi, hi.lo); the last window of a trial group is taken behind the loop.
// ------------------------------------------------------------------------------------------------
template <int LAYER>
__device__ __attribute__((noinline)) void dw_role(const Lstm2BwdArgs &a_in, const int d_in, const int lane, const int n_steps_in) {
    BSmem &sm = g_bsm;
    constexpr int NM = LAYER == 1 ? 2 : 3, NN = LAYER == 1 ? 3 : 2;   // row / column tiles of this wave
    const int d = __builtin_amdgcn_readfirstlane(d_in), n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);
    const Lstm2BwdArgs a = uniform_copy(a_in);
    const int B = a.B;
    const int row0 = 32 * NM * d;                                   // first k' of this wave
    f32x16 acc[NM][NN];
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
#pragma unroll
        for (int ni = 0; ni < NN; ++ni) acc[mi][ni] = zero16();
    // one column tile of window w: B fragments once, A fragments per row tile
    auto part = [&](const int w, const int ni) {
        const bf16x8 bh = window_frag<BRS>(&sm.wb[LAYER][0][w][0][0], 32 * ni, lane);
        const bf16x8 bl = window_frag<BRS>(&sm.wb[LAYER][1][w][0][0], 32 * ni, lane);
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
            const bf16x8 ah = window_frag<ARS>(&sm.wa[LAYER][0][w][0][0], row0 + 32 * mi, lane);
            const bf16x8 al = window_frag<ARS>(&sm.wa[LAYER][1][w][0][0], row0 + 32 * mi, lane);
            if (!ablated(a.ablate, 1)) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mi][ni], 0, 0, 0);
            }
        }
    };
    Prof prof = prof_init(a.dbg);
    const int ngrp = (B + NTR - 1) / NTR;
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += 4) {
            const int w = ((m0 >> 2) + 1) & 1;                      // the window completed at macro step m0 - 1
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (NSD_BX4_DW_SLEEP) __builtin_amdgcn_s_sleep(NSD_BX4_DW_SLEEP);
                if (!NSD_BX4_CONV) { if (m0 >= 4 && k < NN) part(w, k); }
                else               { if (m0 >= 4 && k >= 1 && k - 1 < NN) part(w, k - 1); }     // (the window is complete one step later)
                xstep_barrier(prof);
            }
        }
        const int wl = ((n_steps >> 2) + 1) & 1;                    // the group's last window
#pragma unroll
        for (int k = 0; k < NN; ++k) part(wl, k);
    }
    prof_store(a.dbg, prof);
    // accumulator tile -> slab: register r of lane l = dW[k' = row0 + 32 mi + mfma32_row(r, l)][column 32 ni + (l & 31) of the window]
    float *slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int mi = 0; mi < NM; ++mi)
#pragma unroll
        for (int ni = 0; ni < NN; ++ni) {
            const int c = 32 * ni + (lane & 31);
            long base; int cc, ld;
            if (LAYER == 1) { base = c < H ? a.o_w_hh1 : a.o_w_ih1; cc = c < H ? c : c - H; ld = H; }
            else            { base = c < H ? a.o_w_hh0 : a.o_w_ih0; cc = c < H ? c : c - H; ld = c < H ? H : a.C; }
            const bool okc = LAYER == 1 || c < H + a.C;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kp = row0 + 32 * mi + mfma32_row(r, lane);
                if (okc) slab[base + (size_t)((kp & 3) * H + (kp >> 2)) * ld + cc] = acc[mi][ni][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// aux wave: {alpha, dscore} of the layer-1 steps and the layer-0 dropout multipliers, one 16-step chunk ahead.
// OPEN records ({alpha_t, -, 1, -}: the forward pass of this batch was lstm2_fwd48x4_kernel with the head fused, which does not walk
// the top rows a second time): lane (trial n, step s) reads the 48 floats of top_t with the record and closes it,
//   dL/dscore_t = alpha_t (dpooled . top_t - dpooled . pooled)       (= alpha_t (dd_t - sum_s alpha_s dd_s): pooled IS sum_s alpha_s top_s),
// writes it to the workspace and sums d attn.bias; d attn.weight = sum_t dscore_t h1_t is formed by the layer-1 recurrence's lanes,
// which hold h1_t = o_t tanh(c_t) of their cell.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) f32x4 *gf32x4_p;
__device__ __attribute__((noinline)) void aux_role(const Lstm2BwdArgs &a_in, const int lane_in, const int n_steps_in) {
    BSmem &sm = g_bsm;
    // `lane` is made opaque once per chunk (an empty asm): what the request lambdas derive from it -- 28 + 12 per-lane addresses -- is then
    // recomputed per chunk (a few hundred integer instructions per 16 steps) instead of being hoisted out of the chunk loop into registers
    // this wave does not have: spilled, every request of a chunk waited for a scratch reload first (measured: +85 us per launch)
    int lane = lane_in;
    const int n_steps = __builtin_amdgcn_readfirstlane(n_steps_in);
    const Lstm2BwdArgs a = uniform_copy(a_in);
    const int T = a.T, B = a.B;
    Prof prof = prof_init(a.dbg);
    const int ngrp = (B + NTR - 1) / NTR;
    // chunk c = macro steps 16c .. 16c + 15 into buffer c & 1:  sc: lane (n = lane >> 4, s = lane & 15) one 16-byte record;
    // multipliers: 4 trials x 16 steps x 12 float4 = 768 float4, 12 per lane (explicit tensor), or 3 hashes per lane and step
    auto sc_at = [&](const int b0, const int c) -> f32x4 {
        const int n = lane >> 4, s = lane & 15, b = b0 + n, t = T - 1 - (16 * c + s);
        if (b < B && t >= 0 && t < T) return *(gf32x4_p)(a.dsc_pack + ((size_t)b * T + t) * 4);
        return f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // (requests as buffer loads over a scalar base with ONE 32-bit offset each: no 64-bit address arithmetic, no branch around the
    // request -- a row outside the trial is sent out of the descriptor's range and reads as zeros, which no active step uses)
    const rsrc_t r_mask = make_rsrc(a.mask, a.mask ? (long)B * T * H * 4 : 0);
    auto mask_at = [&](const int b0, const int c, const int e) -> f32x4 {      // e: float4 index in [0, NTR*XCH*12)
        const int n = e / (XCH * 12), rem = e - n * (XCH * 12), s = rem / 12, q = rem - s * 12;
        const int b = b0 + n, t = T + 1 - (16 * c + s);
        if (!a.mask) return f32x4{1.f, 1.f, 1.f, 1.f};
        const unsigned off = (b < B && t >= 0 && t < T) ? (unsigned)(((b * T + t) * H + 4 * q) * 4) : VOFF_DROP;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_mask, (int)off, 0, 0));
    };
    auto rng_row = [&](const int b0, const int m, const int buf) {            // the 192 multipliers of macro step m: 3 per lane
        const int t = T + 1 - m;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int v = lane + 64 * i, n = v / H, uu = v - n * H;
            const int b = b0 + n;
            float mkv = 1.f;
            if (b < B && t >= 0 && t < T) mkv = nsd_rand_u32(a.rng.seed, a.rng.base, ((uint64_t)b * T + t) * H + uu) < a.rng.thr_lstm ? 0.f : a.rng.keep_lstm;
            sm.mk[buf][n][m & (XCH - 1)][uu] = mkv;
        }
    };
    // x rows of chunk c: 4 trials x 16 steps x 16 floats (channels >= C: zeros) = 256 float4, 4 per lane
    const rsrc_t r_x = make_rsrc(a.x, (long)B * T * a.C * 4);
    auto x_at = [&](const int b0, const int c, const int e) -> f32x4 {        // e: float4 index in [0, NTR*XCH*4)
        const int n = e >> 6, s = (e >> 2) & 15, q = e & 3, b = b0 + n, t = T + 1 - (16 * c + s);
        const bool ok = b < B && t >= 0 && t < T;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                 // (rows of C floats: dword requests; channels >= C read as zero)
            const unsigned off = (ok && 4 * q + i < a.C) ? (unsigned)(((b * T + t) * a.C + 4 * q + i) * 4) : VOFF_DROP;
            v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_x, (int)off, 0, 0));
        }
        return v;
    };
    constexpr int MPL = NTR * XCH * 12 / 64;
    const int an = lane_in >> 4, as = lane_in & 15;
    // piece q of top_t of this lane's record in chunk c: ONE 32-bit offset per lane and chunk over a scalar base (a 64-bit address per
    // piece, hoisted out of the chunk loop by hipcc, cost the wave its registers); clamped: a record outside the trial is not used
    const rsrc_t r_top = make_rsrc(a.hseq1, (long)B * T * H * 4);
    auto row_off = [&](const int b0, const int c) -> unsigned {
        int b = b0 + an, t = T - 1 - (16 * c + as);
        b = b < B ? b : B - 1;
        t = t < 0 ? 0 : t;
        return (unsigned)((b * T + t) * (H * 4));
    };
    auto row_at = [&](const unsigned off, const int q) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_top, (int)(off + 16u * (unsigned)q), 0, 0));
    };
    auto dot_piece = [&](const f32x4 r, const int q, float &d0, float &d1) {
        const f32x4 p = *reinterpret_cast<const f32x4 *>(&sm.dpv[an][4 * q]);
        d0 = fmaf(r[0], p[0], d0); d1 = fmaf(r[1], p[1], d1); d0 = fmaf(r[2], p[2], d0); d1 = fmaf(r[3], p[3], d1);
    };
    for (int grp = blockIdx.x; grp < ngrp; grp += gridDim.x) {
        const int b0 = grp * NTR;
        // Start of a group: the first records, the rows they may need, dL/dpooled and pooled are requested AT ONCE (one round trip to
        // memory in front of the group's first barrier, not three in a row; the addresses are valid whatever the records turn out to be)
        typedef const __attribute__((address_space(1))) float *gfl_p;
        f32x4 scr0 = sc_at(b0, 0);
        f32x4 rr0[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) rr0[q] = row_at(row_off(b0, 0), q);
        const int bn = b0 + an < B ? b0 + an : B - 1;
        float dpl[3], pol[3], dpe[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dpl[i] = ((gfl_p)a.dpooled)[(size_t)bn * H + 3 * as + i];
            pol[i] = ((gfl_p)a.pooled)[(size_t)bn * H + 3 * as + i];
            const int e = lane + 64 * i, n = e / H;
            dpe[i] = ((gfl_p)a.dpooled)[(size_t)(b0 + n < B ? b0 + n : B - 1) * H + (e - n * H)];
        }
        const bool open = (NSD_BX4_VAR & 2) ? false : __any(scr0[2] != 0.f) != 0;
        float sdot = 0.f, dsum = 0.f;
        auto close_record = [&](const int c, const f32x4 rec, const float dd) -> f32x4 {
            const int b = b0 + an, t = T - 1 - (16 * c + as);
            const bool ok = b < B && t >= 0 && t < T;
            const float ds = ok ? rec[0] * (dd - sdot) : 0.f;
            if (ok) { a.dscore_out[(size_t)b * T + t] = ds; dsum += ds; }
            return f32x4{rec[0], ds, 1.f, 0.f};
        };
        if (open) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { const int e = lane + 64 * i, n = e / H; sm.dpv[n][e - n * H] = dpe[i]; }
            float sd = fmaf(dpl[0], pol[0], fmaf(dpl[1], pol[1], dpl[2] * pol[2]));
            sd = oct_sum(sd);
            sdot = sd + dpp_quad<0x140>(sd);                         // dpooled . pooled: sum over the 16 lanes of the trial
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) dot_piece(rr0[q], q, d0, d1);
            scr0 = close_record(0, scr0, d0 + d1);
        }
        *reinterpret_cast<f32x4 *>(&sm.sc[0][0][0][0] + 4 * lane) = scr0;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(&sm.xs[0][0][0][0] + 4 * (lane + 64 * q)) = x_at(b0, 0, lane + 64 * q);
        if (a.rng.on) {
#pragma unroll 1
            for (int s = 0; s < XCH; ++s) rng_row(b0, s, 0);
        } else {
#pragma unroll
            for (int q = 0; q < MPL; ++q) *reinterpret_cast<f32x4 *>(&sm.mk[0][0][0][0] + 4 * (lane + 64 * q)) = mask_at(b0, 0, lane + 64 * q);
        }
        xstep_barrier(prof);
        for (int m0 = 0; m0 < n_steps; m0 += XCH) {
            const int c = m0 >> 4, cb = c & 1;
            asm volatile("" : "+v"(lane));
            f32x4 scr = sc_at(b0, c + 1);
            f32x4 xr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = x_at(b0, c + 1, lane + 64 * q);
            auto steps = [&](const int k0, const int k1) {
#pragma unroll 1
                for (int k = k0; k < k1; ++k) {
                    if (a.rng.on) rng_row(b0, m0 + XCH + k, cb ^ 1);
                    xstep_barrier(prof);
                }
            };
            // First part of the chunk: the top rows of the open records, requested six steps before they are used (in a train step they
            // come from HBM behind the kernel's own stream of saved activations: a request used one step later stood at the barrier)
            if (open) {
                f32x4 rr[12];
                const unsigned ro = row_off(b0, c + 1);
#pragma unroll
                for (int q = 0; q < 12; ++q) if (!(NSD_BX4_VAR & 4)) rr[q] = row_at(ro, q);
                steps(0, 6);
                float d0 = 0.f, d1 = 0.f;
#pragma unroll
                for (int q = 0; q < 12; ++q) dot_piece(rr[q], q, d0, d1);
                scr = close_record(c + 1, scr, d0 + d1);
                steps(6, 8);
            } else {
                steps(0, 8);
            }
            // Second half: the explicit multipliers in two batches of six requests (24 registers in flight, not 48).  Nobody reads buffer
            // cb ^ 1 during this chunk (every reader is at a step of chunk c), so each batch is written as soon as it has landed.
            constexpr int MH = MPL / 2;
            static_assert(MPL == 2 * MH, "two equal batches");
            asm volatile("" : "+v"(lane));
            if (!a.rng.on) {
                f32x4 mr[MH];
#pragma unroll
                for (int q = 0; q < MH; ++q) mr[q] = mask_at(b0, c + 1, lane + 64 * q);
                steps(8, 11);
#pragma unroll
                for (int q = 0; q < MH; ++q) *reinterpret_cast<f32x4 *>(&sm.mk[cb ^ 1][0][0][0] + 4 * (lane + 64 * q)) = mr[q];
#pragma unroll
                for (int q = 0; q < MH; ++q) mr[q] = mask_at(b0, c + 1, lane + 64 * (MH + q));
                steps(11, 14);
#pragma unroll
                for (int q = 0; q < MH; ++q) *reinterpret_cast<f32x4 *>(&sm.mk[cb ^ 1][0][0][0] + 4 * (lane + 64 * (MH + q))) = mr[q];
            } else {
                steps(8, 14);
            }
            *reinterpret_cast<f32x4 *>(&sm.sc[cb ^ 1][0][0][0] + 4 * lane) = scr;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(&sm.xs[cb ^ 1][0][0][0] + 4 * (lane + 64 * q)) = xr[q];
            steps(14, 16);
        }
        if (open) {                                                 // d attn.bias of the trial = sum_t dL/dscore_t
            float bs = oct_sum(dsum);
            bs += dpp_quad<0x140>(bs);
            if (as == 0 && b0 + an < B) a.hslabs[(size_t)(b0 + an) * a.Ph + a.o_attn_b] = bs;
        }
    }
    prof_store(a.dbg, prof);
}

__global__ __launch_bounds__(NTHR) void lstm2_bwd48x4_kernel(Lstm2BwdArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // macro steps 0 .. T+2 (the dW waves use the da of macro step m - 1), padded to whole 16-step chunks
    const int n_steps = ((a.T + 3 + XCH - 1) / XCH) * XCH;
    const int g = wave & 3, q = wave >> 2;                          // SIMD, slot
#ifdef NSD_BX4_ONLY_ROLE                                            // resource probe (never built into the library): one role alone
    if (NSD_BX4_ONLY_ROLE == 1) chain_role<1>(a, g, lane, n_steps);
    else if (NSD_BX4_ONLY_ROLE == 2) chain_role<0>(a, g, lane, n_steps);
    else if (NSD_BX4_ONLY_ROLE == 3) x1_role(a, g, lane, n_steps);
    else if (NSD_BX4_ONLY_ROLE == 4) dw_role<1>(a, g, lane, n_steps);
    else if (NSD_BX4_ONLY_ROLE == 5) dw_role<0>(a, g & 1, lane, n_steps);
    else if (NSD_BX4_ONLY_ROLE == 6) rows_role(a, lane, n_steps);
    else aux_role(a, lane, n_steps);
    return;
#endif
#ifndef NSD_BX4_PRIO
#define NSD_BX4_PRIO 0
#endif
#ifndef NSD_BX4_MAP
#define NSD_BX4_MAP 0
#endif
    constexpr int PC = NSD_BX4_PRIO == 0 ? 3 : NSD_BX4_PRIO == 1 ? 0 : 1, PX = NSD_BX4_PRIO == 0 ? 2 : NSD_BX4_PRIO == 1 ? 0 : 1, PD = NSD_BX4_PRIO == 2 ? 3 : 0;
#if NSD_BX4_MAP == 0
    // SIMDs 0..2: the two recurrences + two of {five dW waves, rows wave}; SIMD 3: the three X1 waves + aux (matrix pipe per step and
    // SIMD: 96 x 8 + ~2 x 144 cycles / 144 x 8)
    if (g < 3 && q == 0)      { __builtin_amdgcn_s_setprio(PC); chain_role<1>(a, g, lane, n_steps); }
    else if (g < 3 && q == 1) { __builtin_amdgcn_s_setprio(PC); chain_role<0>(a, g, lane, n_steps); }
    else if (g < 3 && q == 2) { __builtin_amdgcn_s_setprio(PD); dw_role<1>(a, g, lane, n_steps); }
    else if (g < 2)           { __builtin_amdgcn_s_setprio(PD); dw_role<0>(a, g, lane, n_steps); }
    else if (g == 2)          { __builtin_amdgcn_s_setprio(PD); rows_role(a, lane, n_steps); }
    else if (q < 3)           { __builtin_amdgcn_s_setprio(PX); x1_role(a, q, lane, n_steps); }
    else                      aux_role(a, lane, n_steps);
#else
    // the round-4 placement: X1 beside the recurrences of its SIMD, the dW / rows waves on SIMD 3
    if (g < 3 && q == 0)      { __builtin_amdgcn_s_setprio(PC); chain_role<1>(a, g, lane, n_steps); }
    else if (g < 3 && q == 1) { __builtin_amdgcn_s_setprio(PC); chain_role<0>(a, g, lane, n_steps); }
    else if (g < 3 && q == 2) { __builtin_amdgcn_s_setprio(PX); x1_role(a, g, lane, n_steps); }
    else if (g < 3)           { __builtin_amdgcn_s_setprio(PD); dw_role<1>(a, g, lane, n_steps); }
    else if (q < 2)           { __builtin_amdgcn_s_setprio(PD); dw_role<0>(a, q, lane, n_steps); }
    else if (q == 2)          { __builtin_amdgcn_s_setprio(PD); rows_role(a, lane, n_steps); }
    else                      aux_role(a, lane, n_steps);
#endif
    // a workgroup without a trial group (grid = the workspace's slab count) has written a zero slab: every role's sums are zero
}

}  // namespace

bool nsd_lstm2_bwd48x4_ok(const Lstm2BwdArgs &a) {
    return !a.residual && a.C <= 8 && (long)a.B * a.T * H * 16 < 0x7fffffffL && a.dsc_pack != nullptr;
}

int nsd_lstm2_bwd48x4_launch(const Lstm2BwdArgs &a, int grid, hipStream_t st) {
    if (!nsd_lstm2_bwd48x4_ok(a)) { nsd_set_error("lstm2_bwd48x4: launch outside the kernel's domain"); return NSD_E_INVALID; }
    hipLaunchKernelGGL(lstm2_bwd48x4_kernel, dim3(grid), dim3(NTHR), 0, st, a);
    NSD_CHECK_LAUNCH("lstm2_bwd48x4");
    return NSD_OK;
}
