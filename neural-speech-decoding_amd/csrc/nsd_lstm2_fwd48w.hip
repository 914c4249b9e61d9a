// nsd_lstm2_fwd48w.hip -- EXPERIMENTAL forward of the two-layer H=48 LSTM (training launches without the fused head): ONE WAVE PER LAYER.
//
// Replaces self.lstm(x) (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34) like nsd_lstm2_fwd48.hip, with the recurrence laid out
// the way tools/micro/wave_cell.hip probed it at the end of round 4: lane = unit (48 of 64 lanes), ALL four gates x 48 inputs in the
// lane (96 v_pk_fma_f32 per step, 192 weight registers of a 512-register wave), no cross-lane reduction, the cell in the lane, h to the
// wave's own LDS ring and back as broadcast reads -- program order instead of a workgroup barrier.  Four waves, one per SIMD:
//   wave 0  "X"   layer-0 input projection W_ih0 x_t + biases (exp2 arguments) and the dropout multipliers of step t -> LDS rings
//   wave 1  "L0"  layer 0: W_hh0 h0[t-1] + X[t], cell, h0[t] and the masked h0[t] -> rings; saves gates, h, c, in1
//   wave 2  "P"   layer-1 input projection W_ih1 in1[t] + biases -> ring
//   wave 3  "L1"  layer 1: W_hh1 h1[t-1] + P[t], cell; saves gates, h, c (+ top)
// The waves hand off through PROGRESS COUNTERS in LDS (steps completed per role; a producer also waits for its consumer before it
// overwrites a ring entry), not through s_barrier: with a workgroup barrier per step the probe's recurrence took 977 instead of 733
// cycles.  Diagnostic twin only (nsd_diag_force_fwd48(8)): the train step's fused head has not been ported to this shape.
#include "nsd_args.h"

namespace {

#ifndef NSD_F48W_SLEEP
#define NSD_F48W_SLEEP 0            // s_sleep between two looks at a progress counter (0: spin)
#endif
constexpr int H = 48;
constexpr int RG = 8;                // ring depth = unroll of the time loops: every ring slot is an immediate offset
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__host__ __device__ constexpr float gate_scale(const int g) { return g == 2 ? -2.f * LOG2E_F : -LOG2E_F; }
constexpr float KC = -2.f * LOG2E_F;

// (64 columns everywhere: lanes 48..63 compute a copy of unit 47 and write it to columns nobody reads -- no EXEC games in the loops)
struct WSmem {
    float xp[RG][64][4];             // layer-0 input projection + biases, exp2 arguments: [t % RG][unit][gate]
    float pb[RG][64][4];             // layer-1 input projection + biases
    float mk[RG][64];                // layer-0 dropout multipliers
    float h0[RG][64], in1[RG][64], h1[RG][64];
    int cnt[4];                      // steps completed: X, L0, P, L1
};
// (file-scope LDS object; the progress counters are read and written through address_space(3) volatile pointers: as generic volatile
// pointers hipcc emits flat accesses behind an aperture test and trips over it -- "Illegal instruction detected: Operand has
// incorrect register class", V_CMP_NE_U32_e32 0, $src_shared_base)
__shared__ __align__(16) WSmem g_wsm;
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int peek(const int who) { return *(const volatile lds_int *)(&g_wsm.cnt[who]); }
// wait until role `who` has completed step t (cnt > t).  `seen` caches the last value read: while the producer is two or more steps
// ahead no LDS read stands in front of the step
__device__ __forceinline__ void wait_step(const int who, const int t, int &seen) {
    if (seen <= t) {
        while ((seen = peek(who)) <= t) { if (NSD_F48W_SLEEP) __builtin_amdgcn_s_sleep(NSD_F48W_SLEEP); }
    }
    asm volatile("" ::: "memory");
}
// a producer must not overwrite ring entries its consumer has not read: checked once per RG / 2 steps for the steps that follow
__device__ __forceinline__ void wait_room(const int cons, const int t) {
    if ((t & (RG / 2 - 1)) == 0) {
        while (peek(cons) + RG / 2 <= t) { if (NSD_F48W_SLEEP) __builtin_amdgcn_s_sleep(NSD_F48W_SLEEP); }
    }
    asm volatile("" ::: "memory");
}
// (no wait in front of the counter: the LDS takes a wave's requests in order, so whoever reads the new count reads behind the step's
// data writes; every lane writes the same word -- no EXEC switch: 141 -> 134 us)
__device__ __forceinline__ void publish(const int me, const int t, const int lane) {
    asm volatile("" ::: "memory");
    *(volatile lds_int *)(&g_wsm.cnt[me]) = t + 1;
    (void)lane;
}

// all four gates of a unit against a 48-vector in LDS (broadcast reads: every lane reads the same 16 bytes).  The reads run one batch
// of four AHEAD of the FMAs that use them (8 in flight, 32 registers): with four in flight and the next four requested behind the
// batch's FMAs three LDS latencies stood in every step (130 -> 124 us); all twelve at once do not fit beside 192 weight registers.
// (Requesting a step's first batch right behind the previous step's write of h, ahead of its stores: no gain, 127 us.)
__device__ __forceinline__ void gates_dot(const f32x2 (&w)[4][24], const float *v, f32x2 (&acc)[4]) {
    f32x4 hv[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[0][q] = *reinterpret_cast<const f32x4 *>(v + 4 * q);
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) {
        if (qb < 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[qb + 1][q] = *reinterpret_cast<const f32x4 *>(v + 4 * (4 * (qb + 1) + q));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 lo = {hv[qb][q][0], hv[qb][q][1]}, hi = {hv[qb][q][2], hv[qb][q][3]};
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc[g] = pk_fma(w[g][2 * (4 * qb + q)], lo, acc[g]); acc[g] = pk_fma(w[g][2 * (4 * qb + q) + 1], hi, acc[g]); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void load_w(const float *w, const int u, f32x2 (&wv)[4][24]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            wv[g][q].x = gate_scale(g) * w[(size_t)(g * H + u) * H + 2 * q];
            wv[g][q].y = gate_scale(g) * w[(size_t)(g * H + u) * H + 2 * q + 1];
        }
}
// one field of the argument block as a wave-uniform value (the block arrives behind a pointer in VGPRs; a role copies the few fields it
// uses, not all ~170 dwords)
template <class P> __device__ __forceinline__ P *uni(P *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<P *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ rsrc_t rsrc_of(const float *base, const long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
struct Cell { float i, f, g, o, c, h; };
// arg[g]: exp2 arguments of the four gates (pre-activation x -log2e, tanh row x -2 log2e); c: cell state (updated)
__device__ __forceinline__ Cell cell(const float (&arg)[4], float &c) {
    Cell r;
    r.i = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg[0]));
    r.f = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg[1]));
    r.g = fmaf(2.f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg[2])), -1.f);
    r.o = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg[3]));
    c = fmaf(r.f, c, r.i * r.g);
    r.c = c;
    r.h = r.o * fmaf(2.f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(KC * c)), -1.f);
    return r;
}
// every role walks the trials of its workgroup with the same two barriers per trial; time loops run to a multiple of RG (surplus steps
// compute on ring contents nobody uses and save nothing)
#define TRIAL_LOOP_BEGIN                                                                                                    \
    for (int b = blockIdx.x; b < B; b += gridDim.x) {                                                                       \
        if (threadIdx.x < 4) g_wsm.cnt[threadIdx.x] = 0;                                                                    \
        if (threadIdx.x < 64) { g_wsm.h0[RG - 1][threadIdx.x] = 0.f; g_wsm.h1[RG - 1][threadIdx.x] = 0.f; }                \
        __syncthreads();                                                                                                    \
        const size_t bt = (size_t)b * T;
#define TRIAL_LOOP_END                                                                                                      \
        __syncthreads();                                                                                                    \
    }
constexpr unsigned VOFF_DROP = 0x80000000u;

__device__ __attribute__((noinline)) void role_x(const Lstm2FwdArgs &a_in, const int lane) {
    const float *x = uni(a_in.x), *w_ih0 = uni(a_in.w_ih0), *b_ih0 = uni(a_in.b_ih0), *b_hh0 = uni(a_in.b_hh0), *mask = uni(a_in.mask);
    const RngArgs rng = uniform_copy(a_in.rng);
    const int T = uni(a_in.T), B = uni(a_in.B), C = uni(a_in.C);
    const int u = lane < H ? lane : H - 1;
    float wx[4][8], bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bias[g] = gate_scale(g) * (b_ih0[g * H + u] + b_hh0[g * H + u]);
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) wx[g][ch] = ch < C ? gate_scale(g) * w_ih0[(size_t)(g * H + u) * C + ch] : 0.f;
    }
    TRIAL_LOOP_BEGIN
        // (x and the explicit multipliers are requested RG steps ahead: a load used in the step that issued it put the memory latency on every step)
        float xq[RG][8], mq[RG];
#pragma unroll
        for (int k = 0; k < RG; ++k) {
            const int tc = k < T ? k : T - 1;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) xq[k][ch] = ch < C ? x[(bt + tc) * C + ch] : 0.f;
            mq[k] = (mask && !rng.on) ? mask[(bt + tc) * H + u] : 1.f;
        }
        for (int t0 = 0; t0 < T; t0 += RG) {
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                const int t = t0 + k;
                float mkv = mq[k];
                if (rng.on) mkv = nsd_rand_u32(rng.seed, rng.base, (bt + (t < T ? t : T - 1)) * H + u) < rng.thr_lstm ? 0.f : rng.keep_lstm;
                f32x4 acc = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
                for (int ch = 0; ch < 8; ++ch)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = fmaf(wx[g][ch], xq[k][ch], acc[g]);
                const int tn = t + RG < T ? t + RG : T - 1;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) xq[k][ch] = ch < C ? x[(bt + tn) * C + ch] : 0.f;
                mq[k] = (mask && !rng.on) ? mask[(bt + tn) * H + u] : 1.f;
                wait_room(1, t);
                *reinterpret_cast<f32x4 *>(&g_wsm.xp[k][lane][0]) = acc;
                g_wsm.mk[k][lane] = mkv;
                publish(0, t, lane);
            }
        }
    TRIAL_LOOP_END
}

// LAYER 0: X ring + own h0 ring -> h0, in1 rings; LAYER 1: P ring + own h1 ring -> h1 ring.  Saves: gates (16 bytes), h, c (+ in1 / top)
// as buffer stores with the time offset in a scalar register.
template <int LAYER>
__device__ __attribute__((noinline)) void role_cell(const Lstm2FwdArgs &a_in, const int lane) {
    const float *w_hh = uni(LAYER == 0 ? a_in.w_hh0 : a_in.w_hh1);
    float *gact = uni(LAYER == 0 ? a_in.gact0 : a_in.gact1), *hseq = uni(LAYER == 0 ? a_in.hseq0 : a_in.hseq1);
    float *cseq = uni(LAYER == 0 ? a_in.cseq0 : a_in.cseq1), *aux = uni(LAYER == 0 ? a_in.inseq : a_in.top);
    const int T = uni(a_in.T), B = uni(a_in.B);
    const int u = lane < H ? lane : H - 1;
    const unsigned vo4 = lane < H ? 4u * lane : VOFF_DROP, vo16 = lane < H ? 16u * lane : VOFF_DROP;
    f32x2 wv[4][24];
    load_w(w_hh, u, wv);
    TRIAL_LOOP_BEGIN
        const long row4 = (long)T * H * 4;
        const rsrc_t r_g = rsrc_of(gact + bt * H * 4, row4 * 4), r_h = rsrc_of(hseq + bt * H, row4), r_c = rsrc_of(cseq + bt * H, row4);
        const rsrc_t r_a = rsrc_of(aux ? aux + bt * H : hseq, aux ? row4 : 0);
        float c = 0.f;
        int seen = 0;
        for (int t0 = 0; t0 < T; t0 += RG) {
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                const int t = t0 + k;
                wait_step(LAYER == 0 ? 0 : 2, t, seen);
                const f32x4 pa = *reinterpret_cast<const f32x4 *>(LAYER == 0 ? &g_wsm.xp[k][lane][0] : &g_wsm.pb[k][lane][0]);
                // (the projection is added BEHIND the products: as the accumulators' seed its LDS latency stood in front of the first FMA: 134 -> 130 us)
                f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
                gates_dot(wv, LAYER == 0 ? &g_wsm.h0[(k + RG - 1) & (RG - 1)][0] : &g_wsm.h1[(k + RG - 1) & (RG - 1)][0], acc);
                const float arg[4] = {(acc[0].x + pa[0]) + acc[0].y, (acc[1].x + pa[1]) + acc[1].y, (acc[2].x + pa[2]) + acc[2].y, (acc[3].x + pa[3]) + acc[3].y};
                const Cell r = cell(arg, c);
                float hm = r.h;
                if (LAYER == 0) {
                    hm = r.h * g_wsm.mk[k][lane];
                    wait_room(2, t);
                    g_wsm.h0[k][lane] = r.h;
                    g_wsm.in1[k][lane] = hm;
                } else {
                    g_wsm.h1[k][lane] = r.h;
                }
                publish(LAYER == 0 ? 1 : 3, t, lane);
                if (t < T) {                                            // saved activations of the step (behind the hand-off)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{r.i, r.f, r.g, r.o}), r_g, (int)vo16, t * (H * 16), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r.h), r_h, (int)vo4, t * (H * 4), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r.c), r_c, (int)vo4, t * (H * 4), 0);
                    if (aux) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hm), r_a, (int)vo4, t * (H * 4), 0);
                }
            }
        }
    TRIAL_LOOP_END
}

__device__ __attribute__((noinline)) void role_p(const Lstm2FwdArgs &a_in, const int lane) {
    const float *w_ih1 = uni(a_in.w_ih1), *b_ih1 = uni(a_in.b_ih1), *b_hh1 = uni(a_in.b_hh1);
    const int T = uni(a_in.T), B = uni(a_in.B);
    const int u = lane < H ? lane : H - 1;
    f32x2 wv[4][24];
    load_w(w_ih1, u, wv);
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = gate_scale(g) * (b_ih1[g * H + u] + b_hh1[g * H + u]);
    TRIAL_LOOP_BEGIN
        (void)bt;
        int seen = 0;
        for (int t0 = 0; t0 < T; t0 += RG) {
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                const int t = t0 + k;
                wait_step(1, t, seen);
                f32x2 acc[4] = {{bias[0], 0.f}, {bias[1], 0.f}, {bias[2], 0.f}, {bias[3], 0.f}};
                gates_dot(wv, &g_wsm.in1[k][0], acc);
                wait_room(3, t);
                *reinterpret_cast<f32x4 *>(&g_wsm.pb[k][lane][0]) = f32x4{acc[0].x + acc[0].y, acc[1].x + acc[1].y, acc[2].x + acc[2].y, acc[3].x + acc[3].y};
                publish(2, t, lane);
            }
        }
    TRIAL_LOOP_END
}

__global__ __launch_bounds__(256) void lstm2_fwd48w_kernel(Lstm2FwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0)      role_x(a, lane);
    else if (wave == 1) role_cell<0>(a, lane);
    else if (wave == 2) role_p(a, lane);
    else                role_cell<1>(a, lane);
}

}  // namespace

bool nsd_lstm2_fwd48w_ok(const Lstm2FwdArgs &a) {
    return !a.residual && !a.logits_out && !a.head_train && a.C <= 8 && a.gact0 && a.hseq0 && a.cseq0 && a.inseq && a.gact1 && a.hseq1 && a.cseq1;
}

int nsd_lstm2_fwd48w_launch(const Lstm2FwdArgs &a, int grid, hipStream_t st) {
    if (!nsd_lstm2_fwd48w_ok(a)) { nsd_set_error("lstm2_fwd48w: launch outside the kernel's domain"); return NSD_E_INVALID; }
    hipLaunchKernelGGL(lstm2_fwd48w_kernel, dim3(grid), dim3(256), 0, st, a);
    NSD_CHECK_LAUNCH("lstm2_fwd48w");
    return NSD_OK;
}
