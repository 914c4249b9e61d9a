// nsd_common.h -- shared host/device helpers of libnsd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/nsd.h"

// ---------------------------------------------------------------------------------------------
// host side: error text + parameter layout
// ---------------------------------------------------------------------------------------------
void nsd_set_error(const char *fmt, ...);

struct ParamLayout {
    int64_t w_ih[NSD_MAX_LAYERS], w_hh[NSD_MAX_LAYERS], b_ih[NSD_MAX_LAYERS], b_hh[NSD_MAX_LAYERS];
    int64_t ln_w, ln_b, attn_w, attn_b, fc0_w, fc0_b, fc3_w, fc3_b, total;
    int64_t lstm_total;   // floats belonging to the LSTM stack (prefix of the vector)
};
ParamLayout nsd_make_layout(int C, int H, int L, int K, int F);
int nsd_check_dims(const nsd_dims *d);
int nsd_num_cus();

#define NSD_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            nsd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return NSD_E_LAUNCH;                                                      \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
#define LOG2E_F 1.44269504088896340736f

// quad (4 adjacent lanes) cross-lane moves via DPP quad_perm: no LDS, no latency beyond a VALU op.
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
__device__ __forceinline__ float quad_xor1(float v) { return dpp_quad<QP(1, 0, 3, 2)>(v); }
__device__ __forceinline__ float quad_xor2(float v) { return dpp_quad<QP(2, 3, 0, 1)>(v); }
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) { return dpp_quad<QP(Q, Q, Q, Q)>(v); }
__device__ __forceinline__ float quad_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    return v;
}

// rotation inside a row of 16 lanes (DPP row_ror:N): lane i receives the value of lane (i + N) % 16 or (i - N) % 16 --
// only used in sums over all rotations by a multiple of 4, where the direction does not matter
template <int N>
__device__ __forceinline__ float row_ror(float v) { return dpp_quad<0x120 + N>(v); }

// sigma(x) = 1/(1+2^(-x*log2e)); tanh(x) = 2*sigma(2x) - 1.  v_exp_f32 / v_rcp_f32 (1 ulp each).
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E_F * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * LOG2E_F * x)), -1.0f);
}
// unified gate activation: a*rcp(1+exp2(b*x))+c with per-lane constants (sigmoid: 1,-log2e,0; tanh: 2,-2log2e,-1)
__device__ __forceinline__ float gate_act(float x, float a, float b, float c) {
    return fmaf(a, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(b * x)), c);
}

// wave64 reductions (all lanes get the result): DPP inside a row of 16 lanes (quad xor 1/2, half-row mirror, row
// mirror -- each an all-reduce step because both partners end up with the same value), then the four row totals
// through v_readlane.  No LDS crossbar (ds_bpermute) on the way: ~10 short instructions instead of 6 LDS round trips.
__device__ __forceinline__ float rows_combine_sum(float v) {      // v uniform inside each row of 16
    const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a0 + a1) + (a2 + a3);
}
__device__ __forceinline__ float rows_combine_max(float v) {
    const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
}
__device__ __forceinline__ float oct_sum(float v) {               // all-reduce over aligned groups of 8 lanes
    v += quad_xor1(v);
    v += quad_xor2(v);
    v += dpp_quad<0x141>(v);                                      // row_half_mirror: lane i <-> 7 - i
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = oct_sum(v);
    v += dpp_quad<0x140>(v);                                      // row_mirror: lane i <-> 15 - i
    return rows_combine_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, quad_xor1(v));
    v = fmaxf(v, quad_xor2(v));
    v = fmaxf(v, dpp_quad<0x141>(v));
    v = fmaxf(v, dpp_quad<0x140>(v));
    return rows_combine_max(v);
}

// A called function sees the kernel's argument block through a pointer: what it loads from there sits in VGPRs, and hipcc cannot know that
// every lane holds the same value -- buffer descriptors built from such pointers are used inside WATERFALL loops (one pass per distinct
// value, a vmcnt(0) in front), loop bounds become vector compares.  uniform_copy() hands every dword through v_readfirstlane: scalar again.
template <class A>
__device__ __forceinline__ A uniform_copy(const A &in) {
    static_assert(sizeof(A) % 4 == 0, "argument block: whole dwords");
    A out;
    const unsigned *s = reinterpret_cast<const unsigned *>(&in);
    unsigned *d = reinterpret_cast<unsigned *>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(A) / 4; ++i) d[i] = __builtin_amdgcn_readfirstlane(s[i]);
    return out;
}

// The dense head's weights -> LDS once per workgroup (fc.0 [F][H] with row stride W0S, fc.3 [K][F] as it is), by ONE wave, sixteen requests
// of a lane in flight at a time.  (As `for (e = lane; e < n; e += 64) lds[..] = w[e]` hipcc emits one load, a full wait and one LDS
// write per trip: 26 round trips to the L2 in front of the workgroup's first step barrier, ~14 us of a launch.)
template <int HH, int W0S>
__device__ __forceinline__ void stage_head_weights(const float *fc0_w, const float *fc3_w, const int F, const int K, float *w0, float *w3, const int lane) {
    typedef const __attribute__((address_space(1))) float *gw_p;
    const int n0 = F * HH, n3 = K * F;
    for (int e0 = 0; e0 < n0; e0 += 64 * 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int e = e0 + lane + 64 * i; v[i] = ((gw_p)fc0_w)[e < n0 ? e : 0]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int e = e0 + lane + 64 * i; if (e < n0) { const int f = e / HH; w0[f * W0S + (e - f * HH)] = v[i]; } }
    }
    for (int e0 = 0; e0 < n3; e0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = e0 + lane + 64 * i; v[i] = ((gw_p)fc3_w)[e < n3 ? e : 0]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = e0 + lane + 64 * i; if (e < n3) w3[e] = v[i]; }
    }
}

// in-kernel train-mode streams (see nsd_rng in nsd.h): thresholds and keep factors precomputed on the host
struct RngArgs {
    uint64_t seed;
    uint32_t base;            // stream ids base, base+1, base+2
    uint32_t thr_lstm, thr_head;
    float keep_lstm, keep_head;
    int on;
};
static inline uint32_t nsd_drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

// counter-based random stream shared bit-for-bit with oracle/nsd_oracle.c (nsd_oracle_rand_u32)
__host__ __device__ __forceinline__ uint32_t nsd_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__host__ __device__ __forceinline__ uint32_t nsd_rand_u32(uint64_t seed, uint32_t stream, uint64_t index) {
    uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    uint32_t h = nsd_mix32(lo ^ s0);
    h = nsd_mix32(h + 0x9e3779b9U * (stream + 1u) + hi);
    h = nsd_mix32(h ^ s1);
    return h;
}
