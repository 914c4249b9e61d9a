// nsd_prof.h -- in-kernel cycle stamps for the DIAGNOSTIC build only (-DNSD_PROFILE=1 -> libnsd_hip_prof.so).
// The shipped library is compiled with NSD_PROFILE=0: every function below folds to nothing.
#pragma once
#include <hip/hip_runtime.h>

#ifndef NSD_PROFILE
#define NSD_PROFILE 0
#endif
constexpr bool kProfile = NSD_PROFILE != 0;

// timing-experiment switches (env NSD_ABLATE, tools/kbench.py --ablate): compiled into the diagnostic build only -- in the
// shipped library every test folds to false
#ifndef NSD_ABLATE_HOOKS
#define NSD_ABLATE_HOOKS 0          // 1: keep the switches without the cycle stamps (tools/sweep.sh ... -DNSD_ABLATE_HOOKS=1)
#endif
__device__ __forceinline__ bool ablated(const int mask, const int bit) { return (kProfile || NSD_ABLATE_HOOKS) && (mask & bit) != 0; }

struct Prof {
    long long work, wait, last;
    long long seg[6], mark;
    bool on;
};
__device__ __forceinline__ Prof prof_init(long long *dbg) {
    Prof p;
    p.work = 0; p.wait = 0; p.mark = 0;
    for (int i = 0; i < 6; ++i) p.seg[i] = 0;
    p.on = kProfile && (dbg != nullptr) && blockIdx.x == 0;
    p.last = p.on ? clock64() : 0;
    return p;
}
// SLEEP > 0: s_sleep after the barrier (units of 64 cycles) -- waves off the critical path let the chains' LDS reads
// enter the (first-come-first-served) LDS queue first
template <bool RAW, int SLEEP = 0>
__device__ __forceinline__ void step_barrier(Prof &p) {
    if (kProfile && p.on) {
        const long long t = clock64();
        p.work += t - p.last;
        if (RAW) __builtin_amdgcn_s_barrier(); else __syncthreads();
        const long long t2 = clock64();
        p.wait += t2 - t;
        p.last = t2;
    } else {
        if (RAW) __builtin_amdgcn_s_barrier(); else __syncthreads();
    }
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
}
// sub-phase stamp: IDX < 0 only (re)starts the clock; WAIT_LDS drains the LDS queue first so that the segment
// ends when the data has really arrived
template <int IDX, bool WAIT_LDS>
__device__ __forceinline__ void prof_mark(Prof &p) {
    if (kProfile && p.on) {
        __builtin_amdgcn_sched_barrier(0);
        if (WAIT_LDS) __builtin_amdgcn_s_waitcnt(0xC07F);
        const long long t = clock64();
        if (IDX >= 0) p.seg[IDX < 0 ? 0 : IDX] += t - p.mark;
        p.mark = t;
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void prof_store(long long *dbg, const Prof &p) {
    if (kProfile && p.on && (threadIdx.x & 63) == 0) {
        long long *o = dbg + 8 * (threadIdx.x >> 6);
        o[0] = p.work; o[1] = p.wait;
        for (int i = 0; i < 6; ++i) o[2 + i] = p.seg[i];
        dbg[256 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID: wave slot, SIMD, CU ...
    }
}
