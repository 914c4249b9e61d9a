// micro-benchmark: what does a wave that streams MFMAs cost a VALU wave on the SAME SIMD (gfx950)?
// One workgroup of 8 waves per CU (waves w and w+4 share a SIMD: the dispatcher deals waves round-robin).
// Wave 0 runs a dependent VALU chain and clocks itself; wave 4 streams independent MFMAs of one kind until
// wave 0 is done; the other waves exit.  Reported: cycles per chain instruction of wave 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAIN, int MF>
__global__ __launch_bounds__(512) void k(int iters, long long *cyc, float *out, int *hw) {
    __shared__ volatile int done;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) hw[wave] = __builtin_amdgcn_s_getreg(63492);
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        float a = threadIdx.x * 1e-3f + 0.5f, b = 0.999f, c = 1e-4f;
        f32x2 p = {a, a + 1.f}, pb = {b, b}, pc = {c, c};
        const long long t0 = clock64();
        for (int i = 0; i < iters; ++i) {
            if (CHAIN == 0) {
#pragma unroll
                for (int q = 0; q < 64; ++q) a = fmaf(a, b, c);
            } else if (CHAIN == 1) {
#pragma unroll
                for (int q = 0; q < 64; ++q) p = __builtin_elementwise_fma(p, pb, pc);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) a = fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-a)), -1.f);
            }
        }
        const long long t1 = clock64();
        done = 1;
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
        out[blockIdx.x * 64 + threadIdx.x] = a + p.x + p.y;
    } else if (wave == 4 && MF != 0) {
        f32x4 acc[6];
        f32x16 big[2];
        for (int q = 0; q < 6; ++q) acc[q] = (f32x4){0, 0, 0, 0};
        for (int q = 0; q < 2; ++q) for (int e = 0; e < 16; ++e) big[q][e] = 0.f;
        const float av = threadIdx.x * 1e-3f, bv = 0.5f;
        bf16x8 ha, hb;
        for (int e = 0; e < 8; ++e) { ha[e] = (__bf16)0.5f; hb[e] = (__bf16)0.25f; }
        long long n = 0;
        while (!done) {
            if (MF == 1) {
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[q], 0, 0, 0);
            } else if (MF == 2) {
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[q], 0, 0, 0);
            } else if (MF == 3) {
#pragma unroll
                for (int q = 0; q < 2; ++q) big[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, big[q], 0, 0, 0);
            } else if (MF == 4) {
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[q], 0, 0, 0);
            } else if (MF == 5) {           // a VALU wave instead of an MFMA wave: 24 independent pk_fma
                f32x2 z[4] = {{av, bv}, {av, bv}, {av, bv}, {av, bv}};
#pragma unroll
                for (int q = 0; q < 24; ++q) z[q & 3] = __builtin_elementwise_fma(z[q & 3], (f32x2){0.999f, 0.999f}, (f32x2){1e-4f, 1e-4f});
                acc[0][0] += z[0].x + z[1].x + z[2].y + z[3].y;
            }
            ++n;
        }
        float s = 0.f;
        for (int q = 0; q < 6; ++q) s += acc[q][0] + acc[q][3];
        s += big[0][0] + big[1][5];
        out[16384 + blockIdx.x * 64 + (threadIdx.x & 63)] = s;
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[1] = n;
    }
}

template <int CHAIN, int MF>
void run(const char *cn, const char *mn, int ops, long long *cyc, float *out, int *hw) {
    const int iters = 2000;
    long long h[2] = {0, 0};
    int hh[8];
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cyc, 0, 16);
        hipLaunchKernelGGL((k<CHAIN, MF>), dim3(256), dim3(512), 0, 0, iters, cyc, out, hw);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hh, hw, 32, hipMemcpyDeviceToHost);
    printf("%-22s | %-26s : %6.2f cycles per chain op   (mfma-wave loop bodies: %lld; simd of wave0=%d wave4=%d)\n", cn, mn,
           (double)h[0] / ((double)iters * ops), h[1], (hh[0] >> 4) & 3, (hh[4] >> 4) & 3);
}

int main() {
    long long *cyc; float *out; int *hw;
    (void)hipMalloc(&cyc, 64); (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&hw, 64);
#define ROW(C, CN, OPS)                                                   \
    run<C, 0>(CN, "alone", OPS, cyc, out, hw);                            \
    run<C, 1>(CN, "mfma_f32_16x16x4_f32", OPS, cyc, out, hw);             \
    run<C, 2>(CN, "mfma_f32_4x4x1_f32", OPS, cyc, out, hw);               \
    run<C, 3>(CN, "mfma_f32_32x32x2_f32", OPS, cyc, out, hw);             \
    run<C, 4>(CN, "mfma_f32_16x16x32_bf16", OPS, cyc, out, hw);           \
    run<C, 5>(CN, "24 independent pk_fma", OPS, cyc, out, hw);
    ROW(0, "64 dependent v_fma", 64)
    ROW(1, "64 dependent v_pk_fma", 64)
    ROW(2, "16x(exp2,add,rcp,fma)", 64)
    return 0;
}
