// micro-benchmark: latency of DEPENDENT VALU chains on gfx950, by waves per SIMD (1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(int iters, float* out) {
    float a = threadIdx.x * 1e-3f + 0.5f, b = 0.999f, c = 1e-4f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {           // 64 dependent v_fma_f32
#pragma unroll
            for (int q = 0; q < 64; ++q) a = fmaf(a, b, c);
        } else if (MODE == 1) {    // 16 x (exp2 -> add -> rcp -> fma): the sigma/tanh chain
#pragma unroll
            for (int q = 0; q < 16; ++q) a = fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-a)), -1.f);
        } else if (MODE == 2) {    // 32 dependent DPP quad adds
#pragma unroll
            for (int q = 0; q < 32; ++q)
                a = a * 0.5f + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0xB1, 0xF, 0xF, true));
        } else {                   // 64 independent-ish FMAs (4 chains)
            float a1 = a + 1, a2 = a + 2, a3 = a + 3;
#pragma unroll
            for (int q = 0; q < 16; ++q) { a = fmaf(a, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c); }
            a = (a + a1) + (a2 + a3);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[] = {"64 dependent fma", "16x(exp2,add,rcp,fma)", "32x(mul + dpp add)", "64 fma in 4 chains"};
    const int ops[] = {64, 64, 64, 64};
    int sizes[] = {64, 256, 512, 768, 1024};
    for (int mode = 0; mode < 4; ++mode)
        for (int s : sizes) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(s), 0, 0, iters, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(s), 0, 0, iters, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(s), 0, 0, iters, out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(s), 0, 0, iters, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-24s threads %4d : %7.2f ns per loop body  (%.2f ns/op at ~2.4GHz = %.1f cyc)\n", names[mode], s, ms * 1e6 / iters,
                   ms * 1e6 / iters / ops[mode], ms * 1e6 / iters / ops[mode] * 2.4);
        }
    return 0;
}
