// micro-benchmark: does v_mfma_f32_32x32x16_bf16 issue slower with its A operand in AGPRs?   hipcc -O3 --offload-arch=gfx950 mfma_agpr.hip -o mfma_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: A in VGPR (asm "v"), acc "a";  1: A in AGPR ("a");  2: A in AGPR, B in AGPR;  3: A "v", 3 accumulators round robin; 4: A "a", 3 accumulators
__global__ __launch_bounds__(256) void k(const bf16x8 *wsrc, float *out, long long *cyc, int iters) {
    const int tid = threadIdx.x;
    bf16x8 w[16], f[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = wsrc[(i * 256 + tid) % 4096];
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = wsrc[(i * 256 + tid + 77) % 4096];
    f32x16 a0 = {0}, a1 = {0}, a2 = {0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a0) : "v"(w[i % 16]), "v"(f[i % 4]));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a0) : "a"(w[i % 16]), "v"(f[i % 4]));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a0) : "a"(w[i % 16]), "a"(f[i % 4]));
            if (MODE == 3) { f32x16 &a = i % 3 == 0 ? a0 : (i % 3 == 1 ? a1 : a2); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a) : "v"(w[i % 16]), "v"(f[i % 4])); }
            if (MODE == 4) { f32x16 &a = i % 3 == 0 ? a0 : (i % 3 == 1 ? a1 : a2); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a) : "a"(w[i % 16]), "v"(f[i % 4])); }
            if (MODE == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(w[i % 16]), "v"(f[i % 4]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1), "+a"(a2));
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, const bf16x8 *w, float *out, long long *cyc) {
    const int iters = 200, grid = 256;
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    printf("%-60s %7.1f cycles per MFMA\n", name, (double)h[0] / iters / 48);
}
int main() {
    bf16x8 *w; float *out; long long *cyc;
    (void)hipMalloc(&w, 4096 * 16); (void)hipMemset(w, 0x3c, 4096 * 16);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    run<0>("A in VGPR, B in VGPR, accumulator in AGPR, one chain", w, out, cyc);
    run<1>("A in AGPR, B in VGPR, accumulator in AGPR, one chain", w, out, cyc);
    run<2>("A in AGPR, B in AGPR, accumulator in AGPR, one chain", w, out, cyc);
    run<3>("A in VGPR, 3 accumulators round robin", w, out, cyc);
    run<4>("A in AGPR, 3 accumulators round robin", w, out, cyc);
    run<5>("A, B, accumulator all in VGPRs, one chain", w, out, cyc);
    return 0;
}
