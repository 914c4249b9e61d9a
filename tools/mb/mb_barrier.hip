// micro-benchmark: cost of one workgroup barrier per step, by workgroup size, with a tiny dependent
// LDS hand-off per step (write -> barrier -> read), 1 workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void k(int steps, float* out) {
    __shared__ float buf[2][1024];
    const int tid = threadIdx.x;
    float v = tid * 0.001f;
    buf[0][tid] = v; buf[1][tid] = v;
    __syncthreads();
    for (int m = 0; m < steps; ++m) {
        if (MODE >= 1) {   // dependent LDS hand-off
            float w = buf[(m + 1) & 1][(tid + 1) & (blockDim.x - 1)];
            v = v * 0.999f + w * 0.001f;
            buf[m & 1][tid] = v;
        }
        if (MODE >= 2) {   // + a 48-FMA dependent-ish chain
            float a0 = v, a1 = v + 1.f, a2 = v + 2.f, a3 = v + 3.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) { a0 = fmaf(a0, 0.5f, v); a1 = fmaf(a1, 0.5f, v); a2 = fmaf(a2, 0.5f, v); a3 = fmaf(a3, 0.5f, v); }
            v = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + tid] = v;
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int steps = 20000;
    int sizes[] = {64, 128, 256, 384, 512, 768, 1024};
    for (int mode = 0; mode < 3; ++mode)
        for (int s : sizes) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(s), 0, 0, steps, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(s), 0, 0, steps, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(s), 0, 0, steps, out);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d  threads %4d : %.1f ns/step\n", mode, s, ms * 1e6 / steps);
        }
    return 0;
}
