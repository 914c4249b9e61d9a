// micro-benchmark: what paces "MFMA + one ds_read_b128 per gap" in the scan kernels?   hipcc -O3 --offload-arch=gfx950 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW>   // MODE 0: mfma only; 1: + pipelined ds_read_b128 (padded rows); 2: + reads, weights partly in AGPR (192 regs); 3: reads only
__global__ __launch_bounds__(256) void k(const bf16x8 *wsrc, float *out, long long *cyc, int iters) {
    constexpr int LD = 264;
    __shared__ __align__(16) __bf16 tile[3][32 * LD];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hh = lane >> 5;
    for (int i = tid; i < 3 * 32 * LD; i += 256) (&tile[0][0])[i] = (__bf16)(0.001f * (i % 97));
    bf16x8 w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = wsrc[(i * 256 + tid) % 4096];
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    long long t0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int N = 48, D = 12;
        bf16x8 f[D];
        auto rd = [&](int i) { return *reinterpret_cast<const bf16x8 *>(&tile[i / 16][col * LD + 16 * (i % 16) + 8 * hh]); };
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < D; ++i) f[i] = rd(i);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) f[i] = w[i % NW];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (MODE != 3) {
                if (i < 16) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % NW], f[i % D], acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % NW], f[i % D], acc1, 0, 0, 0);
            } else {
                acc0[i % 16] += (float)f[i % D][0];
            }
            if (MODE >= 1 && i + D < N) f[i % D] = rd(i + D);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// MODE 4: the fused backward's MFMA phase: weights + accumulators in AGPRs (asm), 3 accumulators, 2 x 24 MFMAs, 16 fragment reads each
template <int NW>
__global__ __launch_bounds__(256) void k4(const bf16x8 *wsrc, float *out, long long *cyc, int iters) {
    __shared__ __align__(16) __bf16 dab[2][8][512];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 8 * 512; i += 256) (&dab[0][0][0])[i] = (__bf16)(0.001f * (i % 97));
    bf16x8 w[2][3][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 8; ++i) w[r][p][i] = wsrc[((r * 3 + p) * 8 + i) * 64 % 4096 + lane];
    __syncthreads();
    float s = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            f32x16 a0, a1, a2;
            constexpr int D = 3;
            bf16x8 g1[D], g0[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { g1[i] = *reinterpret_cast<const bf16x8 *>(&dab[1][i][lane * 8]); g0[i] = *reinterpret_cast<const bf16x8 *>(&dab[0][i][lane * 8]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i == 0) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(a0) : "a"(w[ri][0][i]), "v"(g1[i % D]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(a1) : "a"(w[ri][1][i]), "v"(g1[i % D]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(a2) : "a"(w[ri][2][i]), "v"(g0[i % D]));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a0) : "a"(w[ri][0][i]), "v"(g1[i % D]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a1) : "a"(w[ri][1][i]), "v"(g1[i % D]));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(a2) : "a"(w[ri][2][i]), "v"(g0[i % D]));
                }
                if (i + D < 8) { g1[i % D] = *reinterpret_cast<const bf16x8 *>(&dab[1][i + D][lane * 8]); g0[i % D] = *reinterpret_cast<const bf16x8 *>(&dab[0][i + D][lane * 8]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1), "+a"(a2));
            s += a0[0] + a1[3] + a2[7];
        }
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NW>
void run(const char *name, const bf16x8 *w, float *out, long long *cyc, int grid) {
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE, NW>), dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, NW>), dim3(grid), dim3(256), 0, 0, w, out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    printf("%-46s grid %3d: %8.1f cycles / 48-MFMA step (wg 0), %7.1f per MFMA; wall %.3f ms -> %.1f ns/step\n", name, grid, (double)h[0] / iters, (double)h[0] / iters / 48, ms, ms * 1e6 / iters);
}

int main() {
    bf16x8 *w; float *out; long long *cyc;
    hipMalloc(&w, 4096 * 16); hipMemset(w, 0x3c, 4096 * 16);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    for (int grid : {1, 256}) {
        run<0, 16>("mfma only, 16 weight frags (VGPR)", w, out, cyc, grid);
        run<0, 48>("mfma only, 48 weight frags (192 regs)", w, out, cyc, grid);
        run<1, 16>("mfma + ds_read_b128/gap, 16 weight frags", w, out, cyc, grid);
        run<1, 48>("mfma + ds_read_b128/gap, 48 weight frags", w, out, cyc, grid);
        run<3, 16>("ds_read_b128 only (48 per step)", w, out, cyc, grid);
        {
            hipLaunchKernelGGL((k4<0>), dim3(grid), dim3(256), 0, 0, w, out, cyc, 200);
            hipDeviceSynchronize();
            std::vector<long long> h(grid);
            hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            printf("%-46s grid %3d: %8.1f cycles / 48-MFMA step (wg 0), %7.1f per MFMA\n", "asm mfma, AGPR weights, 3 accs, 2x24 + settle", grid, (double)h[0] / 200, (double)h[0] / 200 / 48);
        }
    }
    return 0;
}
