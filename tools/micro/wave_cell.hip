// wave_cell.hip -- a design probe for the one-trial FORWARD kernel (H = 48): the whole recurrence of a layer in ONE wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/wave_cell tools/micro/wave_cell.hip && tools/micro/wave_cell
// Today (nsd_lstm2_fwd48.hip) a layer's step is three waves of (unit, k-slice) lanes: 24 v_pk_fma_f32 per lane, a quad reduce-scatter, the
// four gates of a unit in four lanes (DPP broadcasts), h through LDS BEHIND A WORKGROUP BARRIER -- 940 cycles per step even with one
// recurrence alone in the workgroup.  The probe: lane = unit (48 of 64 lanes), all four gates x 48 inputs in the lane (96 v_pk_fma_f32,
// 192 weight registers: a 512-register wave), no reduction, cell in the lane, h to the wave's own LDS row and back as broadcast reads
// (12 ds_read_b128 with one address for all lanes) -- program order instead of a barrier.  Variants: (a) the recurrence alone, (b) with a
// workgroup barrier per step (four such waves, one per SIMD), (c) FMAs only / cell only, for the split.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float sig2(float arg) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(arg)); }

// MODE bit 0: barrier per step, bit 1: no FMAs, bit 2: no cell
template <int MODE>
__global__ __launch_bounds__(256) void cell_kernel(const float *w, float *out, long long *cycles, int steps) {
    __shared__ __align__(16) float hs[4][2][64];                    // [wave][parity][unit]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x2 wv[4][24];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 24; ++q) { wv[g][q].x = w[((g * 48 + (lane % 48)) * 48 + 2 * q) % 4096] * 0.05f; wv[g][q].y = w[((g * 48 + (lane % 48)) * 48 + 2 * q + 1) % 4096] * 0.05f; }
    float cK = 0.f;
    hs[wave][0][lane] = 0.01f * lane; hs[wave][1][lane] = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int t = 0; t < steps; t += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const float *hp = &hs[wave][par][0];
            f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            if (!(MODE & 2)) {
#pragma unroll
                for (int q4 = 0; q4 < 12; ++q4) {
                    const f32x4 hv = *reinterpret_cast<const f32x4 *>(hp + 4 * q4);        // same address in every lane: a broadcast read
                    const f32x2 lo = {hv[0], hv[1]}, hi = {hv[2], hv[3]};
#pragma unroll
                    for (int g = 0; g < 4; ++g) { acc[g] = pk_fma(wv[g][2 * q4], lo, acc[g]); acc[g] = pk_fma(wv[g][2 * q4 + 1], hi, acc[g]); }
                }
            } else {
                const f32x4 hv = *reinterpret_cast<const f32x4 *>(hp);
                acc[0].x = hv[0]; acc[1].x = hv[1]; acc[2].x = hv[2]; acc[3].x = hv[3];
            }
            float h;
            if (!(MODE & 4)) {
                const float ig = sig2(acc[0].x + acc[0].y), fg = sig2(acc[1].x + acc[1].y), og = sig2(acc[3].x + acc[3].y);
                const float gg = fmaf(2.f, sig2(acc[2].x + acc[2].y), -1.f);
                cK = fmaf(fg, cK, ig * gg);
                h = og * fmaf(2.f, sig2(cK), -1.f);
            } else {
                h = (acc[0].x + acc[0].y) + (acc[1].x + acc[1].y) + (acc[2].x + acc[2].y) + (acc[3].x + acc[3].y);
            }
            hs[wave][par ^ 1][lane] = h;                             // this wave's next step reads it back: program order, no barrier
            if (MODE & 1) __syncthreads();
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = hs[wave][0][lane] + cK;
}

template <int MODE>
static void run(const char *name, const float *dw, float *dout, long long *dc, int steps) {
    long long hc[4];
    hipLaunchKernelGGL((cell_kernel<MODE>), dim3(256), dim3(256), 0, 0, dw, dout, dc, steps);
    hipLaunchKernelGGL((cell_kernel<MODE>), dim3(256), dim3(256), 0, 0, dw, dout, dc, steps);
    hipDeviceSynchronize();
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    printf("%-64s %7.1f cycles (clock64 ticks) per step, wave 0 of workgroup 0\n", name, (double)hc[0] / steps);
}

int main() {
    float *hw = (float *)malloc(4096 * 4), *dw, *dout;
    long long *dc;
    for (int i = 0; i < 4096; ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMalloc(&dw, 4096 * 4); hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&dc, 1024 * 8);
    hipMemcpy(dw, hw, 4096 * 4, hipMemcpyHostToDevice);
    const int steps = 2000;
    run<0>("four waves (one per SIMD), each a whole layer, no barrier", dw, dout, dc, steps);
    run<1>("... with a workgroup barrier per step", dw, dout, dc, steps);
    run<2>("... no FMAs (LDS round trip + cell)", dw, dout, dc, steps);
    run<4>("... no cell (LDS round trip + 96 v_pk_fma_f32 + 4 adds)", dw, dout, dc, steps);
    run<6>("... neither (the LDS round trip alone)", dw, dout, dc, steps);
    return 0;
}
