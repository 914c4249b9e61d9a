// mfma4x4.hip -- v_mfma_f32_4x4x1_16B_f32 as the gate product of 4 trials: operand layout, broadcast modifiers, issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma4x4 tools/micro/mfma4x4.hip && tools/micro/mfma4x4
// 16 blocks per instruction, block b = lanes 4b..4b+3:  D_b[4x4] += A_b[4x1] * B_b[1x4].
//   (1) layout probe: A lane (b, i) = row i of block b, B lane (b, j) = column j, D register i of lane (b, j) = D_b[i][j]
//   (2) broadcast probes: BLGP 4..7 = one 16-lane row of B for all four rows; CBSZ = 4 / ABID = n = block n's A for all 16 blocks
//   (3) issue rate of one wave per SIMD: one dependent accumulator chain, 2 and 4 independent chains; with a second wave on the
//       same SIMD streaming transcendental VALU work (does the cell arithmetic of another wave hide behind the MFMAs?); with 2..4 waves
//       per SIMD streaming MFMAs (the pipe's own rate: one 4x4x1 per 8 cycles whatever the number of waves -- the nominal fp32 rate)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID, int BLGP>
__global__ void probe_kernel(const float *a, const float *b, float *d) {
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, CBSZ, ABID, BLGP);
    for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}

// CHAINS independent accumulator chains, N MFMAs in all per iteration, weights (A) distinct registers
template <int CHAINS>
__global__ __launch_bounds__(512) void rate_kernel(const float *w, float *out, long long *cycles, int iters, int valu_partner) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4 && valu_partner != 3) {  // waves 4..7 share SIMDs 0..3 with waves 0..3 (valu_partner == 3: they stream MFMAs too)
        if (!valu_partner) return;
        float x = 0.001f * lane, y = 0.f;
        const long long t0 = clock64();
        for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float e = __builtin_amdgcn_exp2f(x + u);
                y += __builtin_amdgcn_rcpf(1.0f + e);
                x = fmaf(x, 0.999f, 0.001f);
            }
        }
        const long long t1 = clock64();
        if (lane == 0) cycles[wave] = t1 - t0;
        out[512 + threadIdx.x] = y;
        return;
    }
    if (valu_partner == 2) return;         // (the VALU waves alone: their own rate)
    float a[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) a[k] = w[k * 64 + lane];
    float b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = w[2048 + k * 64 + lane];
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            // B row broadcast: register b[k / 4], 16-lane row k % 4 (BLGP = 4 + row)
            switch (k & 3) {
            case 0: acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k >> 2], acc[k % CHAINS], 0, 0, 4); break;
            case 1: acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k >> 2], acc[k % CHAINS], 0, 0, 5); break;
            case 2: acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k >> 2], acc[k % CHAINS], 0, 0, 6); break;
            default: acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k >> 2], acc[k % CHAINS], 0, 0, 7); break;
            }
        }
    }
    const long long t1 = clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) s += acc[c];
    if (lane == 0) cycles[wave] = t1 - t0;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// the outer-product form of the weight gradients: 18 accumulators, A with CBSZ = 2 / ABID = t, B with BLGP = 4 + t
template <int T> __device__ __forceinline__ f32x4 outer(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, T, 4 + T); }
template <bool BCAST>
__global__ __launch_bounds__(1024) void outer_kernel(const float *w, float *out, long long *cycles, int iters, int waves_per_simd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((wave >> 2) >= waves_per_simd) return;
    float a[6], b[3];
    for (int k = 0; k < 6; ++k) a[k] = w[k * 64 + lane];
    for (int k = 0; k < 3; ++k) b[k] = w[1024 + k * 64 + lane];
    f32x4 acc[6][3];
    for (int q = 0; q < 6; ++q) for (int c = 0; c < 3; ++c) acc[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (BCAST) {
                    acc[q][c] = outer<0>(a[q], b[c], acc[q][c]); acc[q][c] = outer<1>(a[q], b[c], acc[q][c]);
                    acc[q][c] = outer<2>(a[q], b[c], acc[q][c]); acc[q][c] = outer<3>(a[q], b[c], acc[q][c]);
                } else {
                    for (int t = 0; t < 4; ++t) acc[q][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q], b[c], acc[q][c], 0, 0, 0);
                }
            }
    }
    const long long t1 = clock64();
    f32x4 s = acc[0][0];
    for (int q = 0; q < 6; ++q) for (int c = 0; c < 3; ++c) s += acc[q][c];
    if (lane == 0) cycles[wave] = t1 - t0;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <bool BCAST>
__global__ __launch_bounds__(1024) void outer_kernel_interleaved(const float *w, float *out, long long *cycles, int iters, int waves_per_simd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((wave >> 2) >= waves_per_simd) return;
    float a[6], b[3];
    for (int k = 0; k < 6; ++k) a[k] = w[k * 64 + lane];
    for (int k = 0; k < 3; ++k) b[k] = w[1024 + k * 64 + lane];
    f32x4 acc[6][3];
    for (int q = 0; q < 6; ++q) for (int c = 0; c < 3; ++c) acc[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {                  // trial-major: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = outer<0>(a[q], b[c], acc[q][c]);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = outer<1>(a[q], b[c], acc[q][c]);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = outer<2>(a[q], b[c], acc[q][c]);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = outer<3>(a[q], b[c], acc[q][c]);
    }
    const long long t1 = clock64();
    f32x4 s = acc[0][0];
    for (int q = 0; q < 6; ++q) for (int c = 0; c < 3; ++c) s += acc[q][c];
    if (lane == 0) cycles[wave] = t1 - t0;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

static int check(const char *what, const float *got, const float *want) {
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (got[i] != want[i]) ++bad;
    printf("%-64s %s\n", what, bad ? "MISMATCH" : "ok");
    return bad;
}

int main() {
    float ha[64], hb[64], hd[256], want[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    int bad = 0;
    // (1) plain: D_b[i][j] = A(b,i) * B(b,j), register i of lane (b,j)
    hipLaunchKernelGGL((probe_kernel<0, 0, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int b = 0; b < 16; ++b) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) want[(4 * b + j) * 4 + i] = ha[4 * b + i] * hb[4 * b + j];
    bad += check("layout: D reg i of lane (b,j) = A(b,i) * B(b,j)", hd, want);
    // (2a) BLGP = 4 + r: row r (lanes 16r..16r+15) of B for every row of blocks
    hipLaunchKernelGGL((probe_kernel<0, 0, 6>), dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int b = 0; b < 16; ++b) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) want[(4 * b + j) * 4 + i] = ha[4 * b + i] * hb[16 * 2 + 4 * (b & 3) + j];
    bad += check("BLGP=6: B taken from lanes 32..47 (block b & 3 of row 2)", hd, want);
    // (2b) CBSZ = 4, ABID = 5: block 5's A for all 16 blocks
    hipLaunchKernelGGL((probe_kernel<4, 5, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int b = 0; b < 16; ++b) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) want[(4 * b + j) * 4 + i] = ha[4 * 5 + i] * hb[4 * b + j];
    bad += check("CBSZ=4 ABID=5: A of block 5 broadcast to all blocks", hd, want);
    // (2c) CBSZ = 2, ABID = 1: within groups of 4 blocks, block 1 of the group
    hipLaunchKernelGGL((probe_kernel<2, 1, 0>), dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int b = 0; b < 16; ++b) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) want[(4 * b + j) * 4 + i] = ha[4 * ((b & ~3) + 1) + i] * hb[4 * b + j];
    bad += check("CBSZ=2 ABID=1: A of block 1 of each group of 4 blocks", hd, want);

    // (3) issue rate
    float *w, *out; long long *cyc;
    hipMalloc(&w, 4096 * 4); hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 128);
    float hw[4096];
    for (int i = 0; i < 4096; ++i) hw[i] = 1e-3f * (float)((i * 37) % 101);
    hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
    const int iters = 2000;
    long long hc[8];
    hipMemset(cyc, 0, 64);
    hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, 2);
    hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, 2);
    hipDeviceSynchronize();
    hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
    printf("VALU wave alone: %.2f ticks per exp2+rcp+add+fma group\n", (double)hc[4] / (64.0 * iters));
    for (int chains = 1; chains <= 4; chains *= 2) {            // two MFMA waves per SIMD: does the pipe take one 4x4x1 per 8 cycles or more?
        hipMemset(cyc, 0, 64);
        for (int rep = 0; rep < 2; ++rep) {
            if (chains == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, 3);
            if (chains == 2) hipLaunchKernelGGL((rate_kernel<2>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, 3);
            if (chains == 4) hipLaunchKernelGGL((rate_kernel<4>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, 3);
        }
        hipDeviceSynchronize();
        hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
        // (the older wave wins the arbitration and runs at its own rate; the pipe's rate is what BOTH needed: the slower wave's time)
        printf("TWO MFMA waves per SIMD, chains=%d: %.2f ticks per MFMA of wave 0, %.2f of wave 4 -> the SIMD's pipe took one MFMA per %.2f ticks\n", chains,
               (double)hc[0] / (32.0 * iters), (double)hc[4] / (32.0 * iters), (double)(hc[0] > hc[4] ? hc[0] : hc[4]) / (64.0 * iters));
    }
    for (int wps = 1; wps <= 4; ++wps) {
        hipMemset(cyc, 0, 128);
        hipLaunchKernelGGL((outer_kernel_interleaved<true>), dim3(256), dim3(1024), 0, 0, w, out, cyc, 500, wps);
        hipLaunchKernelGGL((outer_kernel_interleaved<true>), dim3(256), dim3(1024), 0, 0, w, out, cyc, 500, wps);
        hipDeviceSynchronize();
        long long hc2[16];
        hipMemcpy(hc2, cyc, 128, hipMemcpyDeviceToHost);
        long long slowest = 0;
        for (int wv = 0; wv < 4 * wps; wv += 4) slowest = hc2[wv] > slowest ? hc2[wv] : slowest;      // waves 0, 4, 8, 12 share SIMD 0's pipe
        printf("outer-product form (CBSZ=2, ABID=t, BLGP=4+t), 18 accumulators, %d wave(s) per SIMD: wave 0 %.2f ticks per MFMA, the slowest wave %.2f -> the pipe took one per %.2f ticks\n", wps,
               (double)hc2[0] / (72.0 * 500), (double)slowest / (72.0 * 500), (double)slowest / (72.0 * 500 * wps));
    }
    for (int partner = 0; partner < 2; ++partner) {
        for (int chains = 1; chains <= 4; chains *= 2) {
            hipMemset(cyc, 0, 64);
            for (int rep = 0; rep < 2; ++rep) {
                if (chains == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, partner);
                if (chains == 2) hipLaunchKernelGGL((rate_kernel<2>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, partner);
                if (chains == 4) hipLaunchKernelGGL((rate_kernel<4>), dim3(256), dim3(512), 0, 0, w, out, cyc, iters, partner);
            }
            hipDeviceSynchronize();
            hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
            printf("chains=%d valu_partner=%d: %.2f cycles per MFMA (wave 0; clock64 ticks / %d MFMAs)", chains, partner, (double)hc[0] / (32.0 * iters), 32 * iters);
            if (partner) printf("; partner wave: %.2f ticks per exp2+rcp+add+fma group (alone it needs ~24-28)", (double)hc[4] / (64.0 * iters));
            printf("\n");
        }
    }
    return bad ? 1 : 0;
}
