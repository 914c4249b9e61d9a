// v_dot2c_f32_bf16 as an "unpack + add": acc_bf16x2(lo, hi, w) against shifts and adds, on a few thousand random words.
//   hipcc --offload-arch=gfx950 -O3 -I neural-speech-decoding_amd/csrc -I include tools/micro/dot2_check.hip -o /tmp/dot2_check && /tmp/dot2_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "nsd_bf16.h"
__global__ void k(const unsigned *w, const float *acc, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo = acc[2 * i], hi = acc[2 * i + 1];
    acc_bf16x2(lo, hi, w[i]);
    out[2 * i] = lo; out[2 * i + 1] = hi;
}
int main() {
    const int n = 1 << 14;
    unsigned *hw = (unsigned *)malloc(n * 4); float *ha = (float *)malloc(n * 8), *ho = (float *)malloc(n * 8);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float a = (rand() / (float)RAND_MAX - 0.5f) * powf(10.f, (rand() % 12) - 8), b = (rand() / (float)RAND_MAX - 0.5f) * powf(10.f, (rand() % 12) - 8);
        unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
        hw[i] = (ua >> 16) | (ub & 0xffff0000u);
        ha[2 * i] = (rand() / (float)RAND_MAX - 0.5f) * powf(10.f, (rand() % 12) - 8); ha[2 * i + 1] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    }
    unsigned *dw; float *da, *dout;
    hipMalloc(&dw, n * 4); hipMalloc(&da, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(dw, hw, n * 4, hipMemcpyHostToDevice); hipMemcpy(da, ha, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dw, da, dout, n);
    hipMemcpy(ho, dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
        unsigned l = hw[i] << 16, h = hw[i] & 0xffff0000u; float fl, fh; memcpy(&fl, &l, 4); memcpy(&fh, &h, 4);
        const float rl = ha[2 * i] + fl, rh = ha[2 * i + 1] + fh;
        if (ho[2 * i] != rl || ho[2 * i + 1] != rh) {
            if (bad < 5) printf("i=%d w=%08x acc=(%g,%g) got (%g,%g) want (%g,%g)\n", i, hw[i], ha[2 * i], ha[2 * i + 1], ho[2 * i], ho[2 * i + 1], rl, rh);
            ++bad;
            double e = fabs((double)ho[2 * i] - rl) / (fabs(rl) + 1e-30); if (e > worst) worst = e;
            e = fabs((double)ho[2 * i + 1] - rh) / (fabs(rh) + 1e-30); if (e > worst) worst = e;
        }
    }
    printf("dot2 accumulate: %d of %d words differ from shift + add (worst relative difference %.3g)\n", bad, n, worst);
    return 0;
}
