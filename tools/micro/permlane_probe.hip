#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
    const unsigned lane = threadIdx.x;
    u32x2 a = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    u32x2 b = __builtin_amdgcn_permlane16_swap(lane, 100u + lane, false, false);
    o[lane] = a[0]; o[64 + lane] = a[1]; o[128 + lane] = b[0]; o[192 + lane] = b[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap ret[0]", "permlane32_swap ret[1]", "permlane16_swap ret[0]", "permlane16_swap ret[1]"};
    for (int q = 0; q < 4; ++q) { printf("%s: ", names[q]); for (int l = 0; l < 64; l += 4) printf("%u ", h[64 * q + l]); printf("\n"); }
    return 0;
}
