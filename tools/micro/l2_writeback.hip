// l2_writeback.hip -- does a line that is overwritten again and again stay (dirty) in the XCD's L2, or does every store leave it?
// And what does streaming traffic through the same L2 do to such lines, with and without the nt hint?
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/l2_writeback tools/micro/l2_writeback.hip
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out -- tools/micro/l2_writeback      (and FETCH_SIZE in a second pass)
// Each workgroup owns a 4-KB slice of a small "ring" buffer (256 workgroups -> 1 MB in all, 128 KB per XCD) and rewrites + rereads it
// REPS times; variants: store kind (plain / sc1 / nt), and a stream of STREAM_KB per repetition through the same CU from a big buffer
// (plain or nt loads + stores).  Bytes stored into the ring = 256 * 4 KB * REPS whatever the variant; WRITE_SIZE tells what left the L2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int STORE_KIND, int STREAM_KIND>     // store: 0 plain, 1 sc1 (write-through), 2 nt ; stream: 0 none, 1 plain, 2 nt
__global__ __launch_bounds__(256) void ring_kernel(u32x4 *ring, u32x4 *big, long big_per_wg, int reps, int stream_vec, u32x4 *sink) {
    u32x4 *mine = ring + (long)blockIdx.x * 256;                  // 256 lanes x 16 B = 4 KB
    u32x4 *str = big + (long)blockIdx.x * big_per_wg;
    u32x4 acc = {0u, 0u, 0u, 0u};
    long sp = 0;
    for (int r = 0; r < reps; ++r) {
        const u32x4 v = {(unsigned)r, threadIdx.x, blockIdx.x, acc[0]};
        if (STORE_KIND == 0) mine[threadIdx.x] = v;
        else if (STORE_KIND == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(mine + threadIdx.x), "v"(v) : "memory");
        else __builtin_nontemporal_store(v, mine + threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x4 back;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(back) : "v"(mine + ((threadIdx.x + 64) & 255)) : "memory");
        acc += back;
        if (STREAM_KIND != 0) {
            for (int i = 0; i < stream_vec; ++i) {
                u32x4 *p = str + sp + (long)i * 256 + threadIdx.x;
                u32x4 w;
                if (STREAM_KIND == 1) { w = *p; w[0] += 1u; *p = w; }
                else { w = __builtin_nontemporal_load(p); w[0] += 1u; __builtin_nontemporal_store(w, p); }
            }
            sp += (long)stream_vec * 256;
            if (sp + (long)stream_vec * 256 > big_per_wg) sp = 0;
        }
    }
    if (acc[1] == 0xdeadbeefu) sink[0] = acc;
}

int main() {
    const int wgs = 256, reps = 400;
    u32x4 *ring, *big, *sink;
    const long big_per_wg = 4L * 1024 * 1024 / 16;                // 4 MB per workgroup, 1 GB in all
    hipMalloc(&ring, wgs * 4096);
    hipMalloc(&big, (size_t)wgs * big_per_wg * 16);
    hipMalloc(&sink, 64);
    hipMemset(ring, 0, wgs * 4096);
    hipMemset(big, 0, (size_t)wgs * big_per_wg * 16);
    hipDeviceSynchronize();
    const int sv = 4;                                             // 4 x 4 KB = 16 KB streamed (read + written) per repetition and workgroup
    hipLaunchKernelGGL((ring_kernel<0, 0>), dim3(wgs), dim3(256), 0, 0, ring, big, big_per_wg, reps, sv, sink);
    hipLaunchKernelGGL((ring_kernel<1, 0>), dim3(wgs), dim3(256), 0, 0, ring, big, big_per_wg, reps, sv, sink);
    hipLaunchKernelGGL((ring_kernel<2, 0>), dim3(wgs), dim3(256), 0, 0, ring, big, big_per_wg, reps, sv, sink);
    hipLaunchKernelGGL((ring_kernel<0, 1>), dim3(wgs), dim3(256), 0, 0, ring, big, big_per_wg, reps, sv, sink);
    hipLaunchKernelGGL((ring_kernel<0, 2>), dim3(wgs), dim3(256), 0, 0, ring, big, big_per_wg, reps, sv, sink);
    hipLaunchKernelGGL((ring_kernel<1, 2>), dim3(wgs), dim3(256), 0, 0, ring, big, big_per_wg, reps, sv, sink);
    hipDeviceSynchronize();
    printf("ring bytes stored per launch: %.1f MB; streamed (variants 3-5): %.1f MB read + as much written\n", wgs * 4096.0 * reps / 1e6,
           (double)wgs * reps * sv * 4096 / 1e6);
    return 0;
}
