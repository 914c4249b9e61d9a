// granule_litmus.hip -- is a 16-byte granule written by ONE buffer_store_dwordx4 observed WHOLE by ONE buffer_load_dwordx4 on another CU?
// The forward scans' self-validating exchange (csrc/nsd_scan2.hip, "the exchange ring of the group") carries the step tag only in the
// upper 8 bytes of a 16-byte granule {h0 half | h1 half + tag}: a consumer that saw the new upper half together with a stale lower half
// would compute on a stale h0 without noticing.  This litmus looks for exactly that tear.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/granule_litmus tools/micro/granule_litmus.hip && tools/micro/granule_litmus
// 128 producer workgroups rewrite their 4-KB block (256 lanes x one granule {c, c, c, c}, c = 1, 2, 3, ...) as fast as they can; 128
// consumer workgroups load granules of ONE producer block over and over (sc1 = L1-bypassing loads, as the scans do) and count
//   torn        granules whose four dwords are not all equal (any split of the 16 bytes)
//   torn_halves granules whose upper 8 bytes differ from the lower 8 (the split the scans depend on)
//   changes     loads that returned a different counter than the lane's previous load (proof that the data was moving under the loads)
// Variants = the two modes of the scans: plain stores + sc1 loads with producer and consumer on ONE XCD (block ids equal mod 8), and sc1
// (write-through) stores + sc1 loads with the consumer on ANOTHER XCD; plus the two crossed combinations.  Every loop is bounded: consumers
// do a fixed number of loads, producers stop when all consumers are done or after MAX_ITERS stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int PRODUCERS = 128, CONSUMERS = 128, LOADS = 4096;
constexpr unsigned MAX_ITERS = 1u << 22;

struct Counters { unsigned long long observations, torn, torn_halves, changes, same_xcd_pairs, producer_iters; };

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

template <bool SC1_STORE, int XCD_SHIFT>
__global__ __launch_bounds__(256) void litmus_kernel(u32x4 *ring, unsigned *done, unsigned *xcc_of, Counters *out) {
    const int tid = threadIdx.x;
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;       // HW_REG_XCC_ID[3:0]
    if (blockIdx.x < PRODUCERS) {
        const rsrc_t r = make_rsrc(ring + (long)blockIdx.x * 256, 4096);
        if (tid == 0) __hip_atomic_store(xcc_of + blockIdx.x, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned c = 1;
        for (; c < MAX_ITERS; ++c) {
            const u32x4 v = {c, c, c, c};
            if (SC1_STORE) __builtin_amdgcn_raw_buffer_store_b128(v, r, tid * 16, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(v, r, tid * 16, 0, 0);
            if ((c & 63u) == 0u) {
                const unsigned d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d >= (unsigned)CONSUMERS) break;
            }
        }
        if (tid == 0) atomicAdd(&out->producer_iters, (unsigned long long)c);
        return;
    }
    // consumer i reads producer (i + XCD_SHIFT) % PRODUCERS: blocks whose ids are equal mod 8 share an XCD (checked with XCC_ID below)
    const int i = blockIdx.x - PRODUCERS, p = (i + XCD_SHIFT) % PRODUCERS;
    const rsrc_t r = make_rsrc(ring + (long)p * 256, 4096);
    unsigned pxcc = 0;
    for (unsigned spins = 0; spins < (1u << 20) && pxcc == 0; ++spins) pxcc = __hip_atomic_load(xcc_of + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long torn = 0, halves = 0, changes = 0;
    unsigned prev = 0;
    for (int k = 0; k < LOADS; ++k) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, ((tid + k) & 255) * 16, 0, 16);      // sc1: bypasses the CU's L1
        torn += (v[0] != v[1] || v[1] != v[2] || v[2] != v[3]) ? 1 : 0;
        halves += (v[0] != v[2] || v[1] != v[3]) ? 1 : 0;
        changes += (v[0] != prev) ? 1 : 0;
        prev = v[0];
    }
    atomicAdd(&out->observations, (unsigned long long)LOADS);
    if (torn) atomicAdd(&out->torn, torn);
    if (halves) atomicAdd(&out->torn_halves, halves);
    atomicAdd(&out->changes, changes);
    if (tid == 0 && pxcc == xcc + 1u) atomicAdd(&out->same_xcd_pairs, 1ull);
    __syncthreads();
    if (tid == 0) atomicAdd(done, 1u);
}

template <bool SC1_STORE, int XCD_SHIFT>
static void run(const char *what, u32x4 *ring, unsigned *done, unsigned *xcc_of, Counters *dout, int repeats) {
    Counters tot = {0, 0, 0, 0, 0, 0};
    for (int rep = 0; rep < repeats; ++rep) {
        hipMemset(ring, 0, PRODUCERS * 4096);
        hipMemset(done, 0, 4);
        hipMemset(xcc_of, 0, PRODUCERS * 4);
        hipMemset(dout, 0, sizeof(Counters));
        hipLaunchKernelGGL((litmus_kernel<SC1_STORE, XCD_SHIFT>), dim3(PRODUCERS + CONSUMERS), dim3(256), 0, 0, ring, done, xcc_of, dout);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        Counters c;
        hipMemcpy(&c, dout, sizeof(c), hipMemcpyDeviceToHost);
        tot.observations += c.observations; tot.torn += c.torn; tot.torn_halves += c.torn_halves; tot.changes += c.changes;
        tot.same_xcd_pairs += c.same_xcd_pairs; tot.producer_iters += c.producer_iters;
    }
    printf("{\"variant\": \"%s\", \"observations\": %llu, \"torn\": %llu, \"torn_halves\": %llu, \"changes\": %llu, "
           "\"consumer_workgroups_on_the_producers_xcd\": \"%llu of %d\", \"producer_stores_per_lane\": %llu}\n",
           what, tot.observations, tot.torn, tot.torn_halves, tot.changes, tot.same_xcd_pairs, repeats * CONSUMERS,
           tot.producer_iters / (unsigned long long)(repeats * PRODUCERS));
}

int main(int argc, char **argv) {
    const int repeats = argc > 1 ? atoi(argv[1]) : 4;
    u32x4 *ring; unsigned *done, *xcc_of; Counters *dout;
    hipMalloc(&ring, PRODUCERS * 4096);
    hipMalloc(&done, 4);
    hipMalloc(&xcc_of, PRODUCERS * 4);
    hipMalloc(&dout, sizeof(Counters));
    run<false, 0>("plain store, sc1 load, same XCD (the scans' one-L2 mode)", ring, done, xcc_of, dout, repeats);
    run<true, 1>("sc1 store, sc1 load, other XCD (the scans' write-through mode)", ring, done, xcc_of, dout, repeats);
    run<true, 0>("sc1 store, sc1 load, same XCD", ring, done, xcc_of, dout, repeats);
    run<false, 1>("plain store, sc1 load, other XCD (NOT coherent by design: shown for contrast, the scans never do this)", ring, done, xcc_of, dout, repeats);
    return 0;
}
