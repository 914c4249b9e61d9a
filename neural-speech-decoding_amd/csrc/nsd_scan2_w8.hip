// nsd_scan2_w8.hip -- the fused two-layer BACKWARD scan with EIGHT waves per workgroup (two per SIMD), H = 256.
//
// Why: the four-wave kernel (nsd_scan2.hip, scan2_bwd_kernel) runs one wave per SIMD, and a wave's own VALU / memory instructions do
// not overlap its MFMAs -- putting the step's conversions, the dropout hashes or a whole cell into the gaps of the MFMA stream left
// the step's length unchanged (in-kernel stamps; DESIGN.md, finding 14).  The matrix pipe is busy ~1 500 of a step's ~10 000 cycles;
// the rest is instruction issue (~1 900 instructions) and exposed latencies (flag poll, partial-sum loads, barrier).  Only a SECOND
// wave on the SIMD fills those slots.  A wave of the four-wave kernel holds 192 weight registers (the rows of W^T of TWO consumers)
// + 48 accumulators + ~250 VGPRs; here a wave holds ONE consumer's rows (96) and owns two of its lanes' four units, so two waves
// fit a SIMD's 512 registers:
//   wave w = v + 4 * half (v = 0..3, half = 0..1):  MFMA role: consumer row tile r = w (P = 8 members, one each);
//                                                   cell role: units 2 half, 2 half + 1 of lane (trial, hh) of the four-wave kernel's wave v.
// Everything a lane touches keeps the four-wave layout (saved activations [half][lane][16 B], cell state [lane][2 x 4 B], the da
// tile in LDS [k-step 2 v + half], row-major da), so the forward scan and the weight-gradient GEMMs do not change.  Only the
// partial-sum ring is private to the kernel:
//   block (consumer member, producer member, consumer wave v) = [lane 64][half 2] x 12 B = {rec1 | din0 | rec0} of 2 units:
//   a producer stores a lane's 24 bytes (16 + 8), a consumer wave loads its 12 with one instruction per producer.
// Protocol (publish flags, consume counters, rendezvous, time-outs) as in nsd_scan2.hip, with 8 P words per group instead of 4 P.
#include "nsd_scan_common.h"

namespace {

constexpr int ST2W_BWD_TIMEOUT = 2;
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

// a wave-uniform pointer, told to the compiler (SGPR pair: the access takes the scalar-base form with a 32-bit lane offset)
template <class T>
__device__ __forceinline__ T *uni(T *p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}

struct CellFac2 { float A[2], Fi[2], Ff[2], Fg[2], Fo[2], fg[2]; };
// (the same arithmetic as cell_factors / cell_apply of nsd_scan_common.h, for the two units a lane owns here)
__device__ __forceinline__ void cell_factors2(const u32x4 gq, const unsigned cq, const unsigned cpq, CellFac2 &f) {
    const float cv[2] = {bf16_lo(cq), bf16_hi(cq)}, cp[2] = {bf16_lo(cpq), bf16_hi(cpq)};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned w0 = gq[2 * j], w1 = gq[2 * j + 1];
        const float ig = fabsf(bf16_lo(w0)), fg = bf16_hi(w0), gg = bf16_lo(w1), og = bf16_hi(w1);
        const float tc = fast_tanh(cv[j]);
        f.A[j] = og * (1.f - tc * tc);
        f.Fi[j] = gg * ig * (1.f - ig);
        f.Ff[j] = cp[j] * fg * (1.f - fg);
        f.Fg[j] = ig * (1.f - gg * gg);
        f.Fo[j] = tc * og * (1.f - og);
        f.fg[j] = fg;
    }
}
__device__ __forceinline__ void pin(CellFac2 &f) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { pin(f.A[j]); pin(f.Fi[j]); pin(f.Ff[j]); pin(f.Fg[j]); pin(f.Fo[j]); pin(f.fg[j]); }
}
// (the bias-gradient sums live in LDS -- 16 registers a wave does not have: [unit 2][thread][gate 4], a thread's own slots, read /
// add / written with 16-byte accesses; LDS float atomics took ~700 cycles each here)
__device__ __forceinline__ void cell_apply2(const CellFac2 &f, const float (&dh)[2], float (&dc)[2], f32x4 *dbs, unsigned (&dw)[4]) {
    f32x4 acc[2] = {dbs[0], dbs[512]};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float dct = fmaf(dh[j], f.A[j], dc[j]);
        dc[j] = dct * f.fg[j];
        const float dai = dct * f.Fi[j], daf = dct * f.Ff[j], dag = dct * f.Fg[j], dao = dh[j] * f.Fo[j];
        dw[2 * j] = pack_bf16x2(dai, daf);
        dw[2 * j + 1] = pack_bf16x2(dag, dao);
        acc[j][0] += dai; acc[j][1] += daf; acc[j][2] += dag; acc[j][3] += dao;
    }
    dbs[0] = acc[0]; dbs[512] = acc[1];
}

template <int H>
__global__ __launch_bounds__(512) void scan2_bwd8_kernel(const Scan2BwdArgs a) {
    constexpr int P = H / 32, G = 4 * H, KS = 8;
    static_assert(P == 8, "one consumer row tile per wave");
    constexpr long NBLK = (long)P * P * 4, SLOT_BYTES = NBLK * 1536;
    auto blk_off = [](const int cons, const int prod, const int v) { return (unsigned)(((cons * P + prod) * 4 + v) * 1536); };
    // the workgroup's own da of the step as MFMA B operands: k-step 2 v + half = the 1-KB lane-linear block wave (v, half) writes
    __shared__ __align__(16) bf16_t dab[2][2][KS][512];          // [step parity][layer][k-step][lane * 8]
    __shared__ __align__(16) f32x4 dbsl[2][2][512];              // bias-gradient sums [layer][the lane's unit][thread] x 4 gates
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63, w8 = __builtin_amdgcn_readfirstlane(tid >> 6), v = w8 & 3, half = w8 >> 2;
    const Member me = member_of(blockIdx.x, a.groups, P, a.spread_groups);
    const int b0 = (a.group0 + me.group) * 32;
    const int col = lane & 31, hh = lane >> 5;

    // rows of W^T of consumer r = w8 (its 32 units), columns = this workgroup's 128 gate columns in k-step order
    bf16x8 wq1[KS], wqx[KS], wq0[KS];
    {
        const long ro = (long)(32 * w8 + col) * G + 128 * me.p + 16 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int co = 32 * (ks >> 1) + 8 * (ks & 1);
            wq1[ks] = *reinterpret_cast<const bf16x8 *>(a.wb1 + ro + co);
            wqx[ks] = *reinterpret_cast<const bf16x8 *>(a.wxt1 + ro + co);
            wq0[ks] = *reinterpret_cast<const bf16x8 *>(a.wb0 + ro + co);
        }
    }
    float dc1[2] = {0.f, 0.f}, dc0[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) { dbsl[1][j][tid] = f32x4{0.f, 0.f, 0.f, 0.f}; dbsl[0][j][tid] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (tid == 0) s_abort = 0;
    __syncthreads();

    unsigned *gflags = a.flags + (long)me.group * GROUP_WORDS;
    const int rv = group_rendezvous<P>(gflags, me.p, w8, lane);
    if (rv < 0 && lane == 0) { s_abort = 1; report_timeout(a.status, ST2W_BWD_TIMEOUT); }
    __syncthreads();
    if (s_abort) return;
    const bool same_l2 = rv == 1 && a.allow_l2_mode != 0;
    if (tid == 0 && me.p == 0) atomicAdd(a.status + (rv == 1 ? 2 : 3), 1);
    const int T = a.T;
    const int u0 = 32 * me.p + 8 * v + 4 * hh + 2 * half;       // the lane's two units
    float dpl[2], aw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { aw[j] = a.attn_w[u0 + j]; dpl[j] = a.dpooled[(long)(b0 + col) * H + u0 + j]; }
    const char *ring0 = reinterpret_cast<const char *>(a.xch) + (long)(a.group0 + me.group) * SLOT_BYTES;
    unsigned *gacks = gflags + ACK_WORD;                        // consume counters of the group (the ring has ONE slot)
    const float keep0 = a.rng.on ? a.rng.keep_lstm : 1.f;

    // saved activations / upstream terms of a step: unconditional requests with clamped addresses (exact vmcnt, see nsd_scan2.hip)
    struct Saved { u32x4 q1, q0; unsigned cq1, cp1, cq0, cp0; float al, ds; };
    auto load_saved = [&](const int sx, Saved &sv) {
        const int x1 = T - 1 - sx > 0 ? T - 1 - sx : 0, x0 = T - sx < T ? T - sx : T - 1;
        {
            const long row_u = ((long)(b0 >> 5) * T + x1) * 32;
            const long blk = saved_block(b0 >> 5, P, me.p, T, x1, v);
            const bf16_t *gs = uni(a.ga1 + blk * 1024 + half * 512), *cs = uni(a.cs1 + blk * 256 + 2 * half), *cp = uni(a.cs1 + (x1 == 0 ? blk : blk - 4) * 256 + 2 * half);
            sv.q1 = ld_stream<u32x4>(gs + lane * 8);
            sv.cq1 = ld_stream<unsigned>(cs + lane * 4);
            sv.cp1 = ld_stream<unsigned>(cp + lane * 4);
            sv.al = ld_stream<float>(uni(a.alpha + row_u) + col); sv.ds = ld_stream<float>(uni(a.dscore + row_u) + col);
        }
        {
            const long blk = saved_block(b0 >> 5, P, me.p, T, x0, v);
            const bf16_t *gs = uni(a.ga0 + blk * 1024 + half * 512), *cs = uni(a.cs0 + blk * 256 + 2 * half), *cp = uni(a.cs0 + (x0 == 0 ? blk : blk - 4) * 256 + 2 * half);
            sv.q0 = ld_stream<u32x4>(gs + lane * 8);
            sv.cq0 = ld_stream<unsigned>(cs + lane * 4);
            sv.cp0 = ld_stream<unsigned>(cp + lane * 4);
        }
    };
    Saved sv;
    load_saved(0, sv);
    Stamps stp;
    stp.start();
    for (int s = 0; s <= T; ++s) {
        const bool do1 = s < T, do0 = s >= 1;
        const int t1 = T - 1 - s, t0 = T - s;
        // ---- ahead of the exchange: everything of the two cells that does not need dh
        CellFac2 f1, f0;
        float dup1[2], m0[2];
        if (T - 1 - s <= 0) sv.cp1 = 0u;                        // c_{t-1} of t = 0 is the zero state
        if (T - s <= 0) sv.cp0 = 0u;
        cell_factors2(sv.q1, sv.cq1, sv.cp1, f1);
        cell_factors2(sv.q0, sv.cq0, sv.cp0, f0);
#pragma unroll
        for (int j = 0; j < 2; ++j) dup1[j] = do1 ? fmaf(sv.al, dpl[j], sv.ds * aw[j]) : 0.f;
        m0[0] = (sv.q0[0] & 0x8000u) ? keep0 : 0.f; m0[1] = (sv.q0[2] & 0x8000u) ? keep0 : 0.f;     // (saved_keep_bits: the sign of the saved i)
        pin(f1); pin(f0);
#pragma unroll
        for (int j = 0; j < 2; ++j) { pin(dup1[j]); pin(m0[j]); }
        stp.mark(7);
        float drec1[2] = {0.f, 0.f}, dinx[2] = {0.f, 0.f}, drec0[2] = {0.f, 0.f};
        if (s >= 1) {
            if (!wait_group<8 * P>(gflags, (unsigned)s, lane) && lane == 0) { s_abort = 1; report_timeout(a.status, ST2W_BWD_TIMEOUT); }
            stp.mark(0);
            // the partial sums the P members sent this wave's lanes at step s-1, added in member order
            const nsd_rsrc rr = make_rsrc(ring0, (unsigned)SLOT_BYTES);
            const int voff = 24 * lane + 12 * half;              // (the block's offset is wave-uniform: the instruction's scalar offset)
#pragma unroll
            for (int qb = 0; qb < P; qb += 4) {                  // two batches of four producers: 12 registers in flight instead of 24
                u32x3 pv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[q] = __builtin_amdgcn_raw_buffer_load_b96(rr, voff, (int)blk_off(me.p, qb + q, v), 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc_bf16x2(drec1[0], drec1[1], pv[q][0]);
                    acc_bf16x2(dinx[0], dinx[1], pv[q][1]);
                    acc_bf16x2(drec0[0], drec0[1], pv[q][2]);
                }
            }
            // this wave has taken its partial sums of step s-1 out of the ring: the producers may rewrite the slot
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_xchg_u32(same_l2, gacks + 8 * me.p + w8, (unsigned)s);
            stp.mark(1);
        }
        // ---- the dh-dependent rest of both cells: da1_{t1}, da0_{t0}
        unsigned dw1[4] = {0u, 0u, 0u, 0u}, dw0[4] = {0u, 0u, 0u, 0u};
        if (do1) {
            const float dh[2] = {dup1[0] + drec1[0], dup1[1] + drec1[1]};
            cell_apply2(f1, dh, dc1, &dbsl[1][0][tid], dw1);
        }
        if (do0) {
            const float dh[2] = {fmaf(dinx[0], m0[0], drec0[0]), fmaf(dinx[1], m0[1], drec0[1])};
            cell_apply2(f0, dh, dc0, &dbsl[0][0][tid], dw0);
        }
        stp.mark(2);
        // row-major da for the weight-gradient GEMMs: the lane's 8 gate columns of each layer
        auto store_da_rows = [&]() {
            const unsigned lane_off = (unsigned)(col * G + 4 * u0);
            if (do1) st_stream<u32x4>(uni(a.da1 + ((long)(b0 >> 5) * T + t1) * 32 * G) + lane_off, u32x4{dw1[0], dw1[1], dw1[2], dw1[3]});
            if (do0) st_stream<u32x4>(uni(a.da0 + ((long)(b0 >> 5) * T + t0) * 32 * G) + lane_off, u32x4{dw0[0], dw0[1], dw0[2], dw0[3]});
        };
        if (s < T) {                                            // (after the last step nobody reads a partial sum)
            const int par = s & 1;
            *reinterpret_cast<u32x4 *>(&dab[par][1][2 * v + half][lane * 8]) = u32x4{dw1[0], dw1[1], dw1[2], dw1[3]};
            *reinterpret_cast<u32x4 *>(&dab[par][0][2 * v + half][lane * 8]) = u32x4{dw0[0], dw0[1], dw0[2], dw0[3]};
            __syncthreads();
            if (s_abort) break;
            stp.mark(3);
            unsigned ackv = ld_sc1_u32(gacks + lane);           // 8 P = 64 consume counters: one per lane, requested now, looked at before the ring stores
            __builtin_amdgcn_sched_barrier(0);                  // FIRST in the queue: its wait must not include the HBM-bound requests below
            store_da_rows();
            load_saved(s + 1, sv);
            __builtin_amdgcn_sched_barrier(0);
            // ---- partial sums of dh for the 32 units of consumer w8 from this workgroup's 128 + 128 columns: three products of KS MFMAs
            // through ONE accumulator (W_hh1^T da1, W_ih1^T da1, W_hh0^T da0), each packed to bf16 as it completes -- 16 accumulator
            // registers instead of 48 (the other wave of the SIMD fills the pipe meanwhile); one fragment pipeline over all 24 reads
            unsigned pw[3][8];                                   // [product][word]: words 2q, 2q+1 = registers 4q..4q+3 = the 4 units of consumer wave q's lane
            {
                constexpr int D = 2, NTOT = 3 * KS;
                auto frag_of = [&](const int i) { return *reinterpret_cast<const bf16x8 *>(&dab[par][i / KS == 2 ? 0 : 1][i % KS][lane * 8]); };
                bf16x8 f[D];
#pragma unroll
                for (int i = 0; i < D; ++i) f[i] = frag_of(i);
                __builtin_amdgcn_sched_barrier(0);
                f32x16 acc;
                static_for<0, NTOT>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, m = i / KS, ks = i % KS;
                    if constexpr (m == 0) { if constexpr (ks == 0) mfma_new_a(acc, wq1[ks], f[i % D]); else mfma_acc_a(acc, wq1[ks], f[i % D]); }
                    else if constexpr (m == 1) { if constexpr (ks == 0) mfma_new_a(acc, wqx[ks], f[i % D]); else mfma_acc_a(acc, wqx[ks], f[i % D]); }
                    else { if constexpr (ks == 0) mfma_new_a(acc, wq0[ks], f[i % D]); else mfma_acc_a(acc, wq0[ks], f[i % D]); }
                    if constexpr (i + D < NTOT) f[i % D] = frag_of(i + D);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (ks == KS - 1) {
                        mfma_settle(acc);
#pragma unroll
                        for (int k = 0; k < 8; ++k) pw[m][k] = pack_bf16x2(acc[2 * k], acc[2 * k + 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
            stp.mark(4);
            pin(ackv);                                          // the compare stays HERE: at the load it would expose an L2 round trip per step
            if (!__all(ackv >= (unsigned)s)) {                  // (rare: the counters were read ~1 000 cycles after they were written)
                if (!wait_group<8 * P>(gacks, (unsigned)s, lane) && lane == 0) { s_abort = 1; report_timeout(a.status, ST2W_BWD_TIMEOUT); }
            }
            {
                const nsd_rsrc rw = make_rsrc(ring0, (unsigned)SLOT_BYTES);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 lo = {pw[0][2 * q], pw[1][2 * q], pw[2][2 * q], pw[0][2 * q + 1]};
                    const u32x2 hi = {pw[1][2 * q + 1], pw[2][2 * q + 1]};
                    const int so = (int)blk_off(w8, me.p, q);
                    if (same_l2) { __builtin_amdgcn_raw_buffer_store_b128(lo, rw, 24 * lane, so, 0); __builtin_amdgcn_raw_buffer_store_b64(hi, rw, 24 * lane + 16, so, 0); }
                    else { __builtin_amdgcn_raw_buffer_store_b128(lo, rw, 24 * lane, so, 16); __builtin_amdgcn_raw_buffer_store_b64(hi, rw, 24 * lane + 16, so, 16); }
                }
            }
            stp.mark(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_xchg_u32(same_l2, gflags + 8 * me.p + w8, (unsigned)(s + 1));
            stp.mark(6);
        }
        if (s == T) store_da_rows();
    }
    stp.store(a.status, blockIdx.x == 0 && tid == 0);
    // ---- bias gradients of this batch tile, both layers: the lane's 8 gate columns
    float dbs1[8], dbs0[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v1 = dbsl[1][k >> 2][tid][k & 3], v0 = dbsl[0][k >> 2][tid][k & 3];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) { v1 += __shfl_xor(v1, m, 64); v0 += __shfl_xor(v0, m, 64); }
        dbs1[k] = v1; dbs0[k] = v0;
    }
    if (col == 0) {
        float *d1 = a.dbp1 + (long)(a.group0 + me.group) * G + 4 * u0, *d0 = a.dbp0 + (long)(a.group0 + me.group) * G + 4 * u0;
#pragma unroll
        for (int k = 0; k < 8; k += 4) {
            *reinterpret_cast<f32x4 *>(d1 + k) = f32x4{dbs1[k], dbs1[k + 1], dbs1[k + 2], dbs1[k + 3]};
            *reinterpret_cast<f32x4 *>(d0 + k) = f32x4{dbs0[k], dbs0[k + 1], dbs0[k + 2], dbs0[k + 3]};
        }
    }
}

}  // namespace

bool nsd_scan2_bwd8_supported(int H, int MG) { return H == 256 && MG == 32; }

int nsd_scan2_bwd8_launch(const Scan2BwdArgs &a, int H, int MG, hipStream_t st) {
    if (!nsd_scan2_bwd8_supported(H, MG) || a.groups * (H / 32) > nsd_num_cus()) { nsd_set_error("scan2_bwd8: unsupported geometry H=%d MG=%d groups=%d", H, MG, a.groups); return NSD_E_INVALID; }
    const dim3 grid(a.groups * (H / 32) - a.diag_short_grid);
    hipLaunchKernelGGL((scan2_bwd8_kernel<256>), grid, dim3(512), 0, st, a);
    NSD_CHECK_LAUNCH("scan2_bwd8_kernel");
    return NSD_OK;
}
