// nsd_scan_common.h -- what the persistent scan kernels share (nsd_scan.hip: one layer per launch; nsd_scan2.hip: two
// unidirectional layers skewed by one step in one launch): group geometry, the flag protocol, exchange stores.
#pragma once
#include <type_traits>
#include "nsd_seq.h"

// timing experiments only (make ABL=n -> libnsd_hip_abl.so, never shipped): bit 0 skip the flag wait, bit 1 skip the tile
// gather, bit 2 skip the drain of the published stores -- results are wrong by construction, only the time is of interest
#ifndef NSD_SCAN_ABLATE
#define NSD_SCAN_ABLATE 0
#endif

// diagnostic build only (-DNSD_SCAN_STAMPS=1, never shipped): s_memtime stamps around the phases of a scan step, summed by
// wave 0 of workgroup 0 into the words behind the status word (read with tools/seq_stamps.py)
#ifndef NSD_SCAN_STAMPS
#define NSD_SCAN_STAMPS 0
#endif

namespace {

// a loop whose index is a compile-time constant inside the body: f(std::integral_constant<int, I>) for I = I0 .. N-1
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

struct Stamps {
    unsigned long long last, acc[8];
    __device__ __forceinline__ void start() {
        if (NSD_SCAN_STAMPS) { for (int i = 0; i < 8; ++i) acc[i] = 0; __builtin_amdgcn_sched_barrier(0); last = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    }
    // WAITV: also drain the wave's vector-memory queue first, so that the phase ends when its loads have really arrived
    template <bool WAITV = false>
    __device__ __forceinline__ void mark(const int i) {
        if (NSD_SCAN_STAMPS) {
            __builtin_amdgcn_sched_barrier(0);
            if (WAITV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the stamp lands behind the phase's LDS traffic
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            acc[i] += t - last; last = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __device__ __forceinline__ void store(int *status, const bool who) {
        if (NSD_SCAN_STAMPS && who) for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long *>(status + 4)[i] = acc[i];
    }
};

constexpr unsigned SPIN_LIMIT = 1u << 20;          // polls (each >= ~1 us with the sleep): ~1-2 s, then give up
constexpr int ST_FWD_TIMEOUT = 1, ST_BWD_TIMEOUT = 2;
// a scan group gave up: the code goes into the status word of the evaluation AND into the workspace's sticky word (never cleared
// by a forward call: the caller sees the failure whenever it looks, and the guarded Adam update skips until it does)
__device__ __forceinline__ void report_timeout(int *status, const int code) {
    atomicOr(status, code);
    atomicOr(status - NSD_SEQ_HEADER_WORDS, code);
}
constexpr int GROUP_WORDS = NSD_SEQ_GROUP_WORDS, ACK_WORD = NSD_SEQ_ACK_WORD;                   // flag words per group: [0,64) one per wave of every member, [64,80) XCC ids

// ---------------------------------------------------------------------------------------------------------------------------
// group geometry shared by both scans
// ---------------------------------------------------------------------------------------------------------------------------
struct Member { int dir, group, p; };
// Members of a group get block ids that are equal mod 8 where the grid allows it: those blocks are observed to share an XCD,
// so the exchange stays inside one L2.  Speed only -- the protocol does not depend on placement.
__device__ __forceinline__ Member member_of(const int bid, const int groups, const int P, const int spread) {
    Member m;
    const int nper = groups * P;
    m.dir = bid / nper;
    const int rem = bid - m.dir * nper;
    if (!spread && (nper & 7) == 0 && ((nper >> 3) % P) == 0) {
        const int x = rem & 7, slot = rem >> 3;
        m.group = x * ((nper >> 3) / P) + slot / P;
        m.p = slot % P;
    } else {
        m.group = rem / P;
        m.p = rem % P;
    }
    return m;
}

// one wave: wait until every wave of every member of the group has published `need` steps (NW = 4P flag words, one per
// lane).  Returns false on timeout.
template <int NW>
__device__ __forceinline__ bool wait_group(const unsigned *gflags, const unsigned need, const int lane) {
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        const unsigned v = lane < NW ? ld_sc1_u32(gflags + lane) : 0xffffffffu;
        if (__all(v >= need)) return true;
#if !defined(NSD_POLL_NO_SLEEP)
        __builtin_amdgcn_s_sleep(1);
#endif
    }
    return false;
}

// Start of a scan: publish this workgroup's XCC id, wait for the whole group (every wave polls; bounded), report whether the
// group sits on one XCD.  Returns -1 on timeout, else 0 / 1.
template <int P>
__device__ __forceinline__ int group_rendezvous(unsigned *gwords, const int p, const int wave, const int lane) {
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;       // HW_REG_XCC_ID[3:0]
    if (wave == 0 && lane == 0) st_sc1_u32(gwords + 64 + p, xcc + 1u);
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        const unsigned v = lane < P ? ld_sc1_u32(gwords + 64 + lane) : xcc + 1u;
        if (__all(v != 0u)) return __all(v == xcc + 1u) ? 1 : 0;
        __builtin_amdgcn_s_sleep(4);
    }
    return -1;
}
// exchange stores: plain when the group shares an L2, write-through otherwise
__device__ __forceinline__ void st_xchg_u64(const bool same_l2, void *p, const unsigned long long v) {
    if (same_l2) *reinterpret_cast<unsigned long long *>(p) = v; else st_sc1_u64(p, v);
}
__device__ __forceinline__ void st_xchg_u32(const bool same_l2, void *p, const unsigned v) {
    if (same_l2) *reinterpret_cast<unsigned *>(p) = v; else st_sc1_u32(p, v);
}

__device__ __forceinline__ f32x16 unpack_tile(const u32x4 lo, const u32x4 hi) {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = bf16_lo(lo[i]); v[2 * i + 1] = bf16_hi(lo[i]);
        v[8 + 2 * i] = bf16_lo(hi[i]); v[8 + 2 * i + 1] = bf16_hi(hi[i]);
    }
    return v;
}


// MFMA with the weight fragment (A operand) and the accumulator read straight from accumulation registers.  The scans keep
// 128-192 weight registers per lane; hipcc parks most of them in AGPRs and copies each fragment back with four v_accvgpr_read
// before its MFMA -- with one wave per SIMD those copies issue in the MFMA's own slot, and a 32-cycle MFMA gap became ~55
// (2 900 cycles for 48 MFMAs).  As asm statements with "a" constraints the operands are used where they live.  hipcc knows
// nothing about the latency of these statements: mfma_settle(accumulators) before any other instruction reads one (in-place
// accumulation chains and independent accumulators need nothing, as in hipcc's own output).
// (the s_nop in front: hipcc may satisfy an "a" operand by copying it from a VGPR right before the statement, and it does not
// know that the statement is an MFMA reading that copy -- VALU write -> MFMA read needs wait states, found as NaNs when four
// rarely used fragments were copied in that way.  It is free when MFMAs follow each other: the pipe is busy 32 cycles anyway.)
__device__ __forceinline__ void mfma_acc_a(f32x16 &acc, const bf16x8 &w, const bf16x8 &f) {
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(f));
}
// (A operand in a VGPR: for the few fragments of the in-scan input projection, which hipcc keeps in VGPRs anyway)
__device__ __forceinline__ void mfma_acc_v(f32x16 &acc, const bf16x8 &w, const bf16x8 &f) {
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(f));
}
__device__ __forceinline__ void mfma_new_a(f32x16 &acc, const bf16x8 &w, const bf16x8 &f) {       // acc = w . f
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(w), "v"(f));     // (early clobber: the result must not share registers with the operands)
}
// (the accumulators are operands of the statement: a register read of one cannot be scheduled above it)
__device__ __forceinline__ void mfma_settle(f32x16 &a0) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0)); }
__device__ __forceinline__ void mfma_settle(f32x16 &a0, f32x16 &a1) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1)); }
__device__ __forceinline__ void mfma_settle(f32x16 &a0, f32x16 &a1, f32x16 &a2) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1), "+a"(a2)); }
// (an accumulator whose last MFMA was issued at least one other MFMA ago: only the ordering of the reads is left to state)
__device__ __forceinline__ void mfma_fence(f32x16 &a0) { asm volatile("s_nop 7" : "+a"(a0)); }
__device__ __forceinline__ void mfma_lead_in() { asm volatile("s_nop 7" ::: "memory"); }             // VALU-written accumulator -> first MFMA

// Software-pipelined form for a whole step: NS operand streams (weight rows w[st], LDS tile tile[st]) feeding accumulators
// chosen by acc_of(st, nt).  DEPTH fragments are in flight: the read of fragment i + DEPTH is issued right behind MFMA i, so
// after the first DEPTH reads no MFMA waits for the LDS (reading a batch of 8 fragments and then issuing 8 MFMAs exposes the LDS latency once per batch: 48 MFMAs took
// 4 400 cycles of a 9 600-cycle step where the matrix pipe needs 1 540).  One ds_read_b128 per MFMA gap is free (guide, LDS).
template <int NT, int NK, int LD, int NS, int DEPTH, typename AccOf>
__device__ __forceinline__ void mfma_pipe(const bf16x8 *const (&w)[NS], const bf16_t *const (&tile)[NS], const int col, const int hh, AccOf acc_of) {
    constexpr int N = NS * NK * NT, D = DEPTH < N ? DEPTH : N;
    bf16x8 f[D];
    auto rd = [&](const int i) {
        const int st = i / (NK * NT), k = (i / NT) % NK, nt = i % NT;
        return *reinterpret_cast<const bf16x8 *>(tile[st] + (32 * nt + col) * LD + 16 * k + 8 * hh);
    };
#pragma unroll
    for (int i = 0; i < D; ++i) f[i] = rd(i);
    mfma_lead_in();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int st = i / (NK * NT), k = (i / NT) % NK, nt = i % NT;
        mfma_acc_a(acc_of(st, nt), w[st][k], f[i % D]);
        if (i + D < N) f[i % D] = rd(i + D);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mfma_settle(acc_of(st, nt));   // (an accumulator fed by two streams is named twice: harmless)
}

// true when one of the 8 bf16 values of a weight fragment is Inf / NaN (exponent all ones).  Prologue only: the forward scans skip the
// recurrent product of the first step (h_{-1} = 0), where torch's nn.LSTM forms W_hh . 0 and gets NaN from a non-finite weight.
__device__ __forceinline__ bool frag_nonfinite(const bf16x8 &w) {
    const u32x4 d = __builtin_bit_cast(u32x4, w);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) bad = bad || (d[i] & 0x7F80u) == 0x7F80u || (d[i] & 0x7F800000u) == 0x7F800000u;
    return bad;
}

// ---- exchange rings -------------------------------------------------------------------------------------------------------
// The workgroups of a group do not exchange through the saved sequences (hs[t], da[t]): there an 8- or 32-byte piece of a
// 128-byte line comes from each of up to 8 producer waves, and lines assembled from partial writes are merged beyond the L2
// -- every gather load then waited ~5 000 cycles (in-kernel stamps).  They exchange through small rings (two slots, step
// parity) in which every producer wave owns whole lines and writes them with ONE store instruction each:
//   h  block of a batch tile, MG*H bf16:   [gate tile gt = 4p + wave][nt][trial 32][8 units]           lane (trial, hh): 8 bytes at hh*8
//   (the backward scans exchange partial sums of dh instead of da: their ring is described in nsd_scan2.hip, "backward")
// Two slots are enough: a member publishes step s only after it has gathered step s-1 from every member, i.e. after every
// member has finished reading slot (s & 1) for step s-2.  The row-major tensors are still written (plain stores, behind the
// flag) for the GEMMs and the head that read them after the scan.
__device__ __forceinline__ long ring_h_off(const int gt, const int nt, const int NT, const int col, const int hh) {
    return ((long)(gt * NT + nt) * 32 + col) * 8 + 4 * hh;
}
// ---- saved activations ----------------------------------------------------------------------------------------------------
// cs / ga are private to a forward / backward pair of scans (lane (trial, hh) of wave w of member p owns the same 4 units in
// both), so they are stored per owner instead of row-major: block (32-trial tile, member, step, wave) =
//   gates [half 2][lane 64][16 B] = 2 KB (half 0: the lane's units 0,1 x 4 gates), cell state [lane 64][8 B] = 512 B.
// Every store / load instruction then moves one contiguous KB.  (Row-major, a lane's 32 bytes sat in a row of their own: 64
// line requests per instruction, ~2 400 per CU and step in the backward scan -- ISSUING the step's saved-activation loads took
// ~2 500 cycles wherever they were placed.)
__device__ __forceinline__ long saved_block(const int tile32, const int P, const int p, const int T, const int t, const int wave) {
    return (((long)tile32 * P + p) * T + t) * 4 + wave;
}
__device__ __forceinline__ long saved_ga(const long block, const int half, const int lane) { return block * 1024 + half * 512 + lane * 8; }   // bf16 elements
__device__ __forceinline__ long saved_cs(const long block, const int lane) { return block * 256 + lane * 4; }

// ---- shared pieces of the backward scans (reduce-scatter of partial sums: see nsd_scan2.hip) ------------------------------------
__device__ __forceinline__ u32x2 ld_sc1_b64(const nsd_rsrc r, const unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void st_ring_b128(const bool same_l2, const nsd_rsrc rs, const unsigned off, const u32x4 v) {
    if (same_l2) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 0); else st_sc1_b128(rs, off, v);
}
__device__ __forceinline__ void st_ring_b64(const bool same_l2, const nsd_rsrc rs, const unsigned off, const u32x2 v) {
    if (same_l2) __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)off, 0, 0); else st_sc1_b64(rs, off, v);
}

// What of a cell's backward does not depend on dh: computed from the saved activations BEFORE the wave polls for its partials
struct CellFac { float A[4], Fi[4], Ff[4], Fg[4], Fo[4], fg[4]; };
__device__ __forceinline__ void cell_factors(const u32x4 gq0, const u32x4 gq1, const u32x2 cq, const u32x2 cpq, CellFac &f) {
    const float cv[4] = {bf16_lo(cq[0]), bf16_hi(cq[0]), bf16_lo(cq[1]), bf16_hi(cq[1])};
    const float cp[4] = {bf16_lo(cpq[0]), bf16_hi(cpq[0]), bf16_lo(cpq[1]), bf16_hi(cpq[1])};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned w0 = j < 2 ? gq0[2 * (j & 1)] : gq1[2 * (j & 1)], w1 = j < 2 ? gq0[2 * (j & 1) + 1] : gq1[2 * (j & 1) + 1];
        const float ig = fabsf(bf16_lo(w0)), fg = bf16_hi(w0), gg = bf16_lo(w1), og = bf16_hi(w1);   // (the sign of a saved i is not part of the gate: saved_keep_bits)
        const float tc = fast_tanh(cv[j]);
        f.A[j] = og * (1.f - tc * tc);                 // d c_t / d h_t (through tanh(c_t))
        f.Fi[j] = gg * ig * (1.f - ig);
        f.Ff[j] = cp[j] * fg * (1.f - fg);
        f.Fg[j] = ig * (1.f - gg * gg);
        f.Fo[j] = tc * og * (1.f - og);
        f.fg[j] = fg;
    }
}
// The saved input gate i = sigmoid(.) > 0 has a free sign bit.  The fused forward scan stores there whether the unit's OUTPUT survived the
// dropout between the layers (set = kept), so the backward scan needs no random stream: multiplier = bit ? keep : 0.
__device__ __forceinline__ void saved_keep_bits(const u32x4 gq0, const u32x4 gq1, const float keep, float (&m)[4]) {
    m[0] = (gq0[0] & 0x8000u) ? keep : 0.f; m[1] = (gq0[2] & 0x8000u) ? keep : 0.f;
    m[2] = (gq1[0] & 0x8000u) ? keep : 0.f; m[3] = (gq1[2] & 0x8000u) ? keep : 0.f;
}
// The factors are computed AHEAD of the exchange on purpose.  Where their only use sits inside a conditional block behind the
// poll, the optimizer sinks the whole computation (tanh, the products, the dropout stream) into that block -- onto the critical
// path (1 100 of the fused backward's 11 100 cycles).  An empty asm statement that "modifies" a value pins it where it is.
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(unsigned &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(CellFac &f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { pin(f.A[j]); pin(f.Fi[j]); pin(f.Ff[j]); pin(f.Fg[j]); pin(f.Fo[j]); pin(f.fg[j]); }
}
// ... and what does: da (16 values, unit-major, packed), the carried dc, the bias-gradient sums
__device__ __forceinline__ void cell_apply(const CellFac &f, const float (&dh)[4], float (&dc)[4], float (&dbs)[16], unsigned (&dw)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dct = fmaf(dh[j], f.A[j], dc[j]);
        dc[j] = dct * f.fg[j];
        const float dai = dct * f.Fi[j], daf = dct * f.Ff[j], dag = dct * f.Fg[j], dao = dh[j] * f.Fo[j];
        dw[2 * j] = pack_bf16x2(dai, daf);
        dw[2 * j + 1] = pack_bf16x2(dag, dao);
        dbs[4 * j] += dai; dbs[4 * j + 1] += daf; dbs[4 * j + 2] += dag; dbs[4 * j + 3] += dao;
    }
}

// dropout multipliers of 4 adjacent units of one (layer, trial, step): the product's counter stream (nsd_rand_u32, index
// ((layer * B + b) * T + t) * ld + column); all 1 for padding trials or when the stream is off
__device__ __forceinline__ void drop_mult4(const RngArgs &rng, const bool on, const int layer, const int B, const int T, const int b,
                                           const int t, const long ld, const int col0, float (&m)[4]) {
    m[0] = m[1] = m[2] = m[3] = 1.f;
    if (on && b < B) {
        const uint64_t base = (((uint64_t)layer * B + b) * T + t) * (uint64_t)ld + (uint64_t)col0;
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = nsd_rand_u32(rng.seed, rng.base, base + j) >= rng.thr_lstm ? rng.keep_lstm : 0.f;
    }
}

// one multiplier, branch-free (for use between MFMAs); the same stream as drop_mult4
__device__ __forceinline__ float drop_mult1(const RngArgs &rng, const bool on, const uint64_t index) {
    const float kept = nsd_rand_u32(rng.seed, rng.base, index) >= rng.thr_lstm ? rng.keep_lstm : 0.f;
    return on ? kept : 1.f;
}

}  // namespace
