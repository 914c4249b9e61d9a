// nsd_lstm_batched.hip -- stacked LSTM for LARGE hidden sizes (H a multiple of 16, e.g. BASELINE cfg3: H=256, B=1024):
// the per-time-step gate computation of a whole batch is a real contraction
//     gates[B, 4H] = [in_t | h_{t-1}] [B, I+H] . [W_ih | W_hh]^T [I+H, 4H]
// and runs on the matrix pipe (v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate), one launch per (layer, step) over
// the whole GPU, with the LSTM cell fused into the GEMM epilogue.  Same semantics as nsd_lstm2*.hip / nsd_lstm_generic.hip:
// self.lstm(x) of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34 (torch.nn.LSTM: gate order i,f,g,o, two biases,
// zero initial state, dropout multipliers between layers) and autograd through it.
//
//   forward   lstm_step_fwd_mfma   64 trials x 16 units (x 4 gates) per workgroup; operands staged through LDS in
//                                  32-wide k chunks (coalesced 16-byte loads); the 4 gates of a unit sit in the 4 lanes
//                                  of a quad of the accumulator layout, so sigma/tanh, the cell update and the saves
//                                  (h, c, activated gates, linked output) happen in registers.
//   backward  lstm_cell_bwd        element-wise: da_t from dh_t, dc_{t+1}, the saved gates and cell states
//             lstm_step_bwd_mfma   [d in_t | dh_{t-1}] = da_t [B,4H] . [W_ih | W_hh] [4H, I+H]
//             gemm_tn_mfma         dW = da_seq^T . operand_seq over all (b, t): split-K over 4 workgroup rows,
//                                  partials summed in a fixed order (deterministic); column sums for the biases.
// The per-trial kernels of nsd_lstm_generic.hip remain for H not a multiple of 16 and for tiny batches.
#include <string.h>
#include "nsd_args.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NSD_DW_SPLITS
#define NSD_DW_SPLITS 8         // split-K parts of the weight-gradient GEMMs (workspace sized for 8)
#endif
constexpr int TM = 64;            // trials (or GEMM rows) per workgroup tile
constexpr int TN = 64;            // gate columns (or GEMM columns) per workgroup tile
#ifndef NSD_KC
#define NSD_KC 64
#endif
constexpr int KC = NSD_KC;        // k chunk staged in LDS (32 or 64)
constexpr int LD = KC + 1;        // LDS row stride (odd: fragment reads are conflict-free)

// One k chunk of the 64x64 tile product on the 2x2 wave grid: wave (wm, wn) owns rows 32*wm.., columns 32*wn...
// As[m][k], Bs[n][k]; 32x32x2 fragments: lane (i = lane & 31, kq = lane >> 5) feeds A[i][k + kq] and B[k + kq][i].
// The fragments of the whole chunk are read into registers first: the MFMAs then issue back to back instead of each
// waiting for its own two LDS reads.
template <int KW>
__device__ __forceinline__ void tile_mfma_full(const float (*As)[LD], const float (*Bs)[LD], const int wm, const int wn,
                                               const int lane, f32x16 &acc) {
    const int i = lane & 31, kq = lane >> 5;
    float av[KW / 2], bv[KW / 2];
#pragma unroll
    for (int k = 0; k < KW / 2; ++k) { av[k] = As[32 * wm + i][2 * k + kq]; bv[k] = Bs[32 * wn + i][2 * k + kq]; }
#pragma unroll
    for (int k = 0; k < KW / 2; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[k], acc, 0, 0, 0);
}
__device__ __forceinline__ void tile_mfma(const float (*As)[LD], const float (*Bs)[LD], const int kw, const int wm,
                                          const int wn, const int lane, f32x16 &acc) {
    if (kw == KC) { tile_mfma_full<KC>(As, Bs, wm, wn, lane, acc); return; }
    const int i = lane & 31, kq = lane >> 5;
    for (int k = 0; k < kw; k += 2) {
        const float av = As[32 * wm + i][k + kq];
        const float bv = Bs[32 * wn + i][k + kq];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
}
// Staging of a [64 rows x KC k] chunk whose rows are k-contiguous in memory: thread (row = tid >> 2, part = tid & 3) moves
// KC/16 16-byte pieces; the loads of the NEXT chunk are issued before the MFMAs of the current one (register prefetch).
constexpr int CV = KC / 16;                                       // float4 per thread and operand
struct Chunk { float4 v[CV]; };
__device__ __forceinline__ Chunk load_rows_k(const float *rowp, const bool ok) {      // rowp: this thread's row at k0 + (KC/4)*part
    Chunk c;
#pragma unroll
    for (int q = 0; q < CV; ++q) c.v[q] = ok ? *reinterpret_cast<const float4 *>(rowp + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    return c;
}
__device__ __forceinline__ void store_rows_k(float (*S)[LD], const int row, const int part, const Chunk &c) {
    float *d = &S[row][(KC / 4) * part];
#pragma unroll
    for (int q = 0; q < CV; ++q) { d[4 * q] = c.v[q].x; d[4 * q + 1] = c.v[q].y; d[4 * q + 2] = c.v[q].z; d[4 * q + 3] = c.v[q].w; }
}
// Staging of a [KC k x 64 n] chunk whose k rows are n-contiguous in memory (weights of the backward step, operands of the
// weight-gradient GEMM): thread (k = tid >> 3, part = tid & 7) moves two 16-byte pieces of rows k, k + 32, ... -> S[n][k]
struct ChunkN { float4 v[KC / 32][2]; };
// p: row k of the chunk at n0 + 8*part; row_stride: floats between k rows; ok_row(j): row k + 32 j is valid
template <class F>
__device__ __forceinline__ ChunkN load_rows_n(const float *p, const long row_stride, const bool ok0, const bool ok1, F ok_row) {
    ChunkN c;
#pragma unroll
    for (int j = 0; j < KC / 32; ++j) {
        const bool v = ok_row(j);
        const float *q = p + (v ? (long)(32 * j) * row_stride : 0);
        c.v[j][0] = (v && ok0) ? *reinterpret_cast<const float4 *>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
        c.v[j][1] = (v && ok1) ? *reinterpret_cast<const float4 *>(q + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return c;
}
__device__ __forceinline__ void store_rows_n(float (*S)[LD], const int k, const int part, const ChunkN &c) {
    const int n = 8 * part;
#pragma unroll
    for (int j = 0; j < KC / 32; ++j) {
        const int kk = k + 32 * j;
        S[n][kk] = c.v[j][0].x; S[n + 1][kk] = c.v[j][0].y; S[n + 2][kk] = c.v[j][0].z; S[n + 3][kk] = c.v[j][0].w;
        S[n + 4][kk] = c.v[j][1].x; S[n + 5][kk] = c.v[j][1].y; S[n + 6][kk] = c.v[j][1].z; S[n + 7][kk] = c.v[j][1].w;
    }
}
// ---- optional bf16 operands (NSD_FLAG_BF16: GEMM inputs rounded to bf16 at staging, fp32 accumulate, everything else
// fp32): v_mfma_f32_32x32x16_bf16, lane (i = lane & 31, kq = lane >> 5) feeds 8 consecutive k: A[i][16 ks + 8 kq ..].
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDH = KC + 8;                                       // halfs per LDS row (16-byte aligned rows, conflict-free b128 reads)
__device__ __forceinline__ void tile_mfma_bf16(const __bf16 (*Ah)[LDH], const __bf16 (*Bh)[LDH], const int wm, const int wn,
                                               const int lane, f32x16 &acc) {
    const int i = lane & 31, kq = lane >> 5;
    bf16x8 av[KC / 16], bv[KC / 16];
#pragma unroll
    for (int k = 0; k < KC / 16; ++k) {
        av[k] = *reinterpret_cast<const bf16x8 *>(&Ah[32 * wm + i][16 * k + 8 * kq]);
        bv[k] = *reinterpret_cast<const bf16x8 *>(&Bh[32 * wn + i][16 * k + 8 * kq]);
    }
#pragma unroll
    for (int k = 0; k < KC / 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[k], bv[k], acc, 0, 0, 0);
}
__device__ __forceinline__ void store_rows_k_h(__bf16 (*S)[LDH], const int row, const int part, const Chunk &c) {
    __bf16 *d = &S[row][(KC / 4) * part];
#pragma unroll
    for (int q = 0; q < CV; ++q) {
        d[4 * q] = (__bf16)c.v[q].x; d[4 * q + 1] = (__bf16)c.v[q].y; d[4 * q + 2] = (__bf16)c.v[q].z; d[4 * q + 3] = (__bf16)c.v[q].w;
    }
}
__device__ __forceinline__ void store_rows_n_h(__bf16 (*S)[LDH], const int k, const int part, const ChunkN &c) {
    const int n = 8 * part;
#pragma unroll
    for (int j = 0; j < KC / 32; ++j) {
        const int kk = k + 32 * j;
        S[n][kk] = (__bf16)c.v[j][0].x; S[n + 1][kk] = (__bf16)c.v[j][0].y; S[n + 2][kk] = (__bf16)c.v[j][0].z; S[n + 3][kk] = (__bf16)c.v[j][0].w;
        S[n + 4][kk] = (__bf16)c.v[j][1].x; S[n + 5][kk] = (__bf16)c.v[j][1].y; S[n + 6][kk] = (__bf16)c.v[j][1].z; S[n + 7][kk] = (__bf16)c.v[j][1].w;
    }
}
// LDS of a GEMM kernel: the fp32 tiles, or (same bytes, reinterpreted) the bf16 tiles
struct TileMem {
    float As[TM][LD], Bs[TN][LD];
};
#define TILE_H(mem, Ah, Bh) \
    __bf16 (*Ah)[LDH] = reinterpret_cast<__bf16 (*)[LDH]>(&(mem).As[0][0]); \
    __bf16 (*Bh)[LDH] = reinterpret_cast<__bf16 (*)[LDH]>(&(mem).Bs[0][0])

// accumulator register r of lane l holds D[row][col] with col = l & 31, row = 8 * (r / 4) + 4 * (l >> 5) + (r % 4)
__device__ __forceinline__ int acc_row(const int r, const int lane) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }

// ---------------------------------------------------------------------------------------------------------------
// forward step
// ---------------------------------------------------------------------------------------------------------------
struct StepFwdArgs {
    const float *in;          // [B,T,I] layer input
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    const float *mask;        // [B,T,H] multipliers on this layer's output (null: none)
    const float *res_in;      // [B,T,H] residual input to add (null: none)
    float *hseq, *cseq, *gact;   // [B,T,H], [B,T,H], [B,T,H,4]   (null in inference)
    float *out;               // [B,T,H] linked output = (h + res) * mask
    const float *hprev;       // sequence h_{t-1} is read from: hseq (training) or out (inference: out == h there)
    int B, T, I, H;
};
// One launch advances EVERY layer by one step, layer l working on t = s - l (the layers are skewed by one step: layer l
// reads what layer l-1 wrote in the previous launch).  Half the launches of a layer-by-layer schedule for L = 2, and two
// workgroups per CU in flight instead of one.
struct StepFwdAll {
    StepFwdArgs lay[NSD_MAX_LAYERS];
    const float *c_base;      // inference: cell-state ping-pong, [L][2][B,H]
    int s;
};

template <bool BF>
__global__ __launch_bounds__(256) void lstm_step_fwd_mfma(StepFwdAll all) {
    __shared__ __align__(16) TileMem mem;
    float (*As)[LD] = mem.As; float (*Bs)[LD] = mem.Bs;
    TILE_H(mem, Ah, Bh);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const StepFwdArgs &a = all.lay[blockIdx.z];
    const int t = all.s - (int)blockIdx.z;
    if (t < 0 || t >= a.T) return;
    const int T = a.T, I = a.I, H = a.H;
    const int b0 = blockIdx.x * TM, u0 = blockIdx.y * 16;        // 16 units = 64 gate columns, column c = 4 * unit + gate
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // the cell's own operands do not depend on the GEMM: fetch them now, their latency runs under the k loop
    const int cc = 32 * wn + (lane & 31), g = cc & 3, u = u0 + (cc >> 2);
    const bool uok = u < H;
    const float bias = uok ? a.b_ih[g * H + u] + a.b_hh[g * H + u] : 0.f;
    float *cst = const_cast<float *>(all.c_base) + (size_t)blockIdx.z * 2 * a.B * H;     // inference: cell-state ping-pong
    const float *c_in = cst + (size_t)(t & 1) * a.B * H;
    float *c_out = cst + (size_t)((t + 1) & 1) * a.B * H;
    float cp[16], rs[16], mk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = b0 + 32 * wm + acc_row(r, lane);
        const size_t row = (size_t)(b < a.B ? b : 0) * T + t;
        const size_t bh = (size_t)(b < a.B ? b : 0) * H + (uok ? u : 0);
        const size_t ru = row * H + (uok ? u : 0);
        cp[r] = t == 0 ? 0.f : (a.cseq ? a.cseq[ru - H] : c_in[bh]);
        rs[r] = a.res_in ? a.res_in[ru] : 0.f;
        mk[r] = a.mask ? a.mask[ru] : 1.f;
    }

    // two k segments: the layer input (I wide) and the recurrent state h_{t-1} (H wide; zero at t = 0)
    for (int seg = 0; seg < 2; ++seg) {
        const int K = seg == 0 ? I : H;
        if (seg == 1 && t == 0) break;
        const float *src = seg == 0 ? a.in : a.hprev;
        const int tt = seg == 0 ? t : t - 1;
        const float *w = seg == 0 ? a.w_ih : a.w_hh;
        if ((K & (KC - 1)) == 0) {
            // fast path: whole 32-wide chunks, 16-byte loads, next chunk's loads in flight during the MFMAs
            const int row = tid >> 2, part = tid & 3;
            const int b = b0 + row;
            const int wrow = (row & 3) * H + u0 + (row >> 2);
            const bool aok = b < a.B, bok = u0 + (row >> 2) < H;
            const float *ap = src + ((size_t)(aok ? b : 0) * T + tt) * K + (KC / 4) * part;
            const float *bp = w + (size_t)(bok ? wrow : 0) * K + (KC / 4) * part;
            Chunk ca = load_rows_k(ap, aok), cb = load_rows_k(bp, bok);
            for (int k0 = 0; k0 < K; k0 += KC) {
                __syncthreads();
                if (BF) { store_rows_k_h(Ah, row, part, ca); store_rows_k_h(Bh, row, part, cb); }
                else    { store_rows_k(As, row, part, ca); store_rows_k(Bs, row, part, cb); }
                if (k0 + KC < K) { ca = load_rows_k(ap + k0 + KC, aok); cb = load_rows_k(bp + k0 + KC, bok); }
                __syncthreads();
                if (BF) tile_mfma_bf16(Ah, Bh, wm, wn, lane, acc);
                else    tile_mfma_full<KC>(As, Bs, wm, wn, lane, acc);
            }
        } else {
            for (int k0 = 0; k0 < K; k0 += KC) {
                const int kw = (K - k0 < KC) ? K - k0 : KC;
                const int kwp = (kw + 1) & ~1;                       // MFMA consumes k in pairs: pad with zeros
                __syncthreads();
                for (int e = tid; e < TM * kwp; e += 256) {
                    const int m = e / kwp, kk = e - m * kwp;
                    const int b = b0 + m;
                    As[m][kk] = (b < a.B && kk < kw) ? src[((size_t)b * T + tt) * K + k0 + kk] : 0.f;
                    const int c = m, row = (c & 3) * H + u0 + (c >> 2);          // Bs row = gate column c of this tile
                    Bs[c][kk] = (u0 + (c >> 2) < H && kk < kw) ? w[(size_t)row * K + k0 + kk] : 0.f;
                }
                __syncthreads();
                tile_mfma(As, Bs, kwp, wm, wn, lane, acc);
            }
        }
    }

    // ---- fused LSTM cell: lane column cc = 32*wn + (lane & 31) -> unit u, gate g (the quad holds i,f,g,o); the operands
    // of the cell were fetched at the top of the kernel
    if (u >= H) return;                                           // (whole quads leave together)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = b0 + 32 * wm + acc_row(r, lane);
        const float pre = acc[r] + bias;
        const float act = g == 2 ? fast_tanh(pre) : fast_sigmoid(pre);
        const float ig = quad_bcast<0>(act), fg = quad_bcast<1>(act), gg = quad_bcast<2>(act), og = quad_bcast<3>(act);
        const float cn = fmaf(fg, cp[r], ig * gg);
        const float h = og * fast_tanh(cn);
        if (b < a.B) {
            const size_t row = (size_t)b * T + t;
            if (a.gact) a.gact[(row * H + u) * 4 + g] = act;
            if (g == 0) {
                if (a.cseq) { a.cseq[row * H + u] = cn; a.hseq[row * H + u] = h; }
                else        c_out[(size_t)b * H + u] = cn;
                a.out[row * H + u] = (h + rs[r]) * mk[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward step
// ---------------------------------------------------------------------------------------------------------------
struct CellBwdArgs {
    const float *gact, *cseq;
    const float *dsrc;                // [B,T,H] gradient w.r.t. this layer's linked output from the layer above (null: top layer)
    const float *mask;                // [B,T,H] multipliers that were applied to this layer's output (null: none)
    const float *alpha, *dscore, *dpooled, *attn_w;   // top layer: d out_t = alpha_t * dpooled + dscore_t * attn_w
    const float *dhrec;               // [B,H] W_hh^T da_{t+1} (zeros at t = T-1)
    float *dc;                        // [B,H] dc_{t+1} * f_{t+1} in, dc_t * f_t out
    float *dho;                       // [B,H] d linked output at this step (residual pass-through), may be null
    float *da_seq;                    // [B,T,4H]
    int B, T, H;
};
// The backward launches advance every layer by one step as well: layer l works on t = T-1 - (s - (L-1-l)), one step
// behind the layer above it, whose d(input) it reads from the previous launch.
struct CellBwdAll { CellBwdArgs lay[NSD_MAX_LAYERS]; int L, s; };

__global__ __launch_bounds__(256) void lstm_cell_bwd(CellBwdAll all) {
    const CellBwdArgs &a = all.lay[blockIdx.y];
    const int H = a.H, T = a.T;
    const int t = T - 1 - (all.s - (all.L - 1 - (int)blockIdx.y));
    if (t < 0 || t >= T) return;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.B * H) return;
    const int b = (int)(idx / H), j = (int)(idx - (long)b * H);
    const size_t row = (size_t)b * T + t;
    float dout;
    if (a.dsrc) dout = a.dsrc[row * H + j] * (a.mask ? a.mask[row * H + j] : 1.f);
    else        dout = fmaf(a.alpha[row], a.dpooled[(size_t)b * H + j], a.dscore[row] * a.attn_w[j]);
    if (a.dho) a.dho[idx] = dout;
    const float4 g = *reinterpret_cast<const float4 *>(a.gact + (row * H + j) * 4);
    const float ig = g.x, fg = g.y, gg = g.z, og = g.w;
    const float ct = a.cseq[row * H + j];
    const float cp = t > 0 ? a.cseq[(row - 1) * H + j] : 0.f;
    const float tc = fast_tanh(ct);
    const float dht = dout + (t < T - 1 ? a.dhrec[idx] : 0.f);
    const float dct = fmaf(dht * og, 1.f - tc * tc, t < T - 1 ? a.dc[idx] : 0.f);
    a.dc[idx] = dct * fg;
    float *dg = a.da_seq + row * 4 * H;
    dg[j] = dct * gg * ig * (1.f - ig);
    dg[H + j] = dct * cp * fg * (1.f - fg);
    dg[2 * H + j] = dct * ig * (1.f - gg * gg);
    dg[3 * H + j] = dht * tc * og * (1.f - og);
}

// out[b][n] = sum_r da_t[b][r] * W[r][n], n over the recurrent columns (-> dhrec [B,H]) and, for layers > 0, the input
// columns (-> din_seq[b,t,:] [+ dho]); blockIdx.y walks 64-column tiles: first ceil(H/64) tiles W_hh, then W_ih.
struct StepBwdArgs {
    const float *da_seq;              // [B,T,4H]
    const float *w_ih, *w_hh;
    const float *dho;                 // residual pass-through (null: none)
    float *dhrec;                     // [B,H]
    float *din_seq;                   // [B,T,I] (null: not needed)
    int B, T, I, H, n_hh_tiles, n_tiles;
};
struct StepBwdAll { StepBwdArgs lay[NSD_MAX_LAYERS]; int L, s; };

template <bool BF>
__global__ __launch_bounds__(256) void lstm_step_bwd_mfma(StepBwdAll all) {
    __shared__ __align__(16) TileMem mem;
    float (*As)[LD] = mem.As; float (*Bs)[LD] = mem.Bs;
    TILE_H(mem, Ah, Bh);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const StepBwdArgs &a = all.lay[blockIdx.z];
    const int T = a.T, H = a.H, K = 4 * H;
    const int t = T - 1 - (all.s - (all.L - 1 - (int)blockIdx.z));
    if (t < 0 || t >= T || (int)blockIdx.y >= a.n_tiles) return;
    if (t == 0 && (int)blockIdx.y < a.n_hh_tiles) return;          // dh_{-1} is not needed
    const bool hh = (int)blockIdx.y < a.n_hh_tiles;
    const int N = hh ? H : a.I;
    const float *w = hh ? a.w_hh : a.w_ih;
    const int b0 = blockIdx.x * TM, n0 = (hh ? blockIdx.y : blockIdx.y - a.n_hh_tiles) * TN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {   // K = 4H is a multiple of 64; N (H or I) is a multiple of 4 on this path
        const int row = tid >> 2, part = tid & 3;                 // A: da_t rows, k-contiguous
        const int b = b0 + row;
        const bool aok = b < a.B;
        const float *ap = a.da_seq + ((size_t)(aok ? b : 0) * T + t) * K + (KC / 4) * part;
        const int kk = tid >> 3, np = tid & 7;                     // B: W rows k, n-contiguous
        const bool bok0 = n0 + 8 * np < N, bok1 = n0 + 8 * np + 4 < N;
        const float *bp = w + (size_t)kk * N + n0 + 8 * np;
        auto allk = [](int) { return true; };
        Chunk ca = load_rows_k(ap, aok);
        ChunkN cb = load_rows_n(bp, N, bok0, bok1, allk);
        for (int k0 = 0; k0 < K; k0 += KC) {
            __syncthreads();
            if (BF) { store_rows_k_h(Ah, row, part, ca); store_rows_n_h(Bh, kk, np, cb); }
            else    { store_rows_k(As, row, part, ca); store_rows_n(Bs, kk, np, cb); }
            if (k0 + KC < K) { ca = load_rows_k(ap + k0 + KC, aok); cb = load_rows_n(bp + (size_t)(k0 + KC) * N, N, bok0, bok1, allk); }
            __syncthreads();
            if (BF) tile_mfma_bf16(Ah, Bh, wm, wn, lane, acc);
            else    tile_mfma_full<KC>(As, Bs, wm, wn, lane, acc);
        }
    }
    const int n = n0 + 32 * wn + (lane & 31);
    if (n >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = b0 + 32 * wm + acc_row(r, lane);
        if (b >= a.B) continue;
        if (hh) a.dhrec[(size_t)b * H + n] = acc[r];
        else {
            float v = acc[r];
            if (a.dho) v += a.dho[(size_t)b * H + n];              // residual_add needs I == H
            a.din_seq[((size_t)b * T + t) * N + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradients: C[M,N] = sum_rows A[row][m] * Bm[src(row)][n], rows = B*T split over gridDim.z parts
// ---------------------------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void gemm_tn_mfma(const float *A, int lda, const float *Bm, int ldb, float *part, int M, int N,
                                                    long rows, int shiftT) {
    __shared__ __align__(16) TileMem mem;                         // As[m][k], Bs[n][k]
    float (*As)[LD] = mem.As; float (*Bs)[LD] = mem.Bs;
    TILE_H(mem, Ah, Bh);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const long per = ((rows + gridDim.z - 1) / gridDim.z + KC - 1) / KC * KC;
    const long r_lo = (long)blockIdx.z * per, r_hi = (r_lo + per < rows) ? r_lo + per : rows;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {   // M = 4H and N (I or H) are multiples of 4 on this path; both operands are contiguous along m / n
        const int kk = tid >> 3, cp = tid & 7;
        const bool aok0 = m0 + 8 * cp < M, aok1 = m0 + 8 * cp + 4 < M;
        const bool bok0 = n0 + 8 * cp < N, bok1 = n0 + 8 * cp + 4 < N;
        auto lda_ = [&](long r0) {
            return load_rows_n(A + (r0 + kk) * lda + m0 + 8 * cp, lda, aok0, aok1, [&](int j) { return r0 + kk + 32 * j < r_hi; });
        };
        auto ldb_ = [&](long r0) {
            const long sh = shiftT != 0 ? 1 : 0;                  // operand row of (b, t) is (b, t-1); zero for t == 0
            return load_rows_n(Bm + (r0 + kk - sh) * ldb + n0 + 8 * cp, ldb, bok0, bok1, [&](int j) {
                const long row = r0 + kk + 32 * j;
                return row < r_hi && (shiftT == 0 || row % shiftT != 0);
            });
        };
        ChunkN ca = lda_(r_lo), cb = ldb_(r_lo);
        for (long r0 = r_lo; r0 < r_hi; r0 += KC) {
            __syncthreads();
            if (BF) { store_rows_n_h(Ah, kk, cp, ca); store_rows_n_h(Bh, kk, cp, cb); }
            else    { store_rows_n(As, kk, cp, ca); store_rows_n(Bs, kk, cp, cb); }
            if (r0 + KC < r_hi) { ca = lda_(r0 + KC); cb = ldb_(r0 + KC); }
            __syncthreads();
            if (BF) tile_mfma_bf16(Ah, Bh, wm, wn, lane, acc);
            else    tile_mfma_full<KC>(As, Bs, wm, wn, lane, acc);
        }
    }
    float *dst = part + (size_t)blockIdx.z * M * N;
    const int n = n0 + 32 * wn + (lane & 31);
    if (n >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wm + acc_row(r, lane);
        if (m < M) dst[(size_t)m * N + n] = acc[r];
    }
}
__global__ void sum_parts_kernel(const float *part, int nparts, long n, float *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out[i] = s;
}
// bias gradients: part[z][c] = sum over the rows of split z of A[row][c]  (256 threads = 64 columns x 4 row lanes)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float *A, int lda, int M, long rows, float *part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r_lo = (long)blockIdx.y * per, r_hi = (r_lo + per < rows) ? r_lo + per : rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < M) {
        long r = r_lo + g;
        for (; r + 12 < r_hi; r += 16) { s0 += A[r * lda + c]; s1 += A[(r + 4) * lda + c]; s2 += A[(r + 8) * lda + c]; s3 += A[(r + 12) * lda + c]; }
        for (; r < r_hi; r += 4) s0 += A[r * lda + c];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < M)
        part[(size_t)blockIdx.y * M + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void sum_parts2_kernel(const float *part, int nparts, long n, float *out1, float *out2) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out1[i] = s; out2[i] = s;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host drivers (same contracts as nsd_lstm_generic_fwd / _bwd; training only: the forward needs hseq / cseq)
// ---------------------------------------------------------------------------------------------------------------
bool nsd_lstm_batched_ok(const nsd_dims *d, bool training) {
    // (16-byte operand loads: the channel count must be a multiple of 4 as well)
    (void)training;
    return d->H % 16 == 0 && d->H >= 64 && d->B >= 16 && d->C % 4 == 0;
}

static int launch_fwd_steps(StepFwdAll &all, int B, int T, int H, int L, bool bf16, hipStream_t st) {
    const dim3 grid((B + TM - 1) / TM, (H + 15) / 16, L);
    for (int s = 0; s < T + L - 1; ++s) {
        all.s = s;
        if (bf16) hipLaunchKernelGGL(lstm_step_fwd_mfma<true>, grid, dim3(256), 0, st, all);
        else      hipLaunchKernelGGL(lstm_step_fwd_mfma<false>, grid, dim3(256), 0, st, all);
    }
    NSD_CHECK_LAUNCH("lstm_step_fwd_mfma");
    return NSD_OK;
}

int nsd_lstm_batched_fwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, float *hseq, float *cseq, float *gact, float *inseq, float *top_out, bool bf16, hipStream_t st) {
    const int B = d->B, T = d->T, H = d->H, L = d->L;
    const int64_t BTH = (int64_t)B * T * H;
    StepFwdAll all;
    memset(&all, 0, sizeof(all));
    const float *in = x;
    for (int l = 0; l < L; ++l) {
        StepFwdArgs &a = all.lay[l];
        a.in = in; a.I = l == 0 ? d->C : H;
        a.w_ih = params + pl.w_ih[l]; a.w_hh = params + pl.w_hh[l]; a.b_ih = params + pl.b_ih[l]; a.b_hh = params + pl.b_hh[l];
        a.mask = (l < L - 1 && drop_lstm) ? drop_lstm + (int64_t)l * BTH : nullptr;
        a.res_in = (residual && l >= 1) ? in : nullptr;
        a.hseq = hseq + (int64_t)l * BTH; a.cseq = cseq + (int64_t)l * BTH; a.gact = gact + (int64_t)l * 4 * BTH;
        a.out = (l == L - 1) ? top_out : inseq + (int64_t)l * BTH;
        a.hprev = a.hseq;
        a.B = B; a.T = T; a.H = H;
        in = a.out;
    }
    return launch_fwd_steps(all, B, T, H, L, bf16, st);
}

// inference (no residual): only the linked outputs are produced, ping-ponging between top_out and scratch2 so that the
// last layer lands in top_out (with the one-step skew a layer overwrites a row two launches after its reader is done);
// cstate: [L][2][B,H] cell-state ping-pong
int nsd_lstm_batched_infer(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, float *top_out,
                           float *scratch2, float *cstate, bool bf16, hipStream_t st) {
    const int B = d->B, T = d->T, H = d->H, L = d->L;
    StepFwdAll all;
    memset(&all, 0, sizeof(all));
    all.c_base = cstate;
    const float *in = x;
    for (int l = 0; l < L; ++l) {
        StepFwdArgs &a = all.lay[l];
        a.in = in; a.I = l == 0 ? d->C : H;
        a.w_ih = params + pl.w_ih[l]; a.w_hh = params + pl.w_hh[l]; a.b_ih = params + pl.b_ih[l]; a.b_hh = params + pl.b_hh[l];
        a.out = ((L - 1 - l) & 1) ? scratch2 : top_out;
        a.hprev = a.out;                                     // no residual, no multipliers: the linked output is h itself
        a.B = B; a.T = T; a.H = H;
        in = a.out;
    }
    return launch_fwd_steps(all, B, T, H, L, bf16, st);
}

// scratch: `din_a`, `din_b` [B,T,H] ping-pong for d(layer input) (one writer and one reader each, a step apart);
// `state` >= L*3*B*H + 8 * 4H * max(C,H) floats (per-layer dhrec / dc / dho + the split-K partials); da_seq [L][B,T,4H].
int nsd_lstm_batched_bwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, const float *hseq, const float *cseq, const float *gact, const float *inseq,
                         const float *alpha, const float *dscore, const float *dpooled, float *da_seq, float *din_a, float *din_b,
                         float *state, float *slab, bool bf16, hipStream_t st) {
    const int B = d->B, T = d->T, H = d->H, L = d->L;
    const int64_t BTH = (int64_t)B * T * H;
    const long rows = (long)B * T;
    // state: [L][3][B,H] (dhrec, dc, dho per layer) + split-K partials; da_seq: [L][B,T,4H]
    float *parts = state + (size_t)L * 3 * B * H;
    const int NPART = NSD_DW_SPLITS;
    CellBwdAll call;
    StepBwdAll sall;
    memset(&call, 0, sizeof(call));
    memset(&sall, 0, sizeof(sall));
    call.L = sall.L = L;
    int max_tiles = 0;
    for (int l = L - 1; l >= 0; --l) {
        const int I = l == 0 ? d->C : H;
        float *st_l = state + (size_t)l * 3 * B * H;
        float *da_l = da_seq + (size_t)l * 4 * BTH;
        float *din_l = l > 0 ? ((l & 1) ? din_a : din_b) : nullptr;             // written by layer l, read by layer l-1
        const float *dsrc = l < L - 1 ? (((l + 1) & 1) ? din_a : din_b) : nullptr;
        const bool res_add = residual && l >= 1;
        CellBwdArgs &c = call.lay[l];
        c.gact = gact + (int64_t)l * 4 * BTH; c.cseq = cseq + (int64_t)l * BTH;
        c.dsrc = dsrc;
        c.mask = (l < L - 1 && drop_lstm) ? drop_lstm + (int64_t)l * BTH : nullptr;
        c.alpha = alpha; c.dscore = dscore; c.dpooled = dpooled; c.attn_w = params + pl.attn_w;
        c.dhrec = st_l; c.dc = st_l + (size_t)B * H; c.dho = res_add ? st_l + 2 * (size_t)B * H : nullptr;
        c.da_seq = da_l;
        c.B = B; c.T = T; c.H = H;
        StepBwdArgs &s = sall.lay[l];
        s.da_seq = da_l; s.w_ih = params + pl.w_ih[l]; s.w_hh = params + pl.w_hh[l];
        s.dho = c.dho; s.dhrec = st_l; s.din_seq = din_l;
        s.B = B; s.T = T; s.I = I; s.H = H;
        s.n_hh_tiles = (H + TN - 1) / TN;
        s.n_tiles = s.n_hh_tiles + (l > 0 ? (I + TN - 1) / TN : 0);
        if (s.n_tiles > max_tiles) max_tiles = s.n_tiles;
    }
    const dim3 cgrid((unsigned)(((long)B * H + 255) / 256), L);
    const dim3 sgrid((B + TM - 1) / TM, max_tiles, L);
    for (int s = 0; s < T + L - 1; ++s) {
        call.s = sall.s = s;
        hipLaunchKernelGGL(lstm_cell_bwd, cgrid, dim3(256), 0, st, call);
        if (bf16) hipLaunchKernelGGL(lstm_step_bwd_mfma<true>, sgrid, dim3(256), 0, st, sall);
        else      hipLaunchKernelGGL(lstm_step_bwd_mfma<false>, sgrid, dim3(256), 0, st, sall);
    }
    NSD_CHECK_LAUNCH("lstm_step_bwd_mfma");
    // weight gradients: dW_ih = da^T . in_l ; dW_hh = da^T . h_l[t-1] ; db = column sums of da
    for (int l = 0; l < L; ++l) {
        const int I = l == 0 ? d->C : H;
        const float *da_l = da_seq + (size_t)l * 4 * BTH;
        const float *in_l = l == 0 ? x : inseq + (int64_t)(l - 1) * BTH;
        const int M = 4 * H;
        // (layer 0's input has only C columns: its weight gradient stays fp32)
        if (bf16 && l > 0) hipLaunchKernelGGL(gemm_tn_mfma<true>, dim3((I + TN - 1) / TN, (M + TM - 1) / TM, NPART), dim3(256), 0, st, da_l, M,
                                              in_l, I, parts, M, I, rows, 0);
        else               hipLaunchKernelGGL(gemm_tn_mfma<false>, dim3((I + TN - 1) / TN, (M + TM - 1) / TM, NPART), dim3(256), 0, st, da_l, M,
                                              in_l, I, parts, M, I, rows, 0);
        hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)(((long)M * I + 255) / 256)), dim3(256), 0, st, parts, NPART, (long)M * I,
                           slab + pl.w_ih[l]);
        if (bf16) hipLaunchKernelGGL(gemm_tn_mfma<true>, dim3((H + TN - 1) / TN, (M + TM - 1) / TM, NPART), dim3(256), 0, st, da_l, M,
                                     hseq + (int64_t)l * BTH, H, parts, M, H, rows, T);
        else      hipLaunchKernelGGL(gemm_tn_mfma<false>, dim3((H + TN - 1) / TN, (M + TM - 1) / TM, NPART), dim3(256), 0, st, da_l, M,
                                     hseq + (int64_t)l * BTH, H, parts, M, H, rows, T);
        hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)(((long)M * H + 255) / 256)), dim3(256), 0, st, parts, NPART, (long)M * H,
                           slab + pl.w_hh[l]);
        {   // bias gradients: column sums of da over all rows, 64 row splits (the partials reuse the split-K buffer)
            const int RS = 64;                                   // RS * M <= NPART * M * max(I,H) since max(I,H) >= 64
            hipLaunchKernelGGL(colsum_part_kernel, dim3((M + 63) / 64, RS), dim3(256), 0, st, da_l, M, M, rows, parts);
            hipLaunchKernelGGL(sum_parts2_kernel, dim3((M + 255) / 256), dim3(256), 0, st, parts, RS, (long)M, slab + pl.b_ih[l], slab + pl.b_hh[l]);
        }
        NSD_CHECK_LAUNCH("batched dW");
    }
    return NSD_OK;
}
