// nsd_lstm_generic.hip -- shape-generic stacked LSTM (any C, H, L) for the shapes the fused register-resident
// kernels do not cover (e.g. BASELINE cfg3: H=256).  Same semantics as nsd_lstm2*.hip:
// self.lstm(x) of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:16-22,34 (torch.nn.LSTM: gate order i,f,g,o,
// two biases, zero initial state, dropout multipliers between layers) and autograd through it.
//
// Correctness-first formulation, one launch per layer:
//   forward   one 256-thread workgroup walks a trial; weights are streamed from L2 every step (they do not
//             fit in registers for H > 64), h / c / the layer input live in LDS.
//   backward  BPTT per trial: pre-activation gradients da[t] go to LDS and to HBM (da_seq); the transposed
//             mat-vecs read W column-wise (coalesced); the weight gradients are NOT accumulated in the time
//             loop but afterwards as  dW = da_seq^T . operand_seq  by a tiled fp32 GEMM over all (b, t)
//             (deterministic: one workgroup owns each output tile, fixed summation order).
// Large hidden sizes in training take the batched MFMA path of nsd_lstm_batched.hip instead (H % 16 == 0, B >= 16).
#include <string.h>
#include "nsd_args.h"

#define GEN_NT 256

struct GenFwdArgs {
    const float *in;          // [B,T,I] layer input
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    const float *mask;        // [B,T,H] multipliers on this layer's output (null: none)
    const float *res_in;      // [B,T,H] residual input to add (null: none)
    float *hseq, *cseq, *gact;   // saves (null in inference)
    float *out;               // [B,T,H] linked output = (h + res) * mask  (next layer's input / attention input)
    int B, T, I, H;
};

__global__ __launch_bounds__(GEN_NT) void lstm_layer_fwd_gen(GenFwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int T = a.T, I = a.I, H = a.H;
    float *in_s = lds;            // [I]
    float *h_s = in_s + I;        // [H]
    float *c_s = h_s + H;         // [H]
    float *pre = c_s + H;         // [4H]
    const int tid = threadIdx.x;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        for (int j = tid; j < H; j += GEN_NT) { h_s[j] = 0.f; c_s[j] = 0.f; }
        for (int t = 0; t < T; ++t) {
            const size_t row = (size_t)b * T + t;
            __syncthreads();
            for (int k = tid; k < I; k += GEN_NT) in_s[k] = a.in[row * I + k];
            __syncthreads();
            for (int r = tid; r < 4 * H; r += GEN_NT) {
                float acc = a.b_ih[r] + a.b_hh[r];
                const float *wi = a.w_ih + (size_t)r * I;
                for (int k = 0; k < I; ++k) acc = fmaf(wi[k], in_s[k], acc);
                const float *wh = a.w_hh + (size_t)r * H;
                for (int k = 0; k < H; ++k) acc = fmaf(wh[k], h_s[k], acc);
                pre[r] = acc;
            }
            __syncthreads();
            for (int j = tid; j < H; j += GEN_NT) {
                const float ig = fast_sigmoid(pre[j]), fg = fast_sigmoid(pre[H + j]);
                const float gg = fast_tanh(pre[2 * H + j]), og = fast_sigmoid(pre[3 * H + j]);
                const float c = fmaf(fg, c_s[j], ig * gg);
                const float h = og * fast_tanh(c);
                c_s[j] = c; h_s[j] = h;
                if (a.gact) *reinterpret_cast<float4 *>(a.gact + (row * H + j) * 4) = make_float4(ig, fg, gg, og);
                if (a.cseq) a.cseq[row * H + j] = c;
                if (a.hseq) a.hseq[row * H + j] = h;
                float o = h;
                if (a.res_in) o += a.res_in[row * H + j];
                if (a.mask) o *= a.mask[row * H + j];
                a.out[row * H + j] = o;
            }
        }
        __syncthreads();
    }
}

struct GenBwdArgs {
    const float *w_ih, *w_hh;
    const float *gact, *cseq;         // this layer's saves
    const float *dsrc;                // [B,T,H] gradient w.r.t. this layer's linked output from the layer above (null: top layer)
    const float *mask;                // [B,T,H] multipliers that were applied to this layer's output (null: none)
    const float *alpha, *dscore, *dpooled, *attn_w;   // top layer: d out_t = alpha_t * dpooled + dscore_t * attn_w
    float *da_seq;                    // [B,T,4H] pre-activation gradients (for the weight-gradient GEMMs)
    float *din_seq;                   // [B,T,I] gradient w.r.t. the layer input (null: not needed)
    int residual_add;                 // din += d(linked output) (residual pass-through; needs I == H)
    int B, T, I, H;
};

__global__ __launch_bounds__(GEN_NT) void lstm_layer_bwd_gen(GenBwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int T = a.T, I = a.I, H = a.H;
    float *da_s = lds;            // [4H]
    float *dhrec = da_s + 4 * H;  // [H]
    float *dc_s = dhrec + H;      // [H]
    float *dho = dc_s + H;        // [H] gradient w.r.t. the linked output at this step (residual pass-through)
    const int tid = threadIdx.x;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        for (int j = tid; j < H; j += GEN_NT) { dhrec[j] = 0.f; dc_s[j] = 0.f; }
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            const size_t row = (size_t)b * T + t;
            for (int j = tid; j < H; j += GEN_NT) {
                float dout;
                if (a.dsrc) dout = a.dsrc[row * H + j] * (a.mask ? a.mask[row * H + j] : 1.f);
                else        dout = fmaf(a.alpha[row], a.dpooled[(size_t)b * H + j], a.dscore[row] * a.attn_w[j]);
                dho[j] = dout;
                const float4 g = *reinterpret_cast<const float4 *>(a.gact + (row * H + j) * 4);
                const float ig = g.x, fg = g.y, gg = g.z, og = g.w;
                const float ct = a.cseq[row * H + j];
                const float cp = t > 0 ? a.cseq[(row - 1) * H + j] : 0.f;
                const float tc = fast_tanh(ct);
                const float dht = dout + dhrec[j];
                const float dct = fmaf(dht * og, 1.f - tc * tc, dc_s[j]);
                const float d0 = dct * gg * ig * (1.f - ig), d1 = dct * cp * fg * (1.f - fg);
                const float d2 = dct * ig * (1.f - gg * gg), d3 = dht * tc * og * (1.f - og);
                dc_s[j] = dct * fg;
                da_s[j] = d0; da_s[H + j] = d1; da_s[2 * H + j] = d2; da_s[3 * H + j] = d3;
                float *dg = a.da_seq + row * 4 * H;
                dg[j] = d0; dg[H + j] = d1; dg[2 * H + j] = d2; dg[3 * H + j] = d3;
            }
            __syncthreads();
            // transposed mat-vecs: lanes run along the output index (coalesced reads of W rows)
            for (int k = tid; k < H; k += GEN_NT) {
                float acc = 0.f;
                for (int r = 0; r < 4 * H; ++r) acc = fmaf(a.w_hh[(size_t)r * H + k], da_s[r], acc);
                dhrec[k] = acc;
            }
            if (a.din_seq) {
                for (int k = tid; k < I; k += GEN_NT) {
                    float acc = 0.f;
                    for (int r = 0; r < 4 * H; ++r) acc = fmaf(a.w_ih[(size_t)r * I + k], da_s[r], acc);
                    if (a.residual_add) acc += dho[k];
                    a.din_seq[row * I + k] = acc;
                }
            }
            __syncthreads();
        }
    }
}

// C[M,N] = sum_rows A[row, m] * Bm[src(row), n] over rows = B*T.  shiftT > 0: the operand row of (b,t) is (b,t-1),
// zero for t == 0 (h_{t-1}).  64x64 tile per 256-thread workgroup, 4x4 outputs per thread, 16 rows per LDS stage.
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float *A, int lda, const float *Bm, int ldb, float *Cm, int ldc,
                                                      int M, int N, long rows, int shiftT) {
    __shared__ float As[16][64 + 1], Bs[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long r0 = 0; r0 < rows; r0 += 16) {
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int rr = e >> 6, cc = e & 63;
            const long row = r0 + rr;
            float av = 0.f, bv = 0.f;
            if (row < rows) {
                if (m0 + cc < M) av = A[row * lda + m0 + cc];
                if (n0 + cc < N) {
                    if (shiftT == 0) bv = Bm[row * ldb + n0 + cc];
                    else if (row % shiftT != 0) bv = Bm[(row - 1) * ldb + n0 + cc];
                }
            }
            As[rr][cc] = av; Bs[rr][cc] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[rr][ty * 4 + i]; bv[i] = Bs[rr][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) Cm[(size_t)m * ldc + n] = acc[i][j];
        }
}

// out1[c] = out2[c] = sum_rows A[row, c]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float *A, int lda, int M, long rows, float *out1, float *out2) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < M) for (long r = g; r < rows; r += 4) s += A[r * lda + c];
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < M) {
        const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        out1[c] = v; out2[c] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------------------------
static int gen_grid(int B) { const int cap = 8 * nsd_num_cus(); return B < cap ? B : cap; }

// forward through all layers.  Train mode (save != 0): fills hseq/cseq/gact/inseq/top of the workspace.
// Inference: only `top_out` [B,T,H] is produced; `scratch2` [B,T,H] is a ping-pong buffer.
int nsd_lstm_generic_fwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, float *hseq, float *cseq, float *gact, float *inseq, float *top_out, float *scratch2,
                         hipStream_t st) {
    const int B = d->B, T = d->T, H = d->H, L = d->L;
    const int64_t BTH = (int64_t)B * T * H;
    const float *in = x;
    for (int l = 0; l < L; ++l) {
        GenFwdArgs a;
        memset(&a, 0, sizeof(a));
        a.in = in; a.I = l == 0 ? d->C : H;
        a.w_ih = params + pl.w_ih[l]; a.w_hh = params + pl.w_hh[l]; a.b_ih = params + pl.b_ih[l]; a.b_hh = params + pl.b_hh[l];
        a.mask = (l < L - 1 && drop_lstm) ? drop_lstm + (int64_t)l * BTH : nullptr;
        a.res_in = (residual && l >= 1) ? in : nullptr;
        if (hseq) { a.hseq = hseq + (int64_t)l * BTH; a.cseq = cseq + (int64_t)l * BTH; a.gact = gact + (int64_t)l * 4 * BTH; }
        if (l == L - 1) a.out = top_out;
        else if (inseq) a.out = inseq + (int64_t)l * BTH;
        else a.out = ((L - 1 - l) & 1) ? scratch2 : top_out;   // inference ping-pong; the last layer lands in top_out
        a.B = B; a.T = T; a.H = H;
        const size_t lds = ((size_t)a.I + 6 * (size_t)H) * sizeof(float);
        if (lds > 160 * 1024) { nsd_set_error("generic lstm: H=%d too large for the LDS state", H); return NSD_E_INVALID; }
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)lstm_layer_fwd_gen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lstm_layer_fwd_gen, dim3(gen_grid(B)), dim3(GEN_NT), lds, st, a);
        NSD_CHECK_LAUNCH("lstm_layer_fwd_gen");
        in = a.out;
    }
    return NSD_OK;
}

// backward through all layers; weight / bias gradients are written to slab[0 .. P_lstm)
int nsd_lstm_generic_bwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, const float *hseq, const float *cseq, const float *gact, const float *inseq,
                         const float *alpha, const float *dscore, const float *dpooled, float *da_seq, float *din_a, float *din_b,
                         float *slab, hipStream_t st) {
    const int B = d->B, T = d->T, H = d->H, L = d->L;
    const int64_t BTH = (int64_t)B * T * H;
    const long rows = (long)B * T;
    const float *dsrc = nullptr;
    for (int l = L - 1; l >= 0; --l) {
        const int I = l == 0 ? d->C : H;
        GenBwdArgs a;
        memset(&a, 0, sizeof(a));
        a.w_ih = params + pl.w_ih[l]; a.w_hh = params + pl.w_hh[l];
        a.gact = gact + (int64_t)l * 4 * BTH; a.cseq = cseq + (int64_t)l * BTH;
        a.dsrc = dsrc;
        a.mask = (l < L - 1 && drop_lstm) ? drop_lstm + (int64_t)l * BTH : nullptr;
        a.alpha = alpha; a.dscore = dscore; a.dpooled = dpooled; a.attn_w = params + pl.attn_w;
        a.da_seq = da_seq;
        a.din_seq = l > 0 ? ((l & 1) ? din_a : din_b) : nullptr;
        a.residual_add = (residual && l >= 1) ? 1 : 0;
        a.B = B; a.T = T; a.I = I; a.H = H;
        const size_t lds = 7 * (size_t)H * sizeof(float);
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)lstm_layer_bwd_gen, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lstm_layer_bwd_gen, dim3(gen_grid(B)), dim3(GEN_NT), lds, st, a);
        NSD_CHECK_LAUNCH("lstm_layer_bwd_gen");
        // weight gradients of this layer: dW_ih = da^T . in_l ; dW_hh = da^T . h_l[t-1] ; db = column sums of da
        const float *in_l = l == 0 ? x : inseq + (int64_t)(l - 1) * BTH;
        const int M = 4 * H;
        hipLaunchKernelGGL(gemm_tn_kernel, dim3((I + 63) / 64, (M + 63) / 64), dim3(256), 0, st, da_seq, M, in_l, I,
                           slab + pl.w_ih[l], I, M, I, rows, 0);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3((H + 63) / 64, (M + 63) / 64), dim3(256), 0, st, da_seq, M,
                           hseq + (int64_t)l * BTH, H, slab + pl.w_hh[l], H, M, H, rows, T);
        hipLaunchKernelGGL(colsum_kernel, dim3((M + 63) / 64), dim3(256), 0, st, da_seq, M, M, rows, slab + pl.b_ih[l], slab + pl.b_hh[l]);
        NSD_CHECK_LAUNCH("generic dW");
        dsrc = a.din_seq;
    }
    return NSD_OK;
}
