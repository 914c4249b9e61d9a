// nsd_head.hip -- attention pooling over time + LayerNorm + dense head (+ softmax / CE), forward and backward.
//
// Replaces lines 35-39 of EEG_LSTM.forward (Neuro-Alpha-App/Utilities/lstm_eeg_model.py):
//   scores = attn(out).squeeze(-1); weights = softmax(scores, dim=1); out = (out*weights[...,None]).sum(1)
//   out = ln(out); return fc(out)          fc = Linear(H,F) -> RReLU -> Dropout -> Linear(F,K)
// and the class softmax of SimplePredictor.predict (lstm_eeg_model.py:97).
//
// One 256-thread workgroup per trial (grid-strided over trials).  The [T,H] sequence of a trial (48 KB at
// T=250,H=48) is read twice per pass (scores, weighted sum); rows are contiguous so every wave reads whole
// 128-B lines.  Reductions over T / H use wave shuffles + one LDS hop.  Shape-generic in T,H,F,K.
#include "nsd_args.h"

#define HEAD_NT 256

__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float *red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}


// dynamic LDS layout (floats): sc[T] | vecH[4][H] | vecF[2][F] | vecK[K] | part[max(H,256)... see below] | red[8]
__device__ __forceinline__ void weighted_rowsum(const float *seq, int rs, const float *w_t, int T, int H, float *part,
                                                float *out /*LDS [H]*/) {
    // out[j] = sum_t w_t[t] * seq[t][j].  Lanes run along j (coalesced rows), parts run along t.
    const int HL = H < HEAD_NT ? H : HEAD_NT;
    const int nparts = HEAD_NT / HL;
    const int lane_j = threadIdx.x % HL, p = threadIdx.x / HL;
    for (int j0 = 0; j0 < H; j0 += HL) {
        const int j = j0 + lane_j;
        float acc = 0.f;
        if (p < nparts && j < H)
            for (int t = p; t < T; t += nparts) acc = fmaf(w_t[t], seq[(size_t)t * rs + j], acc);
        __syncthreads();
        if (p < nparts) part[p * HL + lane_j] = acc;
        __syncthreads();
        if (threadIdx.x < HL && j0 + threadIdx.x < H) {
            float s = 0.f;
            for (int q = 0; q < nparts; ++q) s += part[q * HL + threadIdx.x];
            out[j0 + threadIdx.x] = s;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(HEAD_NT) void head_fwd_kernel(HeadArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int T = a.T, H = a.H, F = a.F, K = a.K;
    float *sseq = lds;                // [T][stage_stride] staged sequence (16-byte aligned), empty when not staging
    float *sc = lds + (size_t)T * a.stage_stride;   // [T]
    float *vaw = sc + T;              // [H] attn weights
    float *vp = vaw + H;              // [H] pooled
    float *vln = vp + H;              // [H] ln out
    float *vz = vln + H;              // [F]
    float *vlg = vz + F;              // [K]
    float *part = vlg + K;            // [256]
    float *red = part + HEAD_NT;      // [8]
    const int tid = threadIdx.x;

    for (int j = tid; j < H; j += HEAD_NT) vaw[j] = a.attn_w[j];
    const float ab = a.attn_b[0];
    __syncthreads();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float *seq = a.top + (size_t)b * T * H;
        int rs = H;
        if (a.stage_stride) {
            // the trial's [T,H] sequence is read twice: stage it in LDS once with 16-byte loads (rows padded so
            // that a lane-per-row ds_read_b128 sweep is bank-conflict free)
            __syncthreads();
            const int h4 = H >> 2;
            for (int e = tid; e < T * h4; e += HEAD_NT) {
                const int t = e / h4, q = e - t * h4;
                *reinterpret_cast<float4 *>(sseq + (size_t)t * a.stage_stride + 4 * q) =
                    *reinterpret_cast<const float4 *>(seq + (size_t)t * H + 4 * q);
            }
            __syncthreads();
            seq = sseq; rs = a.stage_stride;
        }
        // scores over time (lstm_eeg_model.py:35)
        float lmax = -INFINITY;
        for (int t = tid; t < T; t += HEAD_NT) {
            const float *row = seq + (size_t)t * rs;
            float s = ab;
            if ((H & 3) == 0) {
                for (int j = 0; j < H; j += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(row + j);
                    s = fmaf(v.x, vaw[j], s); s = fmaf(v.y, vaw[j + 1], s);
                    s = fmaf(v.z, vaw[j + 2], s); s = fmaf(v.w, vaw[j + 3], s);
                }
            } else {
                for (int j = 0; j < H; ++j) s = fmaf(row[j], vaw[j], s);
            }
            sc[t] = s;
            lmax = fmaxf(lmax, s);
        }
        const float mx = block_max(lmax, red);
        // softmax over TIME (lstm_eeg_model.py:36)
        float lsum = 0.f;
        for (int t = tid; t < T; t += HEAD_NT) { const float e = __expf(sc[t] - mx); sc[t] = e; lsum += e; }
        const float den = block_sum(lsum, red);
        const float rden = 1.0f / den;
        for (int t = tid; t < T; t += HEAD_NT) {
            const float al = sc[t] * rden;
            sc[t] = al;
            if (a.alpha) a.alpha[(size_t)b * T + t] = al;
        }
        __syncthreads();
        // weighted sum over time (lstm_eeg_model.py:37)
        weighted_rowsum(seq, rs, sc, T, H, part, vp);
        // LayerNorm (lstm_eeg_model.py:38): biased variance, eps inside the sqrt
        float ls = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) { ls += vp[j]; if (a.pooled) a.pooled[(size_t)b * H + j] = vp[j]; }
        const float mu = block_sum(ls, red) / (float)H;
        float lv = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) { const float d = vp[j] - mu; lv += d * d; }
        const float var = block_sum(lv, red) / (float)H;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int j = tid; j < H; j += HEAD_NT) vln[j] = (vp[j] - mu) * rstd * a.ln_w[j] + a.ln_b[j];
        __syncthreads();
        // fc.0 -> RReLU -> Dropout (lstm_eeg_model.py:26-28)
        for (int f = tid; f < F; f += HEAD_NT) {
            float acc = a.fc0_b[f];
            const float *w = a.fc0_w + (size_t)f * H;
            for (int j = 0; j < H; ++j) acc = fmaf(w[j], vln[j], acc);
            if (a.fc0_pre) a.fc0_pre[(size_t)b * F + f] = acc;
            const float sl = a.rrelu_slope ? a.rrelu_slope[(size_t)b * F + f] : a.eval_slope;
            float v = acc >= 0.f ? acc : acc * sl;
            if (a.drop_head) v *= a.drop_head[(size_t)b * F + f];
            vz[f] = v;
        }
        __syncthreads();
        // fc.3 (lstm_eeg_model.py:29) and optional class softmax (lstm_eeg_model.py:97)
        for (int k = tid; k < K; k += HEAD_NT) {
            float acc = a.fc3_b[k];
            const float *w = a.fc3_w + (size_t)k * F;
            for (int f = 0; f < F; ++f) acc = fmaf(w[f], vz[f], acc);
            vlg[k] = acc;
            a.logits[(size_t)b * K + k] = acc;
        }
        __syncthreads();
        if (a.probs && tid == 0) {
            float m2 = vlg[0];
            for (int k = 1; k < K; ++k) m2 = fmaxf(m2, vlg[k]);
            float d = 0.f;
            for (int k = 0; k < K; ++k) d += __expf(vlg[k] - m2);
            for (int k = 0; k < K; ++k) a.probs[(size_t)b * K + k] = __expf(vlg[k] - m2) / d;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(HEAD_NT) void head_bwd_kernel(HeadArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int T = a.T, H = a.H, F = a.F, K = a.K;
    float *sseq = lds;                // [T][stage_stride] staged sequence (16-byte aligned), empty when not staging
    float *sc = lds + (size_t)T * a.stage_stride;   // [T]  alpha, later dscore
    float *vaw = sc + T;              // [H]
    float *vx = vaw + H;              // [H] xhat
    float *vln = vx + H;              // [H] ln out
    float *vdp = vln + H;             // [H] d pooled
    float *vz = vdp + H;              // [F] activated fc0 output
    float *vdz = vz + F;              // [F]
    float *vdl = vdz + F;             // [K]
    float *part = vdl + K;            // [256]
    float *red = part + HEAD_NT;      // [8]
    const int tid = threadIdx.x;

    for (int j = tid; j < H; j += HEAD_NT) vaw[j] = a.attn_w[j];
    __syncthreads();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float *seq = a.top + (size_t)b * T * H;
        int rs = H;
        if (a.stage_stride) {
            // the trial's [T,H] sequence is read twice: stage it in LDS once with 16-byte loads (rows padded so
            // that a lane-per-row ds_read_b128 sweep is bank-conflict free)
            __syncthreads();
            const int h4 = H >> 2;
            for (int e = tid; e < T * h4; e += HEAD_NT) {
                const int t = e / h4, q = e - t * h4;
                *reinterpret_cast<float4 *>(sseq + (size_t)t * a.stage_stride + 4 * q) =
                    *reinterpret_cast<const float4 *>(seq + (size_t)t * H + 4 * q);
            }
            __syncthreads();
            seq = sseq; rs = a.stage_stride;
        }
        float *slab = a.hslabs + (size_t)b * a.Ph;
        // dlogits (given, or mean-CE from labels)
        if (tid == 0) {
            if (a.dlogits) {
                for (int k = 0; k < K; ++k) vdl[k] = a.dlogits[(size_t)b * K + k];
            } else {
                const float *lg = a.logits_in + (size_t)b * K;
                const int y = a.labels[b];
                float m2 = lg[0];
                for (int k = 1; k < K; ++k) m2 = fmaxf(m2, lg[k]);
                // softmax - onehot without cancellation: for the label class p_y - 1 = -(sum_{k != y} e_k) / d
                float d = 0.f, rest = 0.f;
                for (int k = 0; k < K; ++k) { const float e = expf(lg[k] - m2); d += e; if (k != y) rest += e; }
                for (int k = 0; k < K; ++k) vdl[k] = (k == y ? -rest / d : expf(lg[k] - m2) / d) * a.scale;
                if (a.loss) a.loss[b] = -((lg[y] - m2) - logf(d));
            }
        }
        // recompute LayerNorm statistics from the saved pooled vector
        float ls = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) { vdp[j] = a.pooled[(size_t)b * H + j]; ls += vdp[j]; }
        const float mu = block_sum(ls, red) / (float)H;
        float lv = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) { const float d = vdp[j] - mu; lv += d * d; }
        const float var = block_sum(lv, red) / (float)H;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int j = tid; j < H; j += HEAD_NT) {
            const float xh = (vdp[j] - mu) * rstd;
            vx[j] = xh;
            vln[j] = xh * a.ln_w[j] + a.ln_b[j];
        }
        // activated fc.0 output, and dz through fc.3 / dropout / RReLU
        for (int f = tid; f < F; f += HEAD_NT) {
            const float pre = a.fc0_pre[(size_t)b * F + f];
            const float sl = a.rrelu_slope ? a.rrelu_slope[(size_t)b * F + f] : a.eval_slope;
            const float mk = a.drop_head ? a.drop_head[(size_t)b * F + f] : 1.f;
            vz[f] = (pre >= 0.f ? pre : pre * sl) * mk;
        }
        __syncthreads();
        for (int f = tid; f < F; f += HEAD_NT) {
            const float pre = a.fc0_pre[(size_t)b * F + f];
            const float sl = a.rrelu_slope ? a.rrelu_slope[(size_t)b * F + f] : a.eval_slope;
            const float mk = a.drop_head ? a.drop_head[(size_t)b * F + f] : 1.f;
            float d = 0.f;
            for (int k = 0; k < K; ++k) d = fmaf(a.fc3_w[(size_t)k * F + f], vdl[k], d);
            d *= mk;
            d = pre >= 0.f ? d : d * sl;
            vdz[f] = d;
            slab[a.o_fc0_b + f] = d;
        }
        for (int e = tid; e < K * F; e += HEAD_NT) slab[a.o_fc3_w + e] = vdl[e / F] * vz[e % F];
        for (int k = tid; k < K; k += HEAD_NT) slab[a.o_fc3_b + k] = vdl[k];
        __syncthreads();
        for (int e = tid; e < F * H; e += HEAD_NT) slab[a.o_fc0_w + e] = vdz[e / H] * vln[e % H];
        // d ln_out, LayerNorm backward
        float l1 = 0.f, l2 = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) {
            float d = 0.f;
            for (int f = 0; f < F; ++f) d = fmaf(a.fc0_w[(size_t)f * H + j], vdz[f], d);
            slab[a.o_ln_w + j] = d * vx[j];
            slab[a.o_ln_b + j] = d;
            const float dxh = d * a.ln_w[j];
            vdp[j] = dxh;
            l1 += dxh; l2 += dxh * vx[j];
        }
        const float m1 = block_sum(l1, red) / (float)H;
        const float m2 = block_sum(l2, red) / (float)H;
        for (int j = tid; j < H; j += HEAD_NT) {
            const float d = rstd * (vdp[j] - m1 - vx[j] * m2);
            vdp[j] = d;
            a.dpooled[(size_t)b * H + j] = d;
        }
        __syncthreads();
        // attention backward: dalpha_t = dp . out_t ; ds = alpha * (dalpha - sum alpha*dalpha)
        float lsd = 0.f;
        for (int t = tid; t < T; t += HEAD_NT) {
            const float *row = seq + (size_t)t * rs;
            float d = 0.f;
            for (int j = 0; j < H; ++j) d = fmaf(row[j], vdp[j], d);
            const float al = a.alpha[(size_t)b * T + t];
            sc[t] = d;
            lsd = fmaf(al, d, lsd);
        }
        const float sdot = block_sum(lsd, red);
        float lb = 0.f;
        for (int t = tid; t < T; t += HEAD_NT) {
            const float ds = a.alpha[(size_t)b * T + t] * (sc[t] - sdot);
            sc[t] = ds;
            a.dscore[(size_t)b * T + t] = ds;
            if (a.adpack)
                *reinterpret_cast<float4 *>(a.adpack + ((size_t)b * T + t) * 4) = make_float4(a.alpha[(size_t)b * T + t], ds, 0.f, 0.f);
            lb += ds;
        }
        const float dab = block_sum(lb, red);
        if (tid == 0) slab[a.o_attn_b] = dab;
        // d attn.weight[j] = sum_t ds_t * out_t[j]
        weighted_rowsum(seq, rs, sc, T, H, part, vx);
        for (int j = tid; j < H; j += HEAD_NT) slab[a.o_attn_w + j] = vx[j];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Fused train-step head: forward, mean-CE and backward of one trial in ONE pass of one workgroup.
// Everything lives in LDS: the trial's [T,H] sequence (read from HBM exactly once), the head's
// parameters (staged once per workgroup), and every intermediate.  The only HBM latencies left on
// the kernel's critical path are those two staging loads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HEAD_NT) void head_train_kernel(HeadArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int T = a.T, H = a.H, F = a.F, K = a.K, rs = a.stage_stride;
    float *sseq = lds;                                   // [T][rs]
    float *hp = sseq + (size_t)T * rs;                   // [Ph] parameters, same offsets as a head slab
    float *sc = hp + ((a.Ph + 3) & ~3L);                 // [T]
    float *vp = sc + T;                                  // [H] pooled
    float *vx = vp + H;                                  // [H] xhat
    float *vln = vx + H;                                 // [H] LayerNorm output
    float *vdp = vln + H;                                // [H] d pooled
    float *vpre = vdp + H;                               // [F] fc.0 pre-activation
    float *vz = vpre + F;                                // [F] activated
    float *vdz = vz + F;                                 // [F]
    float *vlg = vdz + F;                                // [K]
    float *vdl = vlg + K;                                // [K]
    float *part = vdl + K;                               // [256]
    float *red = part + HEAD_NT;                         // [8]
    const int tid = threadIdx.x;
    const float *p_lnw = hp + a.o_ln_w, *p_lnb = hp + a.o_ln_b, *p_aw = hp + a.o_attn_w;
    const float *p_w0 = hp + a.o_fc0_w, *p_b0 = hp + a.o_fc0_b, *p_w3 = hp + a.o_fc3_w, *p_b3 = hp + a.o_fc3_b;

    for (long e = tid; e < a.Ph; e += HEAD_NT) hp[e] = a.ln_w[e];     // head parameters are contiguous from ln.weight on
    __syncthreads();
    const float ab = hp[a.o_attn_b];

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float *gseq = a.top + (size_t)b * T * H;
        float *slab = a.hslabs + (size_t)b * a.Ph;
        const int h4 = H >> 2;
        for (int e = tid; e < T * h4; e += HEAD_NT) {
            const int t = e / h4, q = e - t * h4;
            *reinterpret_cast<float4 *>(sseq + (size_t)t * rs + 4 * q) = *reinterpret_cast<const float4 *>(gseq + (size_t)t * H + 4 * q);
        }
        // per-trial small inputs, fetched while the sequence lands
        float sl_f = a.eval_slope, mk_f = 1.f;
        if (tid < F) {
            if (a.rrelu_slope) sl_f = a.rrelu_slope[(size_t)b * F + tid];
            if (a.drop_head) mk_f = a.drop_head[(size_t)b * F + tid];
        }
        const int label = a.labels[b];
        __syncthreads();
        // ---- forward ----
        float lmax = -INFINITY;
        for (int t = tid; t < T; t += HEAD_NT) {
            const float *row = sseq + (size_t)t * rs;
            float s = ab;
            for (int j = 0; j < H; j += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + j);
                s = fmaf(v.x, p_aw[j], s); s = fmaf(v.y, p_aw[j + 1], s); s = fmaf(v.z, p_aw[j + 2], s); s = fmaf(v.w, p_aw[j + 3], s);
            }
            sc[t] = s;
            lmax = fmaxf(lmax, s);
        }
        const float mx = block_max(lmax, red);
        float lsum = 0.f;
        for (int t = tid; t < T; t += HEAD_NT) { const float e = __expf(sc[t] - mx); sc[t] = e; lsum += e; }
        const float rden = 1.0f / block_sum(lsum, red);
        for (int t = tid; t < T; t += HEAD_NT) { const float al = sc[t] * rden; sc[t] = al; a.alpha[(size_t)b * T + t] = al; }
        __syncthreads();
        weighted_rowsum(sseq, rs, sc, T, H, part, vp);
        float ls = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) { ls += vp[j]; a.pooled[(size_t)b * H + j] = vp[j]; }
        const float mu = block_sum(ls, red) / (float)H;
        float lv = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) { const float d = vp[j] - mu; lv += d * d; }
        const float rstd = 1.0f / sqrtf(block_sum(lv, red) / (float)H + 1e-5f);
        for (int j = tid; j < H; j += HEAD_NT) { const float xh = (vp[j] - mu) * rstd; vx[j] = xh; vln[j] = xh * p_lnw[j] + p_lnb[j]; }
        __syncthreads();
        if (tid < F) {
            float acc = p_b0[tid];
            const float *w = p_w0 + (size_t)tid * H;
            for (int j = 0; j < H; ++j) acc = fmaf(w[j], vln[j], acc);
            vpre[tid] = acc;
            a.fc0_pre[(size_t)b * F + tid] = acc;
            vz[tid] = (acc >= 0.f ? acc : acc * sl_f) * mk_f;
        }
        __syncthreads();
        if (tid < K) {
            float acc = p_b3[tid];
            for (int f = 0; f < F; ++f) acc = fmaf(p_w3[(size_t)tid * F + f], vz[f], acc);
            vlg[tid] = acc;
            a.logits[(size_t)b * K + tid] = acc;
        }
        __syncthreads();
        // ---- mean cross-entropy: dlogits = (softmax - onehot) * scale, without cancellation for the label ----
        if (tid == 0) {
            float m2 = vlg[0];
            for (int k = 1; k < K; ++k) m2 = fmaxf(m2, vlg[k]);
            float d = 0.f, rest = 0.f;
            for (int k = 0; k < K; ++k) { const float e = expf(vlg[k] - m2); d += e; if (k != label) rest += e; }
            for (int k = 0; k < K; ++k) vdl[k] = (k == label ? -rest / d : expf(vlg[k] - m2) / d) * a.scale;
            a.loss[b] = -((vlg[label] - m2) - logf(d));
        }
        __syncthreads();
        // ---- backward ----
        if (tid < F) {
            float d = 0.f;
            for (int k = 0; k < K; ++k) d = fmaf(p_w3[(size_t)k * F + tid], vdl[k], d);
            d *= mk_f;
            d = vpre[tid] >= 0.f ? d : d * sl_f;
            vdz[tid] = d;
            slab[a.o_fc0_b + tid] = d;
        }
        for (int e = tid; e < K * F; e += HEAD_NT) slab[a.o_fc3_w + e] = vdl[e / F] * vz[e % F];
        if (tid < K) slab[a.o_fc3_b + tid] = vdl[tid];
        __syncthreads();
        for (int e = tid; e < F * H; e += HEAD_NT) slab[a.o_fc0_w + e] = vdz[e / H] * vln[e % H];
        float l1 = 0.f, l2 = 0.f;
        for (int j = tid; j < H; j += HEAD_NT) {
            float d = 0.f;
            for (int f = 0; f < F; ++f) d = fmaf(p_w0[(size_t)f * H + j], vdz[f], d);
            slab[a.o_ln_w + j] = d * vx[j];
            slab[a.o_ln_b + j] = d;
            const float dxh = d * p_lnw[j];
            vdp[j] = dxh;
            l1 += dxh; l2 += dxh * vx[j];
        }
        const float m1 = block_sum(l1, red) / (float)H;
        const float m2 = block_sum(l2, red) / (float)H;
        for (int j = tid; j < H; j += HEAD_NT) {
            const float d = rstd * (vdp[j] - m1 - vx[j] * m2);
            vdp[j] = d;
            a.dpooled[(size_t)b * H + j] = d;
        }
        __syncthreads();
        float lsd = 0.f;
        float dal[4];                                        // up to 4 time steps per thread (T <= 1024 staged)
        int nt = 0;
        for (int t = tid; t < T; t += HEAD_NT, ++nt) {
            const float *row = sseq + (size_t)t * rs;
            float d = 0.f;
            for (int j = 0; j < H; j += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + j);
                d = fmaf(v.x, vdp[j], d); d = fmaf(v.y, vdp[j + 1], d); d = fmaf(v.z, vdp[j + 2], d); d = fmaf(v.w, vdp[j + 3], d);
            }
            if (nt < 4) dal[nt] = d;
            lsd = fmaf(sc[t], d, lsd);
        }
        const float sdot = block_sum(lsd, red);
        float lb = 0.f;
        nt = 0;
        for (int t = tid; t < T; t += HEAD_NT, ++nt) {
            const float al = sc[t];
            const float ds = al * (dal[nt < 4 ? nt : 3] - sdot);
            a.dscore[(size_t)b * T + t] = ds;
            *reinterpret_cast<float4 *>(a.adpack + ((size_t)b * T + t) * 4) = make_float4(al, ds, 0.f, 0.f);
            lb += ds;
            sc[t] = ds;                                      // alpha is no longer needed: reuse for d attn.weight
        }
        const float dab = block_sum(lb, red);
        if (tid == 0) slab[a.o_attn_b] = dab;
        weighted_rowsum(sseq, rs, sc, T, H, part, vx);
        for (int j = tid; j < H; j += HEAD_NT) slab[a.o_attn_w + j] = vx[j];
        __syncthreads();
    }
}

// fused head is used when everything fits in LDS and T <= 4*256 (per-thread d-alpha registers)
static bool head_train_fits(const HeadArgs &a, int *stride_out, size_t *lds_out) {
    if ((a.H & 3) != 0 || a.T > 4 * HEAD_NT || a.F > HEAD_NT || a.K > HEAD_NT) return false;
    int stride = a.H + 4;
    if (((stride >> 2) & 1) == 0) stride += 4;
    const size_t n = (size_t)a.T * stride + ((a.Ph + 3) & ~3L) + a.T + 4 * (size_t)a.H + 3 * (size_t)a.F + 2 * (size_t)a.K + HEAD_NT + 8;
    if (n * sizeof(float) > 150 * 1024) return false;
    *stride_out = stride; *lds_out = n * sizeof(float);
    return true;
}

// returns 1 if launched, 0 if the shape does not fit (caller falls back to head_fwd + head_bwd), <0 on error
int nsd_head_train_launch(const HeadArgs &a_in, hipStream_t st) {
    HeadArgs a = a_in;
    int stride; size_t lds;
    if (!head_train_fits(a, &stride, &lds)) return 0;
    if (a.B <= 0) return 1;
    a.stage_stride = stride;
    const int cap = 8 * nsd_num_cus();
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)head_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(head_train_kernel, dim3(a.B < cap ? a.B : cap), dim3(HEAD_NT), lds, st, a);
    NSD_CHECK_LAUNCH("head_train");
    return 1;
}

static size_t head_lds_bytes(int T, int H, int F, int K, bool bwd) {
    size_t n = (size_t)T + (bwd ? 4 : 3) * (size_t)H + (bwd ? 2 : 1) * (size_t)F + K + HEAD_NT + 8;
    return n * sizeof(float);
}

int nsd_head_launch(const HeadArgs &a_in, bool bwd, hipStream_t st) {
    HeadArgs a = a_in;
    if (a.B <= 0) return NSD_OK;
    size_t lds = head_lds_bytes(a.T, a.H, a.F, a.K, bwd);
    // stage the sequence in LDS when it fits next to the small vectors (<= 144 KB): row stride = H rounded up
    // to 4 (mod 8) floats
    a.stage_stride = 0;
    if ((a.H & 3) == 0) {
        int stride = a.H + 4;
        if (((stride >> 2) & 1) == 0) stride += 4;
        const size_t need = lds + (size_t)a.T * stride * sizeof(float);
        if (need <= 144 * 1024) { a.stage_stride = stride; lds = need; }
    }
    if (lds > 160 * 1024 || a.H > HEAD_NT * 8) {
        nsd_set_error("head: T=%d H=%d needs %zu B of LDS (max 163840)", a.T, a.H, lds);
        return NSD_E_INVALID;
    }
    const int cap = 8 * nsd_num_cus();
    const int grid = a.B < cap ? a.B : cap;
    if (bwd) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(head_bwd_kernel, dim3(grid), dim3(HEAD_NT), lds, st, a);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)head_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(head_fwd_kernel, dim3(grid), dim3(HEAD_NT), lds, st, a);
    }
    NSD_CHECK_LAUNCH(bwd ? "head_bwd" : "head_fwd");
    return NSD_OK;
}
