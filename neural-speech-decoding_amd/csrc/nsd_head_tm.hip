// nsd_head_tm.hip -- attention pooling over time, LayerNorm and the dense head (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:35-39,
// class softmax :97) on the TIME-MAJOR bf16 sequence the scan kernels leave: top[t][b][DH], DH = D*H.
//
// One wave per trial, 4 trials per workgroup, no LDS: lane l owns the DH/64 adjacent columns l*VPL.. of the sequence.
//   pass 1  online softmax over t: running max / denominator / weighted sum (the sequence is read once); the raw scores go
//           to alpha[] when training
//   dense   LayerNorm (biased variance, eps 1e-5), fc.0, RReLU (eval slope, explicit slopes or the counter stream), dropout,
//           fc.3, softmax / mean cross-entropy; lane f holds fc.0 unit f, lane k class k
//   train   dense backward in the same wave -> dpooled; alpha_t = exp(s_t - m) / l; pass 2 over t:
//           dscore_t = alpha_t (dpooled . h_t - dpooled . pooled)   (sum_s alpha_s dpooled . h_s == dpooled . pooled)
//           and the trial's share of d attn.weight = sum_t dscore_t h_t
//   The parameter gradients are reductions over trials of per-trial vectors: the wave writes its row of `hb`
//   (nsd_head_tm_row_floats) and head_tm_grads_kernel sums them in a fixed order (deterministic, no atomics).
#include "nsd_seq.h"

namespace {

template <int VPL>
__device__ __forceinline__ void load_bf16_vals(const bf16_t *p, float (&v)[VPL]) {
    if constexpr (VPL == 1) {
        v[0] = __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short *>(p)) << 16);
    } else if constexpr (VPL == 2) {
        const unsigned w = *reinterpret_cast<const unsigned *>(p);
        v[0] = bf16_lo(w); v[1] = bf16_hi(w);
    } else if constexpr (VPL == 4) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(p);
        v[0] = bf16_lo(w[0]); v[1] = bf16_hi(w[0]); v[2] = bf16_lo(w[1]); v[3] = bf16_hi(w[1]);
    } else {
#pragma unroll
        for (int q = 0; q < VPL / 8; ++q) {
            const u32x4 w = *reinterpret_cast<const u32x4 *>(p + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[8 * q + 2 * i] = bf16_lo(w[i]); v[8 * q + 2 * i + 1] = bf16_hi(w[i]); }
        }
    }
}
__device__ __forceinline__ float lane_bcast(const float v, const int src) { return __shfl(v, src, 64); }

template <int VPL, bool TRAIN>
__global__ __launch_bounds__(256) void head_tm_kernel(const HeadTmArgs a) {
    // U rows of the trial in flight per lane: the passes over the sequence are latency-bound (one wave per trial, 4 waves per CU at
    // B = 1024), so the bytes in flight set the rate -- 4 rows gave 1.3 TB/s
    constexpr int DH = 64 * VPL, U = VPL >= 16 ? 8 : (VPL >= 8 ? 8 : 16);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;        // one wave per trial; 4, 2 or 1 waves per workgroup (see launch_vpl)
    if (b >= a.B) return;                                       // whole waves leave; nothing below needs the workgroup
    const int c0 = lane * VPL, T = a.T, F = a.F, K = a.K;
    // a scan group of this evaluation (or of an earlier one on this workspace) timed out: the sequence below is garbage
    const bool bad = a.status != nullptr && ((a.status[0] | a.status[-NSD_SEQ_HEADER_WORDS]) & NSD_SEQ_ST_TIMEOUT_MASK) != 0;
    float aw[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) aw[v] = a.attn_w[c0 + v];
    const float ab = a.attn_b[0];
    const long arow = seq_row(0, b, T);                          // tile-major rows: step t of the trial is row arow + 32 t
    const bf16_t *seq = a.top + arow * DH + c0;

    // ---- pass 1: online softmax over time ------------------------------------------------------------------------------
    float m = -3.0e38f, l = 0.f, acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = 0.f;
    for (int t0 = 0; t0 < T; t0 += U) {
        float hv[U][VPL];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int t = t0 + q < T ? t0 + q : T - 1;
            load_bf16_vals<VPL>(seq + (long)t * 32 * DH, hv[q]);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (t0 + q < T) {                                   // (a guard, not a break: the loop must unroll for hv[q] to stay in registers)
                float part = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v) part = fmaf(hv[q][v], aw[v], part);
                const float s = wave_sum(part) + ab;
                if (TRAIN && lane == 0) a.alpha[arow + 32L * (t0 + q)] = s;          // raw score; normalised below
                const float mn = fmaxf(m, s);
                const float sc = __expf(m - mn), e = __expf(s - mn);
                l = fmaf(l, sc, e);
#pragma unroll
                for (int v = 0; v < VPL; ++v) acc[v] = fmaf(acc[v], sc, e * hv[q][v]);
                m = mn;
            }
        }
    }
    const float inv_l = 1.f / l;
    float pooled[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) pooled[v] = acc[v] * inv_l;

    // ---- LayerNorm ------------------------------------------------------------------------------------------------------
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) sum += pooled[v];
    const float mu = wave_sum(sum) * (1.f / DH);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) { const float d = pooled[v] - mu; sq = fmaf(d, d, sq); }
    const float rstd = rsqrtf(wave_sum(sq) * (1.f / DH) + 1e-5f);
    float xhat[VPL], ln[VPL], gam[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        gam[v] = a.ln_w[c0 + v];
        xhat[v] = (pooled[v] - mu) * rstd;
        ln[v] = fmaf(xhat[v], gam[v], a.ln_b[c0 + v]);
    }
    // ---- fc.0 -> RReLU -> dropout: lane f holds unit f ---------------------------------------------------------------------
    float pre = 0.f;
    for (int f = 0; f < F; ++f) {
        const float *wr = a.fc0_w + (long)f * DH + c0;
        float part = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) part = fmaf(ln[v], wr[v], part);
        const float tot = wave_sum(part);
        if (lane == f) pre = tot + a.fc0_b[f];
    }
    float slope = a.eval_slope, dmul = 1.f;
    if (TRAIN && lane < F) {
        const long hi = (long)b * F + lane;
        if (a.rng.on) {
            const float u = (float)(nsd_rand_u32(a.rng.seed, a.rng.base + 1u, (uint64_t)hi) >> 8) * (1.0f / 16777216.0f);
            slope = 0.125f + ((float)(1.0 / 3.0) - 0.125f) * u;
            dmul = nsd_rand_u32(a.rng.seed, a.rng.base + 2u, (uint64_t)hi) >= a.rng.thr_head ? a.rng.keep_head : 0.f;
        } else {
            if (a.rrelu_slope) slope = a.rrelu_slope[hi];
            if (a.drop_head) dmul = a.drop_head[hi];
        }
    }
    const float act = lane < F ? (pre >= 0.f ? pre : pre * slope) * dmul : 0.f;
    // ---- fc.3: lane k holds class k ---------------------------------------------------------------------------------------
    float logit = -3.0e38f;
    for (int k = 0; k < K; ++k) {
        const float tot = wave_sum(lane < F ? act * a.fc3_w[(long)k * F + lane] : 0.f);
        if (lane == k) logit = tot + a.fc3_b[k];
    }
    if (bad) logit = __uint_as_float(0x7fc00000u);              // NaN: logits, probabilities and the loss all carry it
    if (lane < K) a.logits[(long)b * K + lane] = logit;
    const float lmax = wave_max(logit);
    const float ex = lane < K ? __expf(logit - lmax) : 0.f;
    const float den = wave_sum(ex);
    const float prob = ex / den;
    if (a.probs && lane < K) a.probs[(long)b * K + lane] = prob;
    if constexpr (!TRAIN) return;

    // ---- mean cross-entropy and the dense backward ----------------------------------------------------------------------
    const int y = a.labels[b];
    const float ly = lane_bcast(logit, y);
    if (lane == 0) a.loss[b] = (lmax - ly) + __logf(den);
    // p_y - 1 without cancellation: -(sum of the other classes' probabilities)
    const float others = wave_sum((lane < K && lane != y) ? ex : 0.f) / den;
    const float dlog = lane < K ? (lane == y ? -others : prob) * a.scale : 0.f;
    float dact = 0.f;
    for (int k = 0; k < K; ++k) {
        const float dk = lane_bcast(dlog, k);
        if (lane < F) dact = fmaf(dk, a.fc3_w[(long)k * F + lane], dact);
    }
    const float dpre = lane < F ? dact * dmul * (pre >= 0.f ? 1.f : slope) : 0.f;
    float dln[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) dln[v] = 0.f;
    for (int f = 0; f < F; ++f) {
        const float df = lane_bcast(dpre, f);
        const float *wr = a.fc0_w + (long)f * DH + c0;
#pragma unroll
        for (int v = 0; v < VPL; ++v) dln[v] = fmaf(df, wr[v], dln[v]);
    }
    // LayerNorm backward: dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat))
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) { const float dx = dln[v] * gam[v]; s1 += dx; s2 = fmaf(dx, xhat[v], s2); }
    const float m1 = wave_sum(s1) * (1.f / DH), m2 = wave_sum(s2) * (1.f / DH);
    float dpool[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) dpool[v] = rstd * (dln[v] * gam[v] - m1 - xhat[v] * m2);
    // per-trial row for the parameter-gradient reductions
    float *row = a.hb + (long)b * a.hb_stride;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        row[c0 + v] = ln[v];
        row[DH + c0 + v] = dln[v] * xhat[v];
        row[2 * DH + c0 + v] = dln[v];
        a.pooled[(long)b * DH + c0 + v] = pooled[v];
        a.dpooled[(long)b * DH + c0 + v] = dpool[v];
    }
    if (lane < F) { row[4 * DH + lane] = dpre; row[4 * DH + F + lane] = act; }
    if (lane < K) row[4 * DH + 2 * F + lane] = dlog;

    // ---- pass 2: alpha_t, dscore_t, d attn.weight ----------------------------------------------------------------------------
    float dp_pooled = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) dp_pooled = fmaf(dpool[v], pooled[v], dp_pooled);
    dp_pooled = wave_sum(dp_pooled);
    float dattn[VPL], dsum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) dattn[v] = 0.f;
    for (int t0 = 0; t0 < T; t0 += U) {
        float hv[U][VPL], sraw[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int t = t0 + q < T ? t0 + q : T - 1;
            load_bf16_vals<VPL>(seq + (long)t * 32 * DH, hv[q]);
            sraw[q] = a.alpha[arow + 32L * t];                 // written by this wave's lane 0 in pass 1
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            if (t0 + q < T) {
                float part = 0.f;
#pragma unroll
                for (int v = 0; v < VPL; ++v) part = fmaf(hv[q][v], dpool[v], part);
                const float qd = wave_sum(part);
                const float al = __expf(sraw[q] - m) * inv_l;
                const float ds = al * (qd - dp_pooled);
                if (lane == 0) {
                    a.alpha[arow + 32L * (t0 + q)] = al;
                    a.dscore[arow + 32L * (t0 + q)] = ds;
                }
                dsum += ds;
#pragma unroll
                for (int v = 0; v < VPL; ++v) dattn[v] = fmaf(ds, hv[q][v], dattn[v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) row[3 * DH + c0 + v] = dattn[v];
    if (lane == 0) row[4 * DH + 2 * F + K] = dsum;
}

// Head parameter gradients, two stages (fixed order -> deterministic): stage 1, one thread per (gradient element, slice of the
// batch): partial sums over the slice's trials; stage 2 adds the slices and scatters into the flat gradient vector.
// Element order: d ln.weight[DH] | d ln.bias[DH] | d attn.weight[DH] | d fc.0.weight[F*DH] | d fc.0.bias[F] | d fc.3.weight[K*F] |
// d fc.3.bias[K] | d attn.bias[1]
__device__ __forceinline__ long head_grad_count(int DH, int F, int K) { return 3L * DH + (long)F * DH + F + (long)K * F + K + 1; }

__global__ __launch_bounds__(256) void head_tm_grads_part_kernel(const float *hb, long stride, int B, int DH, int F, int K, float *part) {
    const long total = head_grad_count(DH, F, K);
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b_lo = blockIdx.y * per, b_hi = b_lo + per < B ? b_lo + per : B;
    const long o_dpre = 4L * DH, o_act = o_dpre + F, o_dlog = o_act + F, o_ds = o_dlog + K;
    const long n_vec = 3L * DH, n_fc0 = (long)F * DH, n_fc3 = (long)K * F;
    long oa, ob = -1;                                           // element = sum_b hb[b][oa] (* hb[b][ob])
    long i = e;
    if (i < n_vec) { const int which = (int)(i / DH); oa = (which + 1L) * DH + (i - (long)which * DH); }
    else if ((i -= n_vec) < n_fc0) { const int f = (int)(i / DH); oa = o_dpre + f; ob = i - (long)f * DH; }
    else if ((i -= n_fc0) < F) { oa = o_dpre + i; }
    else if ((i -= F) < n_fc3) { const int k = (int)(i / F); oa = o_dlog + k; ob = o_act + (i - (long)k * F); }
    else if ((i -= n_fc3) < K) { oa = o_dlog + i; }
    else { oa = o_ds; }
    float s0 = 0.f, s1 = 0.f;
    int b = b_lo;
    if (ob < 0) {
        for (; b + 1 < b_hi; b += 2) { s0 += hb[(long)b * stride + oa]; s1 += hb[(long)(b + 1) * stride + oa]; }
        if (b < b_hi) s0 += hb[(long)b * stride + oa];
    } else {
        for (; b + 1 < b_hi; b += 2) {
            s0 = fmaf(hb[(long)b * stride + oa], hb[(long)b * stride + ob], s0);
            s1 = fmaf(hb[(long)(b + 1) * stride + oa], hb[(long)(b + 1) * stride + ob], s1);
        }
        if (b < b_hi) s0 = fmaf(hb[(long)b * stride + oa], hb[(long)b * stride + ob], s0);
    }
    part[(long)blockIdx.y * total + e] = s0 + s1;
}
__global__ __launch_bounds__(256) void head_tm_grads_sum_kernel(const float *part, int nparts, int DH, int F, int K, float *g_ln_w, float *g_ln_b,
                                                                float *g_attn_w, float *g_attn_b, float *g_fc0_w, float *g_fc0_b, float *g_fc3_w,
                                                                float *g_fc3_b) {
    const long total = head_grad_count(DH, F, K);
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int z = 0; z < nparts; ++z) s += part[(long)z * total + e];
    const long n_fc0 = (long)F * DH, n_fc3 = (long)K * F;
    long i = e;
    if (i < DH) { g_ln_w[i] = s; return; }
    if ((i -= DH) < DH) { g_ln_b[i] = s; return; }
    if ((i -= DH) < DH) { g_attn_w[i] = s; return; }
    if ((i -= DH) < n_fc0) { g_fc0_w[i] = s; return; }
    if ((i -= n_fc0) < F) { g_fc0_b[i] = s; return; }
    if ((i -= F) < n_fc3) { g_fc3_w[i] = s; return; }
    if ((i -= n_fc3) < K) { g_fc3_b[i] = s; return; }
    g_attn_b[0] = s;
}

template <int VPL>
int launch_vpl(const HeadTmArgs &a, hipStream_t st) {
    // the passes over the sequence are latency-bound, one wave per trial: with few trials (cfg5: 512 per GPU) four waves per workgroup
    // put them on half of the CUs -- spread the waves over as many CUs as there are trials
    const int cus = nsd_num_cus();
    const int wpw = a.B >= 4 * cus ? 4 : (a.B >= 2 * cus ? 2 : 1);
    const dim3 grid((a.B + wpw - 1) / wpw);
    if (a.train) hipLaunchKernelGGL((head_tm_kernel<VPL, true>), grid, dim3(64 * wpw), 0, st, a);
    else         hipLaunchKernelGGL((head_tm_kernel<VPL, false>), grid, dim3(64 * wpw), 0, st, a);
    NSD_CHECK_LAUNCH("head_tm_kernel");
    return NSD_OK;
}

}  // namespace

int nsd_head_tm_launch(const HeadTmArgs &a, hipStream_t st) {
    if (a.B < 1) return NSD_OK;
    if (a.F > 64 || a.K > 64) { nsd_set_error("head_tm: F=%d K=%d exceed 64 (one lane per unit / class)", a.F, a.K); return NSD_E_INVALID; }
    switch (a.DH) {
    case 64: return launch_vpl<1>(a, st);
    case 128: return launch_vpl<2>(a, st);
    case 256: return launch_vpl<4>(a, st);
    case 512: return launch_vpl<8>(a, st);
    case 1024: return launch_vpl<16>(a, st);
    default: nsd_set_error("head_tm: sequence width %d not covered (64, 128, 256, 512, 1024)", a.DH); return NSD_E_INVALID;
    }
}

int nsd_head_tm_grads_launch(const float *hb, long hb_stride, int B, int DH, int F, int K, float *scratch, float *g_ln_w, float *g_ln_b,
                             float *g_attn_w, float *g_attn_b, float *g_fc0_w, float *g_fc0_b, float *g_fc3_w, float *g_fc3_b, hipStream_t st) {
    const long total = 3L * DH + (long)F * DH + F + (long)K * F + K + 1;
    const int NB = B >= 512 ? 16 : (B >= 64 ? 4 : 1);           // scratch: NB * total floats
    const unsigned gx = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(head_tm_grads_part_kernel, dim3(gx, NB), dim3(256), 0, st, hb, hb_stride, B, DH, F, K, scratch);
    hipLaunchKernelGGL(head_tm_grads_sum_kernel, dim3(gx), dim3(256), 0, st, scratch, NB, DH, F, K, g_ln_w, g_ln_b, g_attn_w, g_attn_b, g_fc0_w,
                       g_fc0_b, g_fc3_w, g_fc3_b);
    NSD_CHECK_LAUNCH("head_tm_grads");
    return NSD_OK;
}
