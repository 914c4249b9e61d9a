/*
 * nsd_oracle.c -- CPU restatement of the reference's EEG_LSTM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import, link
 * or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * What it restates (reference paths relative to the upstream repo root):
 *   Neuro-Alpha-App/Utilities/lstm_eeg_model.py:14-30   parameter set / module structure
 *   Neuro-Alpha-App/Utilities/lstm_eeg_model.py:32-39   EEG_LSTM.forward
 *   Neuro-Alpha-App/Utilities/lstm_eeg_model.py:97      softmax over classes (predict)
 *   Neuro-Alpha-App/Frontend/app.py:166-170             normalize_eeg (per-channel z-score)
 * The arithmetic of nn.LSTM / nn.LayerNorm / nn.RReLU / nn.Linear lives in
 * PyTorch (third party, not vendored, version unpinned by the reference); the
 * cell follows torch.nn.LSTM's documented semantics: gate order i,f,g,o, two
 * bias vectors, zero initial state, inter-layer dropout on all but the last
 * layer.  Parity is pinned by the .npz files under tests/golden/, produced by
 * tests/golden/make_goldens.py from the reference module + checkpoint.
 *
 * All arithmetic is fp32 in forward (k-ordered fmaf-free multiply-add chains);
 * the backward keeps fp32 values but accumulates long sums (over B*T) in
 * double so that it is a tighter checker than either oneDNN or the HIP path.
 *
 * Flat parameter layout (identical to the reference state_dict order, see
 * nsd_oracle_layout): for l in 0..L-1 { w_ih[4H,I_l], w_hh[4H,H], b_ih[4H],
 * b_hh[4H] }, ln.w[H], ln.b[H], attn.w[H], attn.b[1], fc0.w[F,H], fc0.b[F],
 * fc3.w[K,F], fc3.b[K];  I_0 = C, I_l = H.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define NSD_MAX_LAYERS 8

typedef struct {
    long w_ih[NSD_MAX_LAYERS], w_hh[NSD_MAX_LAYERS], b_ih[NSD_MAX_LAYERS], b_hh[NSD_MAX_LAYERS];
    long ln_w, ln_b, attn_w, attn_b, fc0_w, fc0_b, fc3_w, fc3_b, total;
} nsd_layout;

static void layout(int C, int H, int L, int K, int F, nsd_layout *o) {
    long p = 0;
    for (int l = 0; l < L; ++l) {
        int I = l == 0 ? C : H;
        o->w_ih[l] = p; p += 4L * H * I;
        o->w_hh[l] = p; p += 4L * H * H;
        o->b_ih[l] = p; p += 4L * H;
        o->b_hh[l] = p; p += 4L * H;
    }
    o->ln_w = p; p += H;   o->ln_b = p; p += H;
    o->attn_w = p; p += H; o->attn_b = p; p += 1;
    o->fc0_w = p; p += (long)F * H; o->fc0_b = p; p += F;
    o->fc3_w = p; p += (long)K * F; o->fc3_b = p; p += K;
    o->total = p;
}

long nsd_oracle_param_count(int C, int H, int L, int K, int F) {
    if (L < 1 || L > NSD_MAX_LAYERS) return -1;
    nsd_layout lo; layout(C, H, L, K, F, &lo); return lo.total;
}

/* offsets[0..4L) = per-layer w_ih,w_hh,b_ih,b_hh; then ln_w,ln_b,attn_w,attn_b,fc0_w,fc0_b,fc3_w,fc3_b */
int nsd_oracle_layout(int C, int H, int L, int K, int F, long *offsets) {
    if (L < 1 || L > NSD_MAX_LAYERS) return -1;
    nsd_layout lo; layout(C, H, L, K, F, &lo);
    for (int l = 0; l < L; ++l) {
        offsets[4*l+0] = lo.w_ih[l]; offsets[4*l+1] = lo.w_hh[l];
        offsets[4*l+2] = lo.b_ih[l]; offsets[4*l+3] = lo.b_hh[l];
    }
    long *q = offsets + 4*L;
    q[0]=lo.ln_w; q[1]=lo.ln_b; q[2]=lo.attn_w; q[3]=lo.attn_b;
    q[4]=lo.fc0_w; q[5]=lo.fc0_b; q[6]=lo.fc3_w; q[7]=lo.fc3_b;
    return 0;
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* nn.RReLU eval slope: (lower+upper)/2 computed in double then cast (lstm_eeg_model.py:27) */
float nsd_oracle_rrelu_eval_slope(void) { return (float)((0.125 + 1.0/3.0) / 2.0); }

/*
 * Forward.  x[B,T,C].  Optional inputs (NULL = absent):
 *   drop_lstm[(L-1),B,T,H]  multiplier mask applied to the output of layer l<L-1 (nn.LSTM dropout)
 *   rrelu_slope[B,F]        per-element negative slope (train-mode RReLU noise); NULL -> eval slope
 *   drop_head[B,F]          multiplier mask of nn.Dropout in fc (lstm_eeg_model.py:28)
 *   residual                extension (not in reference): out_l = LSTM_l(in_l) + in_l for l>=1
 * Outputs (any may be NULL): logits[B,K], probs[B,K], and the saves used by
 * backward / intermediate goldens:
 *   hseq[L,B,T,H] (LSTM's own h), cseq[L,B,T,H], gates[L,B,T,4,H] (activated i,f,g,o),
 *   alpha[B,T], pooled[B,H], ln_out[B,H], fc0_pre[B,F] (pre-activation), fc0_act[B,F] (after rrelu+dropout)
 */
int nsd_oracle_forward(int B, int T, int C, int H, int L, int K, int F,
                       const float *params, const float *x,
                       const float *drop_lstm, const float *rrelu_slope, const float *drop_head,
                       int residual,
                       float *logits, float *probs,
                       float *hseq, float *cseq, float *gates,
                       float *alpha, float *pooled, float *ln_out, float *fc0_pre, float *fc0_act)
{
    if (L < 1 || L > NSD_MAX_LAYERS || B < 0 || T < 1) return -1;
    nsd_layout lo; layout(C, H, L, K, F, &lo);
    const float eval_slope = nsd_oracle_rrelu_eval_slope();
    float *cur = (float*)malloc(sizeof(float) * (size_t)T * (H > C ? H : C)); /* layer input seq */
    float *out = (float*)malloc(sizeof(float) * (size_t)T * H);               /* layer output seq */
    float *g   = (float*)malloc(sizeof(float) * 4 * H);
    float *h   = (float*)malloc(sizeof(float) * H);
    float *c   = (float*)malloc(sizeof(float) * H);
    float *sc  = (float*)malloc(sizeof(float) * T);
    float *pl  = (float*)malloc(sizeof(float) * H);
    float *lnv = (float*)malloc(sizeof(float) * H);
    float *z   = (float*)malloc(sizeof(float) * F);
    float *lg  = (float*)malloc(sizeof(float) * K);
    if (!cur||!out||!g||!h||!c||!sc||!pl||!lnv||!z||!lg) return -2;

    for (int b = 0; b < B; ++b) {
        /* ---- stacked LSTM: lstm_eeg_model.py:34 ---- */
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < C; ++k) cur[(size_t)t*C + k] = x[((size_t)b*T + t)*C + k];
        int I = C;
        for (int l = 0; l < L; ++l) {
            const float *Wih = params + lo.w_ih[l], *Whh = params + lo.w_hh[l];
            const float *bih = params + lo.b_ih[l], *bhh = params + lo.b_hh[l];
            for (int j = 0; j < H; ++j) { h[j] = 0.f; c[j] = 0.f; }
            for (int t = 0; t < T; ++t) {
                const float *in = cur + (size_t)t * I;
                for (int r = 0; r < 4*H; ++r) {
                    float a = bih[r] + bhh[r];
                    const float *wi = Wih + (size_t)r * I;
                    for (int k = 0; k < I; ++k) a += wi[k] * in[k];
                    const float *wh = Whh + (size_t)r * H;
                    for (int k = 0; k < H; ++k) a += wh[k] * h[k];
                    g[r] = a;
                }
                size_t sidx = (((size_t)l*B + b)*T + t);
                for (int j = 0; j < H; ++j) {
                    float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[H+j]);
                    float gg = tanhf(g[2*H+j]), og = sigmoidf_(g[3*H+j]);
                    c[j] = fg * c[j] + ig * gg;
                    float hn = og * tanhf(c[j]);
                    out[(size_t)t*H + j] = hn;
                    if (gates) { float *gp = gates + sidx*4*H; gp[j]=ig; gp[H+j]=fg; gp[2*H+j]=gg; gp[3*H+j]=og; }
                    if (cseq) cseq[sidx*H + j] = c[j];
                    if (hseq) hseq[sidx*H + j] = hn;
                }
                for (int j = 0; j < H; ++j) h[j] = out[(size_t)t*H + j];
            }
            /* residual (extension) then inter-layer dropout -> next layer's input */
            for (int t = 0; t < T; ++t)
                for (int j = 0; j < H; ++j) {
                    float v = out[(size_t)t*H + j];
                    if (residual && l >= 1) v += cur[(size_t)t*H + j];
                    if (l < L-1 && drop_lstm) v *= drop_lstm[(((size_t)l*B + b)*T + t)*H + j];
                    out[(size_t)t*H + j] = v;
                }
            memcpy(cur, out, sizeof(float) * (size_t)T * H);
            I = H;
        }
        /* ---- attention pooling over time: lstm_eeg_model.py:35-37 ---- */
        const float *aw = params + lo.attn_w; float ab = params[lo.attn_b];
        float mx = -INFINITY;
        for (int t = 0; t < T; ++t) {
            float s = ab;
            for (int j = 0; j < H; ++j) s += cur[(size_t)t*H + j] * aw[j];
            sc[t] = s; if (s > mx) mx = s;
        }
        float den = 0.f;
        for (int t = 0; t < T; ++t) { sc[t] = expf(sc[t] - mx); den += sc[t]; }
        for (int j = 0; j < H; ++j) pl[j] = 0.f;
        for (int t = 0; t < T; ++t) {
            float a = sc[t] / den;
            if (alpha) alpha[(size_t)b*T + t] = a;
            for (int j = 0; j < H; ++j) pl[j] += a * cur[(size_t)t*H + j];
        }
        if (pooled) memcpy(pooled + (size_t)b*H, pl, sizeof(float)*H);
        /* ---- LayerNorm(H), eps=1e-5, biased variance: lstm_eeg_model.py:38 ---- */
        float mu = 0.f; for (int j = 0; j < H; ++j) mu += pl[j]; mu /= (float)H;
        float var = 0.f; for (int j = 0; j < H; ++j) { float d = pl[j]-mu; var += d*d; } var /= (float)H;
        float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int j = 0; j < H; ++j) lnv[j] = (pl[j]-mu)*rstd*params[lo.ln_w+j] + params[lo.ln_b+j];
        if (ln_out) memcpy(ln_out + (size_t)b*H, lnv, sizeof(float)*H);
        /* ---- fc: Linear(H,F) -> RReLU -> Dropout -> Linear(F,K): lstm_eeg_model.py:25-30,39 ---- */
        for (int f = 0; f < F; ++f) {
            float a = params[lo.fc0_b + f];
            const float *w = params + lo.fc0_w + (size_t)f*H;
            for (int j = 0; j < H; ++j) a += w[j] * lnv[j];
            if (fc0_pre) fc0_pre[(size_t)b*F + f] = a;
            float sl = rrelu_slope ? rrelu_slope[(size_t)b*F + f] : eval_slope;
            float v = a >= 0.f ? a : a * sl;
            if (drop_head) v *= drop_head[(size_t)b*F + f];
            z[f] = v;
            if (fc0_act) fc0_act[(size_t)b*F + f] = v;
        }
        float lmx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            float a = params[lo.fc3_b + k];
            const float *w = params + lo.fc3_w + (size_t)k*F;
            for (int f = 0; f < F; ++f) a += w[f] * z[f];
            lg[k] = a; if (a > lmx) lmx = a;
            if (logits) logits[(size_t)b*K + k] = a;
        }
        /* ---- softmax over classes: lstm_eeg_model.py:97 ---- */
        if (probs) {
            float d = 0.f;
            for (int k = 0; k < K; ++k) { lg[k] = expf(lg[k]-lmx); d += lg[k]; }
            for (int k = 0; k < K; ++k) probs[(size_t)b*K + k] = lg[k] / d;
        }
    }
    free(cur); free(out); free(g); free(h); free(c); free(sc); free(pl); free(lnv); free(z); free(lg);
    return 0;
}

/* mean cross-entropy over the batch; dlogits = (softmax - onehot) * scale  (scale normally 1/B_global) */
int nsd_oracle_ce_loss(int B, int K, const float *logits, const int32_t *labels, float scale,
                       float *loss_sum, float *dlogits)
{
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *lg = logits + (size_t)b*K;
        int y = labels[b]; if (y < 0 || y >= K) return -1;
        float mx = lg[0]; for (int k = 1; k < K; ++k) if (lg[k] > mx) mx = lg[k];
        double d = 0.0; for (int k = 0; k < K; ++k) d += exp((double)lg[k] - mx);
        tot += -(((double)lg[y] - mx) - log(d));
        if (dlogits) for (int k = 0; k < K; ++k)
            dlogits[(size_t)b*K + k] = (float)((exp((double)lg[k]-mx)/d - (k==y ? 1.0 : 0.0)) * scale);
    }
    if (loss_sum) *loss_sum = (float)tot;   /* SUM of per-trial losses; caller divides */
    return 0;
}

/*
 * Backward.  Needs the saves of a forward run on the same inputs (hseq, cseq,
 * gates, alpha, pooled, fc0_pre) plus the same masks.  grads[P] is
 * ACCUMULATED INTO (caller zeroes it).  dx[B,T,C] optional.
 */
int nsd_oracle_backward(int B, int T, int C, int H, int L, int K, int F,
                        const float *params, const float *x,
                        const float *drop_lstm, const float *rrelu_slope, const float *drop_head,
                        int residual,
                        const float *hseq, const float *cseq, const float *gates,
                        const float *alpha, const float *pooled, const float *fc0_pre,
                        const float *dlogits, float *grads, float *dx)
{
    if (L < 1 || L > NSD_MAX_LAYERS) return -1;
    nsd_layout lo; layout(C, H, L, K, F, &lo);
    const float eval_slope = nsd_oracle_rrelu_eval_slope();
    const long P = lo.total;
    double *G = (double*)calloc((size_t)P, sizeof(double));
    float *dout = (float*)malloc(sizeof(float) * (size_t)T * H);   /* grad wrt current layer's output seq */
    float *din  = (float*)malloc(sizeof(float) * (size_t)T * (H > C ? H : C));
    float *lin  = (float*)malloc(sizeof(float) * (size_t)T * (H > C ? H : C)); /* layer input seq */
    float *top  = (float*)malloc(sizeof(float) * (size_t)T * H);   /* top-of-stack output seq */
    float *da   = (float*)malloc(sizeof(float) * 4 * H);
    float *dc   = (float*)malloc(sizeof(float) * H);
    float *dhn  = (float*)malloc(sizeof(float) * H);
    float *zact = (float*)malloc(sizeof(float) * F);
    float *dz   = (float*)malloc(sizeof(float) * F);
    float *lnv  = (float*)malloc(sizeof(float) * H);
    float *xhat = (float*)malloc(sizeof(float) * H);
    float *dln  = (float*)malloc(sizeof(float) * H);
    float *dp   = (float*)malloc(sizeof(float) * H);
    float *dal  = (float*)malloc(sizeof(float) * T);
    if (!G||!dout||!din||!lin||!top||!da||!dc||!dhn||!zact||!dz||!lnv||!xhat||!dln||!dp||!dal) return -2;

#define HS(l,b,t) (hseq  + ((((size_t)(l)*B + (b))*T + (t))*H))
#define CS(l,b,t) (cseq  + ((((size_t)(l)*B + (b))*T + (t))*H))
#define GS(l,b,t) (gates + ((((size_t)(l)*B + (b))*T + (t))*4*H))
#define DM(l,b,t) (drop_lstm + ((((size_t)(l)*B + (b))*T + (t))*H))

    for (int b = 0; b < B; ++b) {
        /* reconstruct the sequence that feeds attention (top-layer output incl. residual chain) */
        /* layer output o_l = h_l (+ in_l if residual, l>=1); in_{l+1} = o_l * mask_l */
        /* compute in_l for all layers lazily below; first the top output: */
        {
            /* iterate layers forward to build 'top' */
            for (int t = 0; t < T; ++t) for (int k = 0; k < C; ++k) lin[(size_t)t*C+k] = x[((size_t)b*T+t)*C+k];
            for (int l = 0; l < L; ++l) {
                for (int t = 0; t < T; ++t) for (int j = 0; j < H; ++j) {
                    float v = HS(l,b,t)[j];
                    if (residual && l >= 1) v += lin[(size_t)t*H + j];
                    if (l < L-1 && drop_lstm) v *= DM(l,b,t)[j];
                    top[(size_t)t*H + j] = v;
                }
                if (l < L-1) memcpy(lin, top, sizeof(float)*(size_t)T*H);
            }
        }
        /* ---- head backward ---- */
        const float *p = pooled + (size_t)b*H;
        float mu = 0.f; for (int j = 0; j < H; ++j) mu += p[j]; mu /= (float)H;
        float var = 0.f; for (int j = 0; j < H; ++j) { float d = p[j]-mu; var += d*d; } var /= (float)H;
        float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int j = 0; j < H; ++j) { xhat[j] = (p[j]-mu)*rstd; lnv[j] = xhat[j]*params[lo.ln_w+j] + params[lo.ln_b+j]; }
        for (int f = 0; f < F; ++f) {
            float a = fc0_pre[(size_t)b*F + f];
            float sl = rrelu_slope ? rrelu_slope[(size_t)b*F + f] : eval_slope;
            float v = a >= 0.f ? a : a*sl;
            if (drop_head) v *= drop_head[(size_t)b*F + f];
            zact[f] = v;
        }
        const float *dl = dlogits + (size_t)b*K;
        for (int f = 0; f < F; ++f) dz[f] = 0.f;
        for (int k = 0; k < K; ++k) {
            G[lo.fc3_b + k] += dl[k];
            for (int f = 0; f < F; ++f) {
                G[lo.fc3_w + (size_t)k*F + f] += (double)dl[k] * zact[f];
                dz[f] += params[lo.fc3_w + (size_t)k*F + f] * dl[k];
            }
        }
        for (int f = 0; f < F; ++f) {
            float a = fc0_pre[(size_t)b*F + f];
            float sl = rrelu_slope ? rrelu_slope[(size_t)b*F + f] : eval_slope;
            float d = dz[f];
            if (drop_head) d *= drop_head[(size_t)b*F + f];
            dz[f] = a >= 0.f ? d : d*sl;
        }
        for (int j = 0; j < H; ++j) dln[j] = 0.f;
        for (int f = 0; f < F; ++f) {
            G[lo.fc0_b + f] += dz[f];
            for (int j = 0; j < H; ++j) {
                G[lo.fc0_w + (size_t)f*H + j] += (double)dz[f] * lnv[j];
                dln[j] += params[lo.fc0_w + (size_t)f*H + j] * dz[f];
            }
        }
        float m1 = 0.f, m2 = 0.f;
        for (int j = 0; j < H; ++j) {
            G[lo.ln_w + j] += (double)dln[j] * xhat[j];
            G[lo.ln_b + j] += dln[j];
            float dxh = dln[j] * params[lo.ln_w + j];
            dp[j] = dxh; m1 += dxh; m2 += dxh * xhat[j];
        }
        m1 /= (float)H; m2 /= (float)H;
        for (int j = 0; j < H; ++j) dp[j] = rstd * (dp[j] - m1 - xhat[j]*m2);
        /* attention pooling backward */
        const float *al = alpha + (size_t)b*T;
        float sdot = 0.f;
        for (int t = 0; t < T; ++t) {
            float d = 0.f; for (int j = 0; j < H; ++j) d += dp[j] * top[(size_t)t*H + j];
            dal[t] = d; sdot += al[t]*d;
        }
        for (int t = 0; t < T; ++t) {
            float ds = al[t] * (dal[t] - sdot);
            G[lo.attn_b] += ds;
            for (int j = 0; j < H; ++j) {
                G[lo.attn_w + j] += (double)ds * top[(size_t)t*H + j];
                dout[(size_t)t*H + j] = al[t]*dp[j] + ds*params[lo.attn_w + j];
            }
        }
        /* ---- LSTM stack backward (BPTT) ---- */
        for (int l = L-1; l >= 0; --l) {
            int I = l == 0 ? C : H;
            /* rebuild this layer's input sequence */
            if (l == 0) {
                for (int t = 0; t < T; ++t) for (int k = 0; k < C; ++k) lin[(size_t)t*C+k] = x[((size_t)b*T+t)*C+k];
            } else {
                /* in_l = o_{l-1}*mask_{l-1}; o_{l-1} = h_{l-1} (+ in_{l-1} if residual & l-1>=1) : recompute chain */
                for (int t = 0; t < T; ++t) for (int k = 0; k < C; ++k) lin[(size_t)t*C+k] = x[((size_t)b*T+t)*C+k];
                for (int q = 0; q < l; ++q) {
                    for (int t = T-1; t >= 0; --t) for (int j = H-1; j >= 0; --j) {
                        float v = HS(q,b,t)[j];
                        if (residual && q >= 1) v += lin[(size_t)t*H + j];
                        if (drop_lstm) v *= DM(q,b,t)[j];
                        top[(size_t)t*H + j] = v;   /* reuse 'top' as scratch: attention part is done */
                    }
                    memcpy(lin, top, sizeof(float)*(size_t)T*H);
                }
            }
            const float *Wih = params + lo.w_ih[l], *Whh = params + lo.w_hh[l];
            for (int j = 0; j < H; ++j) { dhn[j] = 0.f; dc[j] = 0.f; }
            for (long k = 0; k < (long)T*I; ++k) din[k] = 0.f;
            for (int t = T-1; t >= 0; --t) {
                const float *gt = GS(l,b,t);
                const float *ct = CS(l,b,t);
                for (int j = 0; j < H; ++j) {
                    float dht = dout[(size_t)t*H + j] + dhn[j];
                    float ig = gt[j], fg = gt[H+j], gg = gt[2*H+j], og = gt[3*H+j];
                    float tc = tanhf(ct[j]);
                    float cprev = t > 0 ? CS(l,b,t-1)[j] : 0.f;
                    float dct = dc[j] + dht * og * (1.f - tc*tc);
                    da[j]      = dct * gg * ig * (1.f - ig);
                    da[H+j]    = dct * cprev * fg * (1.f - fg);
                    da[2*H+j]  = dct * ig * (1.f - gg*gg);
                    da[3*H+j]  = dht * tc * og * (1.f - og);
                    dc[j] = dct * fg;
                }
                const float *in = lin + (size_t)t*I;
                const float *hp = t > 0 ? HS(l,b,t-1) : NULL;
                for (int j = 0; j < H; ++j) dhn[j] = 0.f;
                for (int r = 0; r < 4*H; ++r) {
                    float d = da[r];
                    G[lo.b_ih[l] + r] += d; G[lo.b_hh[l] + r] += d;
                    for (int k = 0; k < I; ++k) {
                        G[lo.w_ih[l] + (size_t)r*I + k] += (double)d * in[k];
                        din[(size_t)t*I + k] += Wih[(size_t)r*I + k] * d;
                    }
                    for (int k = 0; k < H; ++k) {
                        if (hp) G[lo.w_hh[l] + (size_t)r*H + k] += (double)d * hp[k];
                        dhn[k] += Whh[(size_t)r*H + k] * d;
                    }
                }
            }
            if (l == 0) {
                if (dx) for (int k = 0; k < T*C; ++k) dx[(size_t)b*T*C + k] = din[k];
            } else {
                /* d o_{l-1} = din * mask_{l-1}  (+ residual passthrough: dout_l also flows to in_l) */
                for (int t = 0; t < T; ++t) for (int j = 0; j < H; ++j) {
                    float d = din[(size_t)t*H + j];
                    if (residual) d += dout[(size_t)t*H + j];
                    if (drop_lstm) d *= DM(l-1,b,t)[j];
                    top[(size_t)t*H + j] = d;
                }
                memcpy(dout, top, sizeof(float)*(size_t)T*H);
            }
        }
    }
    for (long i = 0; i < P; ++i) grads[i] += (float)G[i];
#undef HS
#undef CS
#undef GS
#undef DM
    free(G); free(dout); free(din); free(lin); free(top); free(da); free(dc); free(dhn);
    free(zact); free(dz); free(lnv); free(xhat); free(dln); free(dp); free(dal);
    return 0;
}

/* per-channel z-score over time, app.py:166-170:  (x - mean_T) / (std_T(ddof=0) + 1e-6) */
int nsd_oracle_zscore(int B, int T, int C, const float *x, float *y)
{
    for (int b = 0; b < B; ++b)
        for (int ch = 0; ch < C; ++ch) {
            double s = 0.0;
            for (int t = 0; t < T; ++t) s += x[((size_t)b*T + t)*C + ch];
            double mu = s / T, v = 0.0;
            for (int t = 0; t < T; ++t) { double d = x[((size_t)b*T + t)*C + ch] - mu; v += d*d; }
            double sd = sqrt(v / T) + 1e-6;
            for (int t = 0; t < T; ++t)
                y[((size_t)b*T + t)*C + ch] = (float)((x[((size_t)b*T + t)*C + ch] - mu) / sd);
        }
    return 0;
}

/* torch.optim.Adam (no amsgrad, weight_decay added to grad if non-zero), step counted from 1 */
int nsd_oracle_adam(long n, float *p, const float *g, float *m, float *v,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step)
{
    double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    float step_size = (float)(lr / bc1);
    float rbc2 = (float)(1.0 / sqrt(bc2));
    for (long i = 0; i < n; ++i) {
        float gi = g[i] + weight_decay * p[i];
        m[i] = beta1 * m[i] + (1.f - beta1) * gi;
        v[i] = beta2 * v[i] + (1.f - beta2) * gi * gi;
        float denom = sqrtf(v[i]) * rbc2 + eps;
        p[i] -= step_size * (m[i] / denom);
    }
    return 0;
}

/*
 * Counter-based dropout stream shared with the HIP path (the trainer's own RNG;
 * the reference's training RNG is torch's and is not reproducible across
 * backends).  keep = hash(seed, stream, index) >= p * 2^32.
 */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
uint32_t nsd_oracle_rand_u32(uint64_t seed, uint32_t stream, uint64_t index) {
    uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    uint32_t h = mix32(lo ^ s0);
    h = mix32(h + 0x9e3779b9U * (stream + 1u) + hi);
    h = mix32(h ^ s1);
    return h;
}
/* multiplier mask: 0 or 1/(1-p) */
int nsd_oracle_dropout_mask(uint64_t seed, uint32_t stream, float p, long n, float *mask) {
    uint32_t thr = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
    float keep = 1.0f / (1.0f - p);
    for (long i = 0; i < n; ++i) mask[i] = nsd_oracle_rand_u32(seed, stream, (uint64_t)i) >= thr ? keep : 0.f;
    return 0;
}
/* RReLU train-mode slopes ~ U(lower, upper) from the same stream */
int nsd_oracle_rrelu_noise(uint64_t seed, uint32_t stream, long n, float *slope) {
    const float lower = 0.125f, upper = (float)(1.0/3.0);
    for (long i = 0; i < n; ++i) {
        float u = (float)(nsd_oracle_rand_u32(seed, stream, (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
        slope[i] = lower + (upper - lower) * u;
    }
    return 0;
}
