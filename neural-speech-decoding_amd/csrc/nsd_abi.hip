// nsd_abi.hip -- extern "C" entry points of libnsd_hip.so (declared in include/nsd.h).
// Argument validation, parameter / workspace layout, kernel dispatch.  No allocation, no synchronisation.
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include "nsd_args.h"
#include "nsd_bf16.h"

// ---- error text -------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void nsd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int nsd_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        else
            cus = 256;   // MI355X
    }
    return cus;
}

ParamLayout nsd_make_layout(int C, int H, int L, int K, int F) {
    ParamLayout o;
    memset(&o, 0, sizeof(o));
    int64_t p = 0;
    for (int l = 0; l < L; ++l) {
        const int I = l == 0 ? C : H;
        o.w_ih[l] = p; p += 4LL * H * I;
        o.w_hh[l] = p; p += 4LL * H * H;
        o.b_ih[l] = p; p += 4LL * H;
        o.b_hh[l] = p; p += 4LL * H;
    }
    o.lstm_total = p;
    o.ln_w = p; p += H;   o.ln_b = p; p += H;
    o.attn_w = p; p += H; o.attn_b = p; p += 1;
    o.fc0_w = p; p += (int64_t)F * H; o.fc0_b = p; p += F;
    o.fc3_w = p; p += (int64_t)K * F; o.fc3_b = p; p += K;
    o.total = p;
    return o;
}

static int check_model(int C, int H, int L, int K, int F) {
    if (C < 1 || H < 1 || L < 1 || L > NSD_MAX_LAYERS || K < 1 || F < 1) {
        nsd_set_error("bad model dims C=%d H=%d L=%d K=%d F=%d", C, H, L, K, F);
        return NSD_E_INVALID;
    }
    return NSD_OK;
}
int nsd_check_dims(const nsd_dims *d) {
    if (!d) { nsd_set_error("dims is NULL"); return NSD_E_INVALID; }
    if (d->B < 0 || d->T < 1) { nsd_set_error("bad batch dims B=%d T=%d", d->B, d->T); return NSD_E_INVALID; }
    return check_model(d->C, d->H, d->L, d->K, d->F);
}

static inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

static int fast_path_ok(const nsd_dims *d) {
    // H = 64: the first-generation fused kernels win for small batches, the batched MFMA path from ~400 trials on
    // (measured B=256: 6.9 vs 8.2 ms/step, B=1024: 27.5 vs 13.4 ms/step)
    if (d->H == 64 && d->B >= 384 && nsd_lstm_batched_ok(d, true)) return 0;
    return d->L == 2 && (d->H == 32 || d->H == 48 || d->H == 64) && d->C <= 8;
}

static nsd_ws_layout make_ws(const nsd_dims *d, bool have_device) {
    nsd_ws_layout w;
    memset(&w, 0, sizeof(w));
    const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
    const int64_t B = d->B, T = d->T, H = d->H, L = d->L, F = d->F;
    int64_t p = 0;
    w.hseq = p;    p = align4(p + L * B * T * H);
    w.cseq = p;    p = align4(p + L * B * T * H);
    w.gact = p;    p = align4(p + L * B * T * H * 4);
    w.inseq = p;   p = align4(p + (L - 1) * B * T * H);
    w.top = p;     p = align4(p + B * T * H);
    w.alpha = p;   p = align4(p + B * T);
    w.pooled = p;  p = align4(p + B * H);
    w.fc0_pre = p; p = align4(p + B * F);
    w.dscore = p;  p = align4(p + B * T);
    w.dpooled = p; p = align4(p + B * H);
    w.loss = p;    p = align4(p + B);
    w.adpack = p;  p = align4(p + B * T * 4);
    // LSTM slabs: one per backward workgroup (<= #CUs); head slabs: one per trial, stored behind them.
    // Without a device (symbol / layout checks on CPU) assume the MI355X's 256 CUs.
    const bool fast = fast_path_ok(d) != 0;
    int64_t nsl = B < 256 ? B : 256;
    if (have_device) nsl = nsd_lstm2_bwd_grid((int)B);
    if (nsl < 1 || !fast) nsl = 1;       // generic path: the weight-gradient GEMMs write one slab
    w.n_slabs = nsl;
    w.slabs = p;   p = align4(p + nsl * align4(pl.lstm_total));
    w.hslabs = p;  p = align4(p + B * (pl.total - pl.lstm_total));
    w.da_seq = p;  if (!fast) p = align4(p + L * B * T * 4 * H);      // one per layer: the batched path keeps all layers in flight
    // din: two [B,T,H] ping-pong buffers + the batched path's per-step state [L,3,B,H] and split-K partials [4][4H x max(C,H)]
    w.din = p;     if (!fast) p = align4(p + 2 * B * T * H + L * 3 * B * H + 8 * 4 * H * (H > d->C ? H : (int64_t)d->C));
    w.total = p;
    return w;
}

// Diagnostic build only (make prof -> libnsd_hip_prof.so, or -DNSD_ABLATE_HOOKS=1): the cycle-stamp buffer and the
// NSD_ABLATE timing switches.  The shipped library has neither the symbol nor the getenv.
#if NSD_PROFILE || NSD_ABLATE_HOOKS
static long long *g_dbg = nullptr;
static int ablate_mask() { const char *e = getenv("NSD_ABLATE"); return e ? atoi(e) : 0; }     // (read per launch: tools/kbench.py sweeps it)
extern "C" int nsd_debug_profile_buffer(void *p) { g_dbg = (long long *)p; return NSD_OK; }
#else
static long long *const g_dbg = nullptr;
static int ablate_mask() { return 0; }
#endif

// workspace size check shared by every entry point that touches the training workspace
static int check_ws(const nsd_dims *d, const void *workspace, int64_t workspace_bytes, const char *who, nsd_ws_layout *w) {
    if (!workspace) { nsd_set_error("%s: workspace is NULL", who); return NSD_E_INVALID; }
    *w = make_ws(d, true);
    const int64_t need = w->total * (int64_t)sizeof(float);
    if (workspace_bytes < need) {
        nsd_set_error("%s: workspace of %lld bytes is smaller than nsd_workspace_bytes() = %lld", who, (long long)workspace_bytes, (long long)need);
        return NSD_E_WORKSPACE;
    }
    return NSD_OK;
}

static bool device_present() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}


extern "C" {

int nsd_version(void) { return NSD_VERSION; }
const char *nsd_last_error(void) { return g_err; }

int64_t nsd_param_count(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F) {
    if (check_model(C, H, L, K, F) != NSD_OK) return NSD_E_INVALID;
    return nsd_make_layout(C, H, L, K, F).total;
}

int nsd_param_layout(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F, int64_t *offsets) {
    if (check_model(C, H, L, K, F) != NSD_OK || !offsets) return NSD_E_INVALID;
    const ParamLayout o = nsd_make_layout(C, H, L, K, F);
    for (int l = 0; l < L; ++l) {
        offsets[4 * l + 0] = o.w_ih[l]; offsets[4 * l + 1] = o.w_hh[l];
        offsets[4 * l + 2] = o.b_ih[l]; offsets[4 * l + 3] = o.b_hh[l];
    }
    int64_t *q = offsets + 4 * L;
    q[0] = o.ln_w; q[1] = o.ln_b; q[2] = o.attn_w; q[3] = o.attn_b;
    q[4] = o.fc0_w; q[5] = o.fc0_b; q[6] = o.fc3_w; q[7] = o.fc3_b;
    return NSD_OK;
}

int64_t nsd_workspace_bytes(const nsd_dims *d, nsd_ws_layout *layout_out) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    const nsd_ws_layout w = make_ws(d, device_present());
    if (layout_out) *layout_out = w;
    return w.total * (int64_t)sizeof(float);
}

int nsd_fast_path(const nsd_dims *d) {
    if (nsd_check_dims(d) != NSD_OK) return 0;
    return fast_path_ok(d);
}

int nsd_zscore_fwd(const float *x, float *y, int32_t B, int32_t T, int32_t C, void *stream) {
    if (!x || !y || B < 0) { nsd_set_error("zscore: null pointer or B<0"); return NSD_E_INVALID; }
    return nsd_zscore_launch(x, y, B, T, C, (hipStream_t)stream);
}

// ---- shared argument builders -----------------------------------------------------------------------------
static int build_lstm_fwd(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm,
                          uint32_t flags, float *ws, const nsd_ws_layout &w, bool train, float *top_only,
                          Lstm2FwdArgs *out) {
    const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
    Lstm2FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.w_ih0 = params + pl.w_ih[0]; a.w_hh0 = params + pl.w_hh[0]; a.b_ih0 = params + pl.b_ih[0]; a.b_hh0 = params + pl.b_hh[0];
    a.w_ih1 = params + pl.w_ih[1]; a.w_hh1 = params + pl.w_hh[1]; a.b_ih1 = params + pl.b_ih[1]; a.b_hh1 = params + pl.b_hh[1];
    a.mask = drop_lstm;
    a.dbg = g_dbg;
    a.B = d->B; a.T = d->T; a.C = d->C;
    a.residual = (flags & NSD_FLAG_RESIDUAL) ? 1 : 0;
    a.ablate = ablate_mask();
    const int64_t BTH = (int64_t)d->B * d->T * d->H;
    if (train) {
        a.hseq0 = ws + w.hseq; a.hseq1 = ws + w.hseq + BTH;
        a.cseq0 = ws + w.cseq; a.cseq1 = ws + w.cseq + BTH;
        a.gact0 = ws + w.gact; a.gact1 = ws + w.gact + 4 * BTH;
        a.inseq = ws + w.inseq;
        a.top = ws + w.top;
    } else {
        a.top = top_only;
    }
    *out = a;
    return NSD_OK;
}

static HeadArgs build_head(const nsd_dims *d, const float *params) {
    const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
    HeadArgs h;
    memset(&h, 0, sizeof(h));
    h.ln_w = params + pl.ln_w; h.ln_b = params + pl.ln_b; h.attn_w = params + pl.attn_w; h.attn_b = params + pl.attn_b;
    h.fc0_w = params + pl.fc0_w; h.fc0_b = params + pl.fc0_b; h.fc3_w = params + pl.fc3_w; h.fc3_b = params + pl.fc3_b;
    h.eval_slope = (float)((0.125 + 1.0 / 3.0) / 2.0);   // nn.RReLU eval slope, lstm_eeg_model.py:27
    h.o_ln_w = pl.ln_w - pl.lstm_total; h.o_ln_b = pl.ln_b - pl.lstm_total;
    h.o_attn_w = pl.attn_w - pl.lstm_total; h.o_attn_b = pl.attn_b - pl.lstm_total;
    h.o_fc0_w = pl.fc0_w - pl.lstm_total; h.o_fc0_b = pl.fc0_b - pl.lstm_total;
    h.o_fc3_w = pl.fc3_w - pl.lstm_total; h.o_fc3_b = pl.fc3_b - pl.lstm_total;
    h.Ph = pl.total - pl.lstm_total;
    h.B = d->B; h.T = d->T; h.H = d->H; h.F = d->F; h.K = d->K;
    return h;
}

#define REQUIRE_FAST(d, name)                                                                              \
    do {                                                                                                   \
        if (!fast_path_ok(d)) {                                                                            \
            nsd_set_error("%s: dims C=%d H=%d L=%d not covered yet (fast path: L==2, H in {32,48,64}, C<=8)", \
                          name, (d)->C, (d)->H, (d)->L);                                                   \
            return NSD_E_INVALID;                                                                          \
        }                                                                                                  \
    } while (0)

static bool fused_train_shape(const nsd_dims *d) {
    return fast_path_ok(d) && d->H == 48 && nsd_lstm2_fwd48_head_train_fits(d->T, d->F, d->K);
}
static int make_rng(const nsd_rng *r, RngArgs *out) {
    if (!r) { nsd_set_error("rng: null pointer"); return NSD_E_INVALID; }
    if (!(r->p_lstm >= 0.f && r->p_lstm < 1.f) || !(r->p_head >= 0.f && r->p_head < 1.f)) { nsd_set_error("rng: p out of [0,1)"); return NSD_E_INVALID; }
    out->seed = r->seed; out->base = r->base_stream;
    out->thr_lstm = nsd_drop_threshold(r->p_lstm); out->thr_head = nsd_drop_threshold(r->p_head);
    out->keep_lstm = 1.0f / (1.0f - r->p_lstm); out->keep_head = 1.0f / (1.0f - r->p_head);
    out->on = 1;
    return NSD_OK;
}

int64_t nsd_infer_scratch_bytes(const nsd_dims *d) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (fast_path_ok(d)) return align4((int64_t)d->B * d->T * d->H) * (int64_t)sizeof(float);
    // two [B,T,H] ping-pong buffers + the batched path's cell-state ping-pong [L][2][B,H]
    return (2 * align4((int64_t)d->B * d->T * d->H) + align4(2 * (int64_t)d->L * d->B * d->H)) * (int64_t)sizeof(float);
}

int nsd_infer(const nsd_dims *d, const float *params, const float *x, uint32_t flags, float *logits, float *probs,
              void *scratch, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !x || !logits || !scratch) { nsd_set_error("infer: null pointer"); return NSD_E_INVALID; }
    if (d->B == 0) return NSD_OK;
    int rc;
    if (fast_path_ok(d)) {
        nsd_ws_layout w;
        memset(&w, 0, sizeof(w));
        Lstm2FwdArgs a;
        build_lstm_fwd(d, params, x, nullptr, flags, nullptr, w, false, (float *)scratch, &a);
        if (d->H == 48 && d->F <= 64 && d->K <= 64) {
            // single launch: attention pooling (online softmax), LayerNorm, dense head and class softmax run in the
            // LSTM kernel's tail wave; nothing but x, the parameters and the [B,K] outputs touches HBM
            const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
            a.top = nullptr;
            a.attn_w = params + pl.attn_w; a.attn_b = params + pl.attn_b; a.ln_w = params + pl.ln_w; a.ln_b = params + pl.ln_b;
            a.fc0_w = params + pl.fc0_w; a.fc0_b = params + pl.fc0_b; a.fc3_w = params + pl.fc3_w; a.fc3_b = params + pl.fc3_b;
            a.eval_slope = (float)((0.125 + 1.0 / 3.0) / 2.0);
            a.logits_out = logits; a.probs_out = probs; a.K = d->K; a.F = d->F;
            return nsd_lstm2_fwd_launch(a, d->H, (hipStream_t)stream);
        }
        rc = nsd_lstm2_fwd_launch(a, d->H, (hipStream_t)stream);
    } else {
        const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
        float *top = (float *)scratch;
        const int64_t bth = align4((int64_t)d->B * d->T * d->H);
        if (nsd_lstm_batched_ok(d, false) && !(flags & NSD_FLAG_RESIDUAL))
            rc = nsd_lstm_batched_infer(d, pl, params, x, top, top + bth, top + 2 * bth, (flags & NSD_FLAG_BF16) != 0, (hipStream_t)stream);
        else
            rc = nsd_lstm_generic_fwd(d, pl, params, x, nullptr, (flags & NSD_FLAG_RESIDUAL) ? 1 : 0, nullptr, nullptr, nullptr, nullptr,
                                      top, top + bth, (hipStream_t)stream);
    }
    if (rc != NSD_OK) return rc;
    HeadArgs h = build_head(d, params);
    h.top = (const float *)scratch;
    h.logits = logits; h.probs = probs;
    return nsd_head_launch(h, false, (hipStream_t)stream);
}

int nsd_lstm_fwd(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm, uint32_t flags,
                 float *workspace, int64_t workspace_bytes, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !x || !workspace) { nsd_set_error("lstm_fwd: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "lstm_fwd", &w)) return rc;
    if (d->B == 0) return NSD_OK;
    if (!fast_path_ok(d)) {
        const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
        if (nsd_lstm_batched_ok(d, true))        // large H: per-step batched gate GEMM on the matrix pipe
            return nsd_lstm_batched_fwd(d, pl, params, x, drop_lstm, (flags & NSD_FLAG_RESIDUAL) ? 1 : 0, workspace + w.hseq,
                                        workspace + w.cseq, workspace + w.gact, workspace + w.inseq, workspace + w.top,
                                        (flags & NSD_FLAG_BF16) != 0, (hipStream_t)stream);
        return nsd_lstm_generic_fwd(d, pl, params, x, drop_lstm, (flags & NSD_FLAG_RESIDUAL) ? 1 : 0, workspace + w.hseq,
                                    workspace + w.cseq, workspace + w.gact, workspace + w.inseq, workspace + w.top, nullptr,
                                    (hipStream_t)stream);
    }
    Lstm2FwdArgs a;
    build_lstm_fwd(d, params, x, drop_lstm, flags, workspace, w, true, nullptr, &a);
    return nsd_lstm2_fwd_launch(a, d->H, (hipStream_t)stream);
}

int nsd_head_fwd(const nsd_dims *d, const float *params, const float *rrelu_slope, const float *drop_head,
                 float *workspace, int64_t workspace_bytes, float *logits, float *probs, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !workspace || !logits) { nsd_set_error("head_fwd: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "head_fwd", &w)) return rc;
    if (d->B == 0) return NSD_OK;
    HeadArgs h = build_head(d, params);
    h.top = workspace + w.top;
    h.rrelu_slope = rrelu_slope; h.drop_head = drop_head;
    h.logits = logits; h.probs = probs;
    h.alpha = workspace + w.alpha; h.pooled = workspace + w.pooled; h.fc0_pre = workspace + w.fc0_pre;
    return nsd_head_launch(h, false, (hipStream_t)stream);
}

int nsd_head_bwd(const nsd_dims *d, const float *params, const float *rrelu_slope, const float *drop_head,
                 const float *logits, const float *dlogits, const int32_t *labels, float scale, float *workspace,
                 int64_t workspace_bytes, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !workspace) { nsd_set_error("head_bwd: null pointer"); return NSD_E_INVALID; }
    if (!dlogits && !(labels && logits)) {
        nsd_set_error("head_bwd: need dlogits, or labels together with logits");
        return NSD_E_INVALID;
    }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "head_bwd", &w)) return rc;
    if (d->B == 0) return NSD_OK;
    HeadArgs h = build_head(d, params);
    h.top = workspace + w.top;
    h.rrelu_slope = rrelu_slope; h.drop_head = drop_head;
    h.alpha = workspace + w.alpha; h.pooled = workspace + w.pooled; h.fc0_pre = workspace + w.fc0_pre;
    h.logits_in = logits; h.dlogits = dlogits; h.labels = labels; h.scale = scale;
    h.loss = workspace + w.loss; h.dscore = workspace + w.dscore; h.dpooled = workspace + w.dpooled;
    h.hslabs = workspace + w.hslabs;
    h.adpack = workspace + w.adpack;
    return nsd_head_launch(h, true, (hipStream_t)stream);
}

int nsd_head_train(const nsd_dims *d, const float *params, const float *rrelu_slope, const float *drop_head,
                   const int32_t *labels, float scale, float *workspace, int64_t workspace_bytes, float *logits, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !workspace || !logits || !labels) { nsd_set_error("head_train: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "head_train", &w)) return rc;
    if (d->B == 0) return NSD_OK;
    HeadArgs h = build_head(d, params);
    h.top = workspace + w.top;
    h.rrelu_slope = rrelu_slope; h.drop_head = drop_head;
    h.alpha = workspace + w.alpha; h.pooled = workspace + w.pooled; h.fc0_pre = workspace + w.fc0_pre;
    h.logits = logits; h.logits_in = logits; h.labels = labels; h.scale = scale;
    h.loss = workspace + w.loss; h.dscore = workspace + w.dscore; h.dpooled = workspace + w.dpooled;
    h.hslabs = workspace + w.hslabs; h.adpack = workspace + w.adpack;
    const int rc = nsd_head_train_launch(h, (hipStream_t)stream);
    if (rc != 0) return rc < 0 ? rc : NSD_OK;
    // shape does not fit the fused kernel's LDS budget: two passes
    const int rc2 = nsd_head_launch(h, false, (hipStream_t)stream);
    if (rc2 != NSD_OK) return rc2;
    return nsd_head_launch(h, true, (hipStream_t)stream);
}

int nsd_rng_path(const nsd_dims *d) {
    if (nsd_check_dims(d) != NSD_OK) return 0;
    return fused_train_shape(d) ? 1 : 0;
}

static int lstm_head_train_impl(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm,
                                const float *rrelu_slope, const float *drop_head, const RngArgs *rng, const int32_t *labels,
                                float scale, uint32_t flags, float *workspace, int64_t workspace_bytes, float *logits, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !x || !workspace || !logits || !labels) { nsd_set_error("lstm_head_train: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "lstm_head_train", &w)) return rc;
    if (d->B == 0) return NSD_OK;
    if (!fused_train_shape(d)) {
        if (rng) { nsd_set_error("lstm_head_train_rng: shape outside the single-launch path (nsd_rng_path() == 0)"); return NSD_E_INVALID; }
        // shapes outside the fused kernel: the two launches it replaces
        const int rc = nsd_lstm_fwd(d, params, x, drop_lstm, flags, workspace, workspace_bytes, stream);
        if (rc != NSD_OK) return rc;
        return nsd_head_train(d, params, rrelu_slope, drop_head, labels, scale, workspace, workspace_bytes, logits, stream);
    }
    Lstm2FwdArgs a;
    build_lstm_fwd(d, params, x, drop_lstm, flags, workspace, w, true, nullptr, &a);
    const HeadArgs h = build_head(d, params);
    a.attn_w = h.attn_w; a.attn_b = h.attn_b; a.ln_w = h.ln_w; a.ln_b = h.ln_b;
    a.fc0_w = h.fc0_w; a.fc0_b = h.fc0_b; a.fc3_w = h.fc3_w; a.fc3_b = h.fc3_b;
    a.eval_slope = h.eval_slope; a.K = d->K; a.F = d->F;
    a.head_train = 1;
    if (!a.residual) a.top = nullptr;      // top == layer-1 h: the kernel's tail reads hseq1, the saver skips the duplicate
    a.labels = labels; a.rrelu_slope = rrelu_slope; a.drop_head = drop_head; a.scale = scale;
    a.logits = logits; a.loss = workspace + w.loss; a.alpha = workspace + w.alpha; a.pooled = workspace + w.pooled;
    a.fc0_pre = workspace + w.fc0_pre; a.dscore = workspace + w.dscore; a.dpooled = workspace + w.dpooled;
    a.adpack = workspace + w.adpack; a.hslabs = workspace + w.hslabs;
    a.o_ln_w = h.o_ln_w; a.o_ln_b = h.o_ln_b; a.o_attn_w = h.o_attn_w; a.o_attn_b = h.o_attn_b;
    a.o_fc0_w = h.o_fc0_w; a.o_fc0_b = h.o_fc0_b; a.o_fc3_w = h.o_fc3_w; a.o_fc3_b = h.o_fc3_b; a.Ph = h.Ph;
    if (rng) a.rng = *rng;
    return nsd_lstm2_fwd_launch(a, d->H, (hipStream_t)stream);
}

int nsd_lstm_head_train(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm,
                        const float *rrelu_slope, const float *drop_head, const int32_t *labels, float scale, uint32_t flags,
                        float *workspace, int64_t workspace_bytes, float *logits, void *stream) {
    return lstm_head_train_impl(d, params, x, drop_lstm, rrelu_slope, drop_head, nullptr, labels, scale, flags, workspace, workspace_bytes, logits, stream);
}

int nsd_lstm_head_train_rng(const nsd_dims *d, const float *params, const float *x, const nsd_rng *rng, const int32_t *labels,
                            float scale, uint32_t flags, float *workspace, int64_t workspace_bytes, float *logits, void *stream) {
    RngArgs r;
    if (make_rng(rng, &r) != NSD_OK) return NSD_E_INVALID;
    return lstm_head_train_impl(d, params, x, nullptr, nullptr, nullptr, &r, labels, scale, flags, workspace, workspace_bytes, logits, stream);
}

static int lstm_bwd_impl(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm, const RngArgs *rng,
                         uint32_t flags, float *workspace, int64_t workspace_bytes, float *dx, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!params || !x || !workspace) { nsd_set_error("lstm_bwd: null pointer"); return NSD_E_INVALID; }
    // dx = dL/dx [B,T,C] (optional): H = 48 (the one-trial kernel leaves da0 in place of layer 0's saved gates: ONE backward per forward
    // then) and the generic path (its da_seq holds layer 0 last); the other paths do not form it
    const bool dx_ok = (fast_path_ok(d) && d->H == 48) || (!fast_path_ok(d) && !nsd_lstm_batched_ok(d, true));
    if (dx && !dx_ok) { nsd_set_error("lstm_bwd: dx is available for H = 48 (L = 2, C <= 8) and on the generic path only (include/nsd.h)"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "lstm_bwd", &w)) return rc;
    if (d->B == 0) return NSD_OK;
    const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
    const int64_t BTH = (int64_t)d->B * d->T * d->H;
    if (!fast_path_ok(d) && nsd_lstm_batched_ok(d, true))
        return nsd_lstm_batched_bwd(d, pl, params, x, drop_lstm, (flags & NSD_FLAG_RESIDUAL) ? 1 : 0, workspace + w.hseq,
                                    workspace + w.cseq, workspace + w.gact, workspace + w.inseq, workspace + w.alpha,
                                    workspace + w.dscore, workspace + w.dpooled, workspace + w.da_seq, workspace + w.din,
                                    workspace + w.din + BTH, workspace + w.din + 2 * BTH, workspace + w.slabs,
                                    (flags & NSD_FLAG_BF16) != 0, (hipStream_t)stream);
    if (!fast_path_ok(d)) {
        if (const int rc = nsd_lstm_generic_bwd(d, pl, params, x, drop_lstm, (flags & NSD_FLAG_RESIDUAL) ? 1 : 0, workspace + w.hseq,
                                    workspace + w.cseq, workspace + w.gact, workspace + w.inseq, workspace + w.alpha,
                                    workspace + w.dscore, workspace + w.dpooled, workspace + w.da_seq, workspace + w.din,
                                    workspace + w.din + BTH, workspace + w.slabs, (hipStream_t)stream)) return rc;
        return dx ? nsd_dx_launch(workspace + w.da_seq, params + pl.w_ih[0], dx, (long)d->B * d->T, 4 * d->H, d->C, (hipStream_t)stream) : NSD_OK;
    }
    Lstm2BwdArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.w_hh0 = params + pl.w_hh[0]; a.w_ih1 = params + pl.w_ih[1]; a.w_hh1 = params + pl.w_hh[1];
    a.attn_w = params + pl.attn_w;
    a.mask = drop_lstm;
    a.hseq0 = workspace + w.hseq; a.hseq1 = workspace + w.hseq + BTH;
    a.cseq0 = workspace + w.cseq; a.cseq1 = workspace + w.cseq + BTH;
    a.gact0 = workspace + w.gact; a.gact1 = workspace + w.gact + 4 * BTH;
    a.in1seq = workspace + w.inseq;
    a.alpha = workspace + w.alpha; a.dscore = workspace + w.dscore; a.dpooled = workspace + w.dpooled;
    a.dsc_pack = workspace + w.adpack;
    {
        const HeadArgs h = build_head(d, params);
        a.pooled = workspace + w.pooled; a.dscore_out = workspace + w.dscore; a.hslabs = workspace + w.hslabs;
        a.Ph = h.Ph; a.o_attn_w = h.o_attn_w; a.o_attn_b = h.o_attn_b;
    }
    a.dbg = g_dbg;
    a.slabs = workspace + w.slabs;
    a.slab_stride = align4(pl.lstm_total);
    a.o_w_ih0 = pl.w_ih[0]; a.o_w_hh0 = pl.w_hh[0]; a.o_b_ih0 = pl.b_ih[0]; a.o_b_hh0 = pl.b_hh[0];
    a.o_w_ih1 = pl.w_ih[1]; a.o_w_hh1 = pl.w_hh[1]; a.o_b_ih1 = pl.b_ih[1]; a.o_b_hh1 = pl.b_hh[1];
    a.B = d->B; a.T = d->T; a.C = d->C;
    a.residual = (flags & NSD_FLAG_RESIDUAL) ? 1 : 0;
    a.ablate = ablate_mask();
    if (rng) a.rng = *rng;
    if (dx) a.da0_out = workspace + w.gact;                         // (in place of layer 0's saved gates, 4H floats per step: see Lstm2BwdArgs)
    if (const int rc = nsd_lstm2_bwd_launch(a, d->H, (hipStream_t)stream)) return rc;
    return dx ? nsd_dx_launch(a.da0_out, params + pl.w_ih[0], dx, (long)d->B * d->T, 4 * d->H, d->C, (hipStream_t)stream) : NSD_OK;
}

int nsd_lstm_bwd(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm, uint32_t flags,
                 float *workspace, int64_t workspace_bytes, float *dx, void *stream) {
    return lstm_bwd_impl(d, params, x, drop_lstm, nullptr, flags, workspace, workspace_bytes, dx, stream);
}

int nsd_lstm_bwd_rng(const nsd_dims *d, const float *params, const float *x, const nsd_rng *rng, uint32_t flags,
                     float *workspace, int64_t workspace_bytes, void *stream) {
    RngArgs r;
    if (make_rng(rng, &r) != NSD_OK) return NSD_E_INVALID;
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!fused_train_shape(d)) { nsd_set_error("lstm_bwd_rng: shape outside the single-launch path (nsd_rng_path() == 0)"); return NSD_E_INVALID; }
    return lstm_bwd_impl(d, params, x, nullptr, &r, flags, workspace, workspace_bytes, nullptr, stream);
}

int nsd_grad_reduce(const nsd_dims *d, const float *workspace, int64_t workspace_bytes, float *grads, int32_t accumulate, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!workspace || !grads) { nsd_set_error("grad_reduce: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "grad_reduce", &w)) return rc;
    const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
    return nsd_grad_reduce_launch(workspace + w.slabs, align4(pl.lstm_total), d->B > 0 ? (int)w.n_slabs : 0, pl.lstm_total,
                                  workspace + w.hslabs, pl.total - pl.lstm_total, d->B, grads, accumulate,
                                  (hipStream_t)stream);
}

int nsd_grad_reduce_adam(const nsd_dims *d, const float *workspace, int64_t workspace_bytes, float *grads, float *p, float *m, float *v, float lr,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t step, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!workspace || !grads || !p || !m || !v) { nsd_set_error("grad_reduce_adam: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "grad_reduce_adam", &w)) return rc;
    const ParamLayout pl = nsd_make_layout(d->C, d->H, d->L, d->K, d->F);
    return nsd_grad_reduce_adam_launch(workspace + w.slabs, align4(pl.lstm_total), d->B > 0 ? (int)w.n_slabs : 0, pl.lstm_total,
                                       workspace + w.hslabs, pl.total - pl.lstm_total, d->B, grads, p, m, v, lr, beta1, beta2,
                                       eps, weight_decay, grad_scale, step, (hipStream_t)stream);
}

int nsd_loss_sum(const nsd_dims *d, const float *workspace, int64_t workspace_bytes, float *out, void *stream) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (!workspace || !out) { nsd_set_error("loss_sum: null pointer"); return NSD_E_INVALID; }
    nsd_ws_layout w;
    if (const int rc = check_ws(d, workspace, workspace_bytes, "loss_sum", &w)) return rc;
    return nsd_loss_sum_launch(workspace + w.loss, d->B, out, (hipStream_t)stream);
}

int nsd_adam_step(int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, int32_t step, void *stream) {
    if (n < 0 || !p || !g || !m || !v) { nsd_set_error("adam: null pointer or n<0"); return NSD_E_INVALID; }
    return nsd_adam_launch(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, step, nullptr, (hipStream_t)stream);
}

int nsd_adam_step_guarded(int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float grad_scale, int32_t step, const float *skip, void *stream) {
    if (n < 0 || !p || !g || !m || !v || !skip) { nsd_set_error("adam_guarded: null pointer or n<0"); return NSD_E_INVALID; }
    return nsd_adam_launch(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, step, skip, (hipStream_t)stream);
}

int nsd_train_masks(uint64_t seed, uint32_t base_stream, float p_lstm, float p_head, int64_t n_lstm, float *drop_lstm,
                    int64_t n_head, float *rrelu_slope, float *drop_head, void *stream) {
    if (n_lstm < 0 || n_head < 0 || (n_lstm > 0 && !drop_lstm) || (n_head > 0 && (!rrelu_slope || !drop_head))) {
        nsd_set_error("train_masks: null pointer or negative size");
        return NSD_E_INVALID;
    }
    return nsd_train_masks_launch(seed, base_stream, nullptr, p_lstm, p_head, n_lstm, drop_lstm, n_head, rrelu_slope, drop_head,
                                  (hipStream_t)stream);
}

int nsd_train_masks_dev(uint64_t seed, const int64_t *step_dev, float p_lstm, float p_head, int64_t n_lstm, float *drop_lstm,
                        int64_t n_head, float *rrelu_slope, float *drop_head, void *stream) {
    if (!step_dev || n_lstm < 0 || n_head < 0 || (n_lstm > 0 && !drop_lstm) || (n_head > 0 && (!rrelu_slope || !drop_head))) {
        nsd_set_error("train_masks_dev: null pointer or negative size");
        return NSD_E_INVALID;
    }
    return nsd_train_masks_launch(seed, 0, (const long long *)step_dev, p_lstm, p_head, n_lstm, drop_lstm, n_head, rrelu_slope,
                                  drop_head, (hipStream_t)stream);
}

int nsd_adam_step_dev(int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float grad_scale, const int64_t *step_dev, void *stream) {
    if (n < 0 || !p || !g || !m || !v || !step_dev) { nsd_set_error("adam_dev: null pointer or n<0"); return NSD_E_INVALID; }
    return nsd_adam_dev_launch(n, p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, (const long long *)step_dev,
                               (hipStream_t)stream);
}

int nsd_step_counter_inc(int64_t *step_dev, void *stream) {
    if (!step_dev) { nsd_set_error("step_counter_inc: null pointer"); return NSD_E_INVALID; }
    return nsd_step_inc_launch((long long *)step_dev, (hipStream_t)stream);
}

int nsd_dropout_mask(uint64_t seed, uint32_t stream_id, float p, int64_t n, float *out, void *stream) {
    if (n < 0 || !out) { nsd_set_error("dropout_mask: null pointer or n<0"); return NSD_E_INVALID; }
    return nsd_dropout_mask_launch(seed, stream_id, p, n, out, (hipStream_t)stream);
}

int nsd_rrelu_noise(uint64_t seed, uint32_t stream_id, int64_t n, float *out, void *stream) {
    if (n < 0 || !out) { nsd_set_error("rrelu_noise: null pointer or n<0"); return NSD_E_INVALID; }
    return nsd_rrelu_noise_launch(seed, stream_id, n, out, (hipStream_t)stream);
}

int nsd_gemm_bf16(const void *A, int64_t lda, int32_t a_kmajor, const void *B, int64_t ldb, int32_t b_kmajor, int64_t b_shift,
                  void *C, int64_t ldc, int32_t epilogue, const float *bias, int32_t M, int32_t N, int64_t K, int32_t splits,
                  void *stream) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t *)A; g.B = (const bf16_t *)B; g.lda = lda; g.ldb = ldb; g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor;
    g.b_shift = b_shift; g.C = C; g.ldc = ldc; g.bias = bias; g.M = M; g.N = N; g.K = K; g.splits = splits; g.epi = epilogue;
    return nsd_gemm_bf16_launch(g, (hipStream_t)stream);
}

}  // extern "C"
