// nsd_seq.hip -- orchestration and C ABI of the sequence-batched path (declared in include/nsd.h, "nsd_seq_*"):
// parameter / workspace layout, the per-layer sequence  input-projection GEMM -> persistent scan  (forward) and
// persistent scan -> weight-gradient / input-gradient GEMMs  (backward), head on the time-major sequence.
// Model: EEG_LSTM of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:13-39 (+ bidirectional nn.LSTM where :16-22 would take the
// kwarg); parameter order = torch's state_dict order (…_l{k}, then …_l{k}_reverse per layer).
#include <string.h>
#include <vector>
#include "nsd_seq.h"
#include "nsd_args.h"
#include "nsd_diag.h"

namespace {

// ---- opt-in launch timing: DIAGNOSTIC BUILD ONLY (make diag -> libnsd_hip_diag.so, -DNSD_DIAG=1; declared in nsd_diag.h, not in
// include/nsd.h).  HIP events on the launch stream around the kernels of the path, so that a benchmark can quote the dominant
// kernel's own duration.  The product library has neither the state nor the entry points.
enum { PK_SCAN_FWD = 0, PK_SCAN_BWD = 1, PK_GEMM_XPROJ = 2, PK_GEMM_DW = 3, PK_GEMM_DIN = 4, PK_HEAD = 5, PK_HEAD_GRADS = 6, PK_PREP = 7, PK_COUNT = 8 };
#if NSD_DIAG
struct ProfRec { hipEvent_t a, b; int kind; };
struct ProfState { bool on = false; std::vector<ProfRec> recs; } g_prof;
struct ProfScope {
    hipStream_t st; int idx = -1;
    ProfScope(int kind, hipStream_t s) : st(s) {
        if (!g_prof.on) return;
        ProfRec r; r.kind = kind;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        (void)hipEventRecord(r.a, st);
        g_prof.recs.push_back(r);
        idx = (int)g_prof.recs.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof.recs[idx].b, st); }
};
#else
struct ProfScope { ProfScope(int, hipStream_t) {} ~ProfScope() {} };
#endif

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int derive(const nsd_dims *d, uint32_t flags, SeqDims *o) {
    if (nsd_check_dims(d) != NSD_OK) return NSD_E_INVALID;
    if (flags & ~(uint32_t)(NSD_FLAG_BIDIR | NSD_FLAG_RESIDUAL | NSD_FLAG_TRAIN | (NSD_DIAG ? NSD_DIAG_FLAG_ALL : 0u))) {
        nsd_set_error("seq path: unknown flag bits 0x%x", flags);
        return NSD_E_INVALID;
    }
    SeqDims s;
    s.B = d->B; s.T = d->T; s.C = d->C; s.H = d->H; s.L = d->L; s.K = d->K; s.F = d->F;
    s.D = (flags & NSD_FLAG_BIDIR) ? 2 : 1;
    s.residual = (flags & NSD_FLAG_RESIDUAL) ? 1 : 0;
    if (!nsd_scan_supported(s.H)) { nsd_set_error("seq path: hidden size %d not covered (64, 128, 256, 512)", s.H); return NSD_E_INVALID; }
    if (s.F > 64 || s.K > 64) { nsd_set_error("seq path: F=%d K=%d exceed 64", s.F, s.K); return NSD_E_INVALID; }
    s.P = s.H / 32;
    s.CP = (int)align_up(s.C, 16);
    const int cus = nsd_num_cus();
    const int cap = cus / (s.P * s.D);                          // groups that can be resident at once
    if (cap < 1) { nsd_set_error("seq path: H=%d D=%d needs %d workgroups per group, device has %d CUs", s.H, s.D, s.P * s.D, cus); return NSD_E_INVALID; }
    const int g32 = (s.B + 31) / 32;
    s.MG = (g32 <= cap) ? 32 : 64;
    s.Bp = (int)align_up(s.B > 0 ? s.B : 1, s.MG);
    s.groups = s.Bp / s.MG;
    // two unidirectional layers: one launch advances both, layer 1 a step behind layer 0 (nsd_scan2.hip)
    s.fused2 = (s.D == 1 && s.L == 2 && !s.residual && !(NSD_DIAG && (flags & NSD_DIAG_FLAG_NO_FUSED_LAYERS)) && s.CP <= 64 && nsd_scan2_supported(s.H, s.MG)) ? 1 : 0;
    *o = s;
    return NSD_OK;
}

constexpr long PARTS_FLOATS = 18L * 1024 * 1024;               // 64 splits of cfg3's 1024 x 256 weight gradient

SeqWs make_ws(const SeqDims &s) {
    SeqWs w;
    memset(&w, 0, sizeof(w));
    const int64_t R = (int64_t)s.T * s.Bp, H = s.H, G = 4 * H, DH = (int64_t)s.D * H;
    int64_t p = 0;
    auto take = [&](int64_t bytes) { const int64_t at = p; p = align_up(p + bytes, 256); return at; };
    (void)take(NSD_SEQ_HEADER_BYTES);                            // persistent header: sticky status (nsd_seq_workspace_init zeroes it)
    w.status = take(NSD_SEQ_STATUS_WORDS * 4);                   // == NSD_SEQ_HEADER_BYTES: nsd_seq_status / nsd_seq_guard rely on it
    w.flags_bytes = 2LL * s.L * s.D * s.groups * NSD_SEQ_GROUP_WORDS * 4;        // forward + backward flag sets of every layer (one word per wave)
    w.flags = take(w.flags_bytes);
    w.xbf = take(R * s.CP * 2);
    for (int l = 0; l < s.L; ++l) {
        const int64_t I = l == 0 ? s.C : DH, Ip = l == 0 ? s.CP : DH;
        (void)I;
        for (int d = 0; d < s.D; ++d) {
            w.wf[l][d] = take(G * H * 2);
            w.wb[l][d] = take(H * G * 2);
            w.wx[l][d] = take(G * Ip * 2);
            w.bsum[l][d] = take(G * 4);
        }
        w.wxt[l] = l > 0 ? take(DH * s.D * G * 2) : 0;
        w.hs[l] = take(R * DH * 2);
        w.lk[l] = (l < s.L - 1 || s.residual) ? take(R * DH * 2) : 0;
        for (int d = 0; d < s.D; ++d) {
            w.cs[l][d] = take(R * H * 2);
            w.ga[l][d] = take(R * G * 2);
        }
    }
    for (int d = 0; d < s.D; ++d) w.xproj[d] = take(R * G * 2);
    w.da = take(R * s.D * G * 2);
    w.da2 = s.fused2 ? take(R * G * 2) : 0;                  // fused two-layer scans: layer 1's da next to layer 0's
    w.din[0] = take(R * DH * (s.residual ? 4 : 2));             // bf16 owner-ordered tiles; row-major fp32 with the residual extension
    w.din[1] = s.residual ? take(R * DH * 4) : 0;           // residual extension: d(linked output) passed around the LSTM
    w.alpha = take(R * 4);
    w.dscore = take(R * 4);
    w.pooled = take((int64_t)s.Bp * DH * 4);
    w.dpooled = take((int64_t)s.Bp * DH * 4);
    w.loss = take((int64_t)s.Bp * 4);
    w.hb_stride = align_up(nsd_head_tm_row_floats((int)DH, s.F, s.K), 4);
    w.hb = take((int64_t)s.Bp * w.hb_stride * 4);
    w.dbp = take((int64_t)(s.D > 2 ? s.D : 2) * s.groups * G * 4);
    {   // exchange rings of the scans, sized for the widest user: h tiles of the forward scans (two slots, step parity) or the
        // partial-sum blocks of the fused backward (nsd_scan2.hip, PartRing: P * P * NT * 4 blocks of 1.5 KB per group, ONE slot)
        const int64_t tiles = 2LL * s.groups * 3 * s.MG * (s.D * 4LL * H) * 2, Pm = H / 32;
        const int64_t parts = 1LL * s.groups * s.D * Pm * Pm * (s.MG / 32) * 4 * 1536;
        w.xch = take(tiles > parts ? tiles : parts);
    }
    w.parts = take(PARTS_FLOATS * 4);                            // split-K partials of the weight-gradient GEMMs
    w.total = p;
    return w;
}

template <class T>
T *at(void *ws, int64_t off) { return reinterpret_cast<T *>(reinterpret_cast<char *>(ws) + off); }

int make_rng_args(const nsd_rng *r, RngArgs *out) {
    memset(out, 0, sizeof(*out));
    if (!r) return NSD_OK;
    if (!(r->p_lstm >= 0.f && r->p_lstm < 1.f) || !(r->p_head >= 0.f && r->p_head < 1.f)) { nsd_set_error("rng: p out of [0,1)"); return NSD_E_INVALID; }
    out->seed = r->seed; out->base = r->base_stream;
    out->thr_lstm = nsd_drop_threshold(r->p_lstm); out->thr_head = nsd_drop_threshold(r->p_head);
    out->keep_lstm = 1.0f / (1.0f - r->p_lstm); out->keep_head = 1.0f / (1.0f - r->p_head);
    out->on = 1;
    return NSD_OK;
}

// ---- small reduction kernels of the backward pass -----------------------------------------------------------------------------
// grads rows are torch's (g*H + u); the GEMM's rows are unit-major c = 4u + g.  out[(g*H+u)*I + i] = sum_z part[z][c][i], i < I
__global__ __launch_bounds__(256) void seq_reduce_dw_kernel(const float *part, int nparts, int H, int N, int I, float *out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = 4L * H * I;
    if (e >= total) return;
    const int c = (int)(e / I), i = (int)(e - (long)c * I), u = c >> 2, g = c & 3;
    const long MN = 4L * H * N;
    float s = 0.f;
    for (int z = 0; z < nparts; ++z) s += part[(long)z * MN + (long)c * N + i];
    out[(long)(g * H + u) * I + i] = s;
}
// bias gradients: sum of the per-tile rows the backward scan left
__global__ __launch_bounds__(256) void seq_reduce_db_kernel(const float *part, int nparts, int H, float *b_ih, float *b_hh) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 4 * H) return;
    float s = 0.f;
    for (int z = 0; z < nparts; ++z) s += part[(long)z * 4 * H + c];
    const int u = c >> 2, g = c & 3;
    b_ih[g * H + u] = s; b_hh[g * H + u] = s;
}

struct Ctx {
    SeqDims s;
    SeqParamLayout pl;
    SeqWs w;
    void *ws;
    const float *params;
    hipStream_t st;
    int cap;                                                     // groups per scan launch
    int spread;                                                  // NSD_FLAG_SPREAD_GROUPS
    int l2_mode;                                                 // same-XCD exchange shortcut allowed (diagnostic build: NSD_DIAG_FLAG_NO_L2_EXCHANGE clears it)
    int short_grid;                                              // diagnostic build: NSD_DIAG_FLAG_LOSE_MEMBER -> every scan launch misses its last workgroup
};

int split_count(int M, int N, long K) {
    // the 256 x 256 kernel (nsd_gemm_bf16.hip) wants at least one workgroup per CU, the 128 x 128 kernel about two
    const bool big = M >= 256 && N >= 256;
    const int tiles = big ? ((M + 255) / 256) * ((N + 255) / 256) : ((M + 127) / 128) * ((N + 127) / 128);
    int S = big ? (nsd_num_cus() + tiles - 1) / (tiles > 0 ? tiles : 1) : 512 / (tiles > 0 ? tiles : 1);
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    while (S > 1 && K / S < 256) --S;
    while ((long)S * M * N > PARTS_FLOATS && S > 1) --S;
    return S;
}

bool writes_lk(const SeqDims &s, int l, bool lstm_drop) { return (lstm_drop && l < s.L - 1) || (s.residual && l >= 1); }
const bf16_t *out_of(Ctx &c, int l, bool lstm_drop) { return at<bf16_t>(c.ws, writes_lk(c.s, l, lstm_drop) ? c.w.lk[l] : c.w.hs[l]); }

int forward(Ctx &c, const float *x, const RngArgs &rng, bool train) {
    const SeqDims &s = c.s;
    const int H = s.H, G = 4 * H, DH = s.D * H;
    const long R = (long)s.T * s.Bp;
    if (hipMemsetAsync(at<char>(c.ws, c.w.status), 0, (size_t)(c.w.flags + c.w.flags_bytes - c.w.status), c.st) != hipSuccess) {
        nsd_set_error("seq: memset failed"); return NSD_E_LAUNCH;
    }
    ProfScope *prep = new ProfScope(PK_PREP, c.st);
    if (const int rc = nsd_seq_xbf_launch(x, at<bf16_t>(c.ws, c.w.xbf), s.B, s.Bp, s.T, s.C, s.CP, c.st)) { delete prep; return rc; }
    for (int l = 0; l < s.L; ++l)
        for (int d = 0; d < s.D; ++d) {
            PrepArgs p;
            memset(&p, 0, sizeof(p));
            p.w_ih = c.params + c.pl.w_ih[l][d]; p.w_hh = c.params + c.pl.w_hh[l][d];
            p.b_ih = c.params + c.pl.b_ih[l][d]; p.b_hh = c.params + c.pl.b_hh[l][d];
            p.wf = at<bf16_t>(c.ws, c.w.wf[l][d]); p.wb = at<bf16_t>(c.ws, c.w.wb[l][d]); p.wx = at<bf16_t>(c.ws, c.w.wx[l][d]);
            p.wxt = l > 0 ? at<bf16_t>(c.ws, c.w.wxt[l]) : nullptr;
            p.bsum = at<float>(c.ws, c.w.bsum[l][d]);
            p.H = H; p.I = l == 0 ? s.C : DH; p.Ipad = l == 0 ? s.CP : DH; p.wxt_ld = s.D * G; p.wxt_off = d * G;
            if (const int rc = nsd_seq_prep_launch(p, c.st)) { delete prep; return rc; }
        }
    delete prep;
    if (s.fused2) {
        const bool lstm_drop = train && rng.on && rng.thr_lstm != 0;
        // (no projection GEMM: W_ih0 . x_t is CP / 16 <= 4 MFMAs per step inside the scan -- one launch and a 1-GB tile round trip less)
        for (int g0 = 0; g0 < s.groups; g0 += c.cap) {
            Scan2FwdArgs a;
            memset(&a, 0, sizeof(a));
            a.wf0 = at<bf16_t>(c.ws, c.w.wf[0][0]); a.wx1 = at<bf16_t>(c.ws, c.w.wx[1][0]); a.wf1 = at<bf16_t>(c.ws, c.w.wf[1][0]);
            a.bsum1 = at<float>(c.ws, c.w.bsum[1][0]);
            a.wx0 = at<bf16_t>(c.ws, c.w.wx[0][0]); a.bsum0 = at<float>(c.ws, c.w.bsum[0][0]); a.xbf = at<bf16_t>(c.ws, c.w.xbf); a.CP = s.CP;
            a.hs0 = at<bf16_t>(c.ws, c.w.hs[0]); a.lk0 = lstm_drop ? at<bf16_t>(c.ws, c.w.lk[0]) : nullptr; a.hs1 = at<bf16_t>(c.ws, c.w.hs[1]);
            a.xch = at<bf16_t>(c.ws, c.w.xch); a.groups_total = s.groups;
            if (train) {
                a.cs0 = at<bf16_t>(c.ws, c.w.cs[0][0]); a.ga0 = at<bf16_t>(c.ws, c.w.ga[0][0]);
                a.cs1 = at<bf16_t>(c.ws, c.w.cs[1][0]); a.ga1 = at<bf16_t>(c.ws, c.w.ga[1][0]);
            }
            a.flags = at<unsigned>(c.ws, c.w.flags) + (long)g0 * NSD_SEQ_GROUP_WORDS;
            a.status = at<int>(c.ws, c.w.status);
            a.B = s.B; a.Bp = s.Bp; a.T = s.T; a.groups = s.groups - g0 < c.cap ? s.groups - g0 : c.cap; a.group0 = g0;
            a.rng = rng; a.rng.on = lstm_drop ? 1 : 0;
            a.allow_l2_mode = c.l2_mode; a.spread_groups = c.spread; a.diag_short_grid = c.short_grid;
            ProfScope ps(PK_SCAN_FWD, c.st);
            if (const int rc = nsd_scan2_fwd_launch(a, H, s.MG, c.st)) return rc;
        }
        return NSD_OK;
    }
    for (int l = 0; l < s.L; ++l) {
        // a layer writes its LINKED output lk[l] = (h [+ its input]) * multiplier when that differs from h: inter-layer dropout
        // active (l < L-1) or the residual extension (l >= 1); readers take lk[l] then, hs[l] otherwise
        const bool lstm_drop = train && rng.on && rng.thr_lstm != 0;
        const bool masked = lstm_drop && l < s.L - 1;
        const bf16_t *in = l == 0 ? at<bf16_t>(c.ws, c.w.xbf) : out_of(c, l - 1, lstm_drop);
        const int Kin = l == 0 ? s.CP : DH;
        // layer 0 with at most 64 channels: W_ih . x_t is CP / 16 <= 4 MFMAs per step inside the scan (no GEMM, no tile round trip)
        const bool inproj = l == 0 && s.CP <= 64;
        for (int d = 0; d < s.D && !inproj; ++d) {
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A = at<bf16_t>(c.ws, c.w.wx[l][d]); g.lda = Kin; g.B = in; g.ldb = Kin;
            g.C = at<bf16_t>(c.ws, c.w.xproj[d]); g.bias = at<float>(c.ws, c.w.bsum[l][d]);
            g.M = G; g.N = (int)R; g.K = Kin; g.splits = 1; g.epi = GEMM_EPI_TILE_BF16;
            ProfScope ps(PK_GEMM_XPROJ, c.st);
            if (const int rc = nsd_gemm_bf16_launch(g, c.st)) return rc;
        }
        for (int g0 = 0; g0 < s.groups; g0 += c.cap) {
            ScanFwdArgs a;
            memset(&a, 0, sizeof(a));
            const int ng = s.groups - g0 < c.cap ? s.groups - g0 : c.cap;
            for (int d = 0; d < s.D; ++d) {
                a.wf[d] = at<bf16_t>(c.ws, c.w.wf[l][d]);
                a.xproj[d] = at<bf16_t>(c.ws, c.w.xproj[d]);
                if (inproj) { a.wx0[d] = at<bf16_t>(c.ws, c.w.wx[0][d]); a.bsum0[d] = at<float>(c.ws, c.w.bsum[0][d]); }
                a.cs[d] = train ? at<bf16_t>(c.ws, c.w.cs[l][d]) : nullptr;
                a.ga[d] = train ? at<bf16_t>(c.ws, c.w.ga[l][d]) : nullptr;
            }
            a.hs = at<bf16_t>(c.ws, c.w.hs[l]); a.xch = at<bf16_t>(c.ws, c.w.xch); a.groups_total = s.groups;
            a.xbf = at<bf16_t>(c.ws, c.w.xbf); a.CP = s.CP;
            a.lk = writes_lk(s, l, lstm_drop) ? at<bf16_t>(c.ws, c.w.lk[l]) : nullptr;
            a.res = (s.residual && l >= 1) ? in : nullptr;
            a.flags = at<unsigned>(c.ws, c.w.flags) + ((long)l * s.D * s.groups + (long)g0 * s.D) * NSD_SEQ_GROUP_WORDS;   // disjoint per chunk
            a.status = at<int>(c.ws, c.w.status);
            a.B = s.B; a.Bp = s.Bp; a.T = s.T; a.D = s.D; a.ld = DH; a.groups = ng; a.group0 = g0; a.layer = l;
            a.rng = rng;
            a.rng.on = masked ? 1 : 0;
            a.allow_l2_mode = c.l2_mode; a.spread_groups = c.spread; a.diag_short_grid = c.short_grid;
            ProfScope ps(PK_SCAN_FWD, c.st);
            if (const int rc = nsd_scan_fwd_launch(a, H, s.MG, c.st)) return rc;
        }
    }
    return NSD_OK;
}

HeadTmArgs head_args(Ctx &c, float *logits, float *probs) {
    const SeqDims &s = c.s;
    HeadTmArgs h;
    memset(&h, 0, sizeof(h));
    h.top = out_of(c, s.L - 1, false);                           // (the last layer is never multiplied; with the residual extension it is lk)
    h.ln_w = c.params + c.pl.ln_w; h.ln_b = c.params + c.pl.ln_b; h.attn_w = c.params + c.pl.attn_w; h.attn_b = c.params + c.pl.attn_b;
    h.fc0_w = c.params + c.pl.fc0_w; h.fc0_b = c.params + c.pl.fc0_b; h.fc3_w = c.params + c.pl.fc3_w; h.fc3_b = c.params + c.pl.fc3_b;
    h.eval_slope = (float)((0.125 + 1.0 / 3.0) / 2.0);           // nn.RReLU eval slope, lstm_eeg_model.py:27
    h.logits = logits; h.probs = probs;
    h.B = s.B; h.Bp = s.Bp; h.T = s.T; h.DH = s.D * s.H; h.F = s.F; h.K = s.K;
    h.status = at<int>(c.ws, c.w.status);
    return h;
}

int backward(Ctx &c, const RngArgs &rng, float *grads) {
    const SeqDims &s = c.s;
    const int H = s.H, G = 4 * H, DH = s.D * H;
    const long R = (long)s.T * s.Bp;
    float *parts = at<float>(c.ws, c.w.parts);
    const bool lstm_drop = rng.on && rng.thr_lstm != 0;
    // the backward scans' flag sets (rendezvous words, per-wave step counters) start from zero on EVERY backward call: a second
    // nsd_seq_train_bwd on the same forward (retain_graph) would otherwise find last call's counters at T and race through
    if (hipMemsetAsync(at<char>(c.ws, c.w.flags) + c.w.flags_bytes / 2, 0, (size_t)(c.w.flags_bytes / 2), c.st) != hipSuccess) {
        nsd_set_error("seq: memset failed"); return NSD_E_LAUNCH;
    }
    // contractions over the whole sequence for layer l: dW_hh, dW_ih (split-K, fixed-order reduction), biases from the scan's
    // per-tile sums.  da: [T*Bp][ldda] with direction d in columns d*4H..; dbp: [D][groups][4H]
    auto weight_grads = [&](int l, const bf16_t *da, long ldda, const float *dbp) -> int {
        const bf16_t *in = l == 0 ? at<bf16_t>(c.ws, c.w.xbf) : out_of(c, l - 1, lstm_drop);
        const int Kin = l == 0 ? s.CP : DH, I = l == 0 ? s.C : DH;
        for (int d = 0; d < s.D; ++d) {
            ProfScope ps(PK_GEMM_DW, c.st);
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A = da + (long)d * G; g.lda = ldda; g.a_kmajor = 1; g.b_kmajor = 1;
            g.C = parts; g.M = G; g.K = R; g.epi = GEMM_EPI_F32;
            // recurrent weights: operand h_{t-1} of the direction: one time step = 32 rows away inside the batch tile's block of
            // T * 32 rows (tile-major rows); the step before the first of a tile is zero
            g.B = at<bf16_t>(c.ws, c.w.hs[l]) + (long)d * H; g.ldb = DH;
            g.b_shift = d == 0 ? -32 : 32; g.b_period = (long)s.T * 32;
            if (H % 256 == 0 && Kin >= 256 && (G / 256) * ((H + Kin) / 256) <= 16) {
                // ONE pass over da for both gradients: columns [0, H) meet h_{t-1}, columns [H, H + Kin) the layer's input (da does not
                // fit the Infinity Cache: as two launches it is read from HBM twice).  Measured: cfg3's layer 1 (1024 x 512, 8 tiles
                // x 32 slices) 2 x 190 -> 305 us; cfg5's (2048 x 1536: 48 tiles, 5 slices fit the partial-sum buffer) 3.6 -> 4.0 ms,
                // so only where the fused problem still has few tiles and many slices
                g.B2 = in; g.ldb2 = Kin; g.b2_shift = 0; g.n_split = H;
                g.N = H + Kin; g.ldc = H + Kin;
                g.splits = split_count(G, H + Kin, R);
                if (const int rc = nsd_gemm_bf16_launch(g, c.st)) return rc;
                hipLaunchKernelGGL(seq_reduce_dw_kernel, dim3((unsigned)((4L * H * H + 255) / 256)), dim3(256), 0, c.st, parts, g.splits, H, H + Kin, H,
                                   grads + c.pl.w_hh[l][d]);
                hipLaunchKernelGGL(seq_reduce_dw_kernel, dim3((unsigned)((4L * H * I + 255) / 256)), dim3(256), 0, c.st, parts + H, g.splits, H, H + Kin, I,
                                   grads + c.pl.w_ih[l][d]);
            } else {
                g.N = H; g.ldc = H;
                g.splits = split_count(G, H, R);
                if (const int rc = nsd_gemm_bf16_launch(g, c.st)) return rc;
                hipLaunchKernelGGL(seq_reduce_dw_kernel, dim3((unsigned)((4L * H * H + 255) / 256)), dim3(256), 0, c.st, parts, g.splits, H, H, H,
                                   grads + c.pl.w_hh[l][d]);
                // input weights
                g.B = in; g.ldb = Kin; g.N = Kin; g.ldc = Kin; g.b_shift = 0; g.b_period = 0;
                g.splits = split_count(G, Kin, R);
                if (const int rc = nsd_gemm_bf16_launch(g, c.st)) return rc;
                hipLaunchKernelGGL(seq_reduce_dw_kernel, dim3((unsigned)((4L * H * I + 255) / 256)), dim3(256), 0, c.st, parts, g.splits, H, Kin, I,
                                   grads + c.pl.w_ih[l][d]);
            }
            hipLaunchKernelGGL(seq_reduce_db_kernel, dim3((G + 255) / 256), dim3(256), 0, c.st, dbp + (long)d * s.groups * G, s.groups, H,
                               grads + c.pl.b_ih[l][d], grads + c.pl.b_hh[l][d]);
            NSD_CHECK_LAUNCH("seq weight gradients");
        }
        return NSD_OK;
    };
    if (s.fused2) {
        float *dbp0 = at<float>(c.ws, c.w.dbp), *dbp1 = dbp0 + (long)s.groups * G;
        for (int g0 = 0; g0 < s.groups; g0 += c.cap) {
            Scan2BwdArgs a;
            memset(&a, 0, sizeof(a));
            a.wb0 = at<bf16_t>(c.ws, c.w.wb[0][0]); a.wb1 = at<bf16_t>(c.ws, c.w.wb[1][0]); a.wxt1 = at<bf16_t>(c.ws, c.w.wxt[1]);
            a.cs0 = at<bf16_t>(c.ws, c.w.cs[0][0]); a.ga0 = at<bf16_t>(c.ws, c.w.ga[0][0]);
            a.cs1 = at<bf16_t>(c.ws, c.w.cs[1][0]); a.ga1 = at<bf16_t>(c.ws, c.w.ga[1][0]);
            a.da0 = at<bf16_t>(c.ws, c.w.da); a.da1 = at<bf16_t>(c.ws, c.w.da2); a.dbp0 = dbp0; a.dbp1 = dbp1;
            a.xch = at<bf16_t>(c.ws, c.w.xch);
            a.alpha = at<float>(c.ws, c.w.alpha); a.dscore = at<float>(c.ws, c.w.dscore); a.dpooled = at<float>(c.ws, c.w.dpooled);
            a.attn_w = c.params + c.pl.attn_w;
            a.flags = at<unsigned>(c.ws, c.w.flags) + ((long)s.L * s.D * s.groups + (long)g0) * NSD_SEQ_GROUP_WORDS;
            a.status = at<int>(c.ws, c.w.status);
            a.B = s.B; a.Bp = s.Bp; a.T = s.T; a.groups = s.groups - g0 < c.cap ? s.groups - g0 : c.cap; a.group0 = g0; a.groups_total = s.groups;
            a.rng = rng; a.rng.on = lstm_drop ? 1 : 0;
            a.allow_l2_mode = c.l2_mode; a.spread_groups = c.spread; a.diag_short_grid = c.short_grid;
            ProfScope ps(PK_SCAN_BWD, c.st);
            if (const int rc = nsd_scan2_bwd_launch(a, H, s.MG, c.st)) return rc;
        }
        if (const int rc = weight_grads(1, at<bf16_t>(c.ws, c.w.da2), G, dbp1)) return rc;
        if (const int rc = weight_grads(0, at<bf16_t>(c.ws, c.w.da), G, dbp0)) return rc;
    } else
    for (int l = s.L - 1; l >= 0; --l) {
        const bool masked = lstm_drop && l < s.L - 1;
        for (int g0 = 0; g0 < s.groups; g0 += c.cap) {
            ScanBwdArgs a;
            memset(&a, 0, sizeof(a));
            const int ng = s.groups - g0 < c.cap ? s.groups - g0 : c.cap;
            for (int d = 0; d < s.D; ++d) {
                a.wb[d] = at<bf16_t>(c.ws, c.w.wb[l][d]);
                a.cs[d] = at<bf16_t>(c.ws, c.w.cs[l][d]);
                a.ga[d] = at<bf16_t>(c.ws, c.w.ga[l][d]);
            }
            a.da = at<bf16_t>(c.ws, c.w.da);
            // the upstream gradient of a lower layer: bf16 tiles in owner order, or (residual extension) row-major fp32
            a.din = (l == s.L - 1 || !s.residual) ? nullptr : at<float>(c.ws, c.w.din[0]);
            a.din_tiles = (l == s.L - 1 || s.residual) ? nullptr : at<bf16_t>(c.ws, c.w.din[0]);
            a.dres = (s.residual && l >= 1) ? at<float>(c.ws, c.w.din[1]) : nullptr;
            a.alpha = at<float>(c.ws, c.w.alpha); a.dscore = at<float>(c.ws, c.w.dscore); a.dpooled = at<float>(c.ws, c.w.dpooled);
            a.attn_w = c.params + c.pl.attn_w;
            a.flags = at<unsigned>(c.ws, c.w.flags) + ((long)(s.L + l) * s.D * s.groups + (long)g0 * s.D) * NSD_SEQ_GROUP_WORDS;
            a.dbp = at<float>(c.ws, c.w.dbp); a.groups_total = s.groups; a.xch = at<bf16_t>(c.ws, c.w.xch);
            a.status = at<int>(c.ws, c.w.status);
            a.B = s.B; a.Bp = s.Bp; a.T = s.T; a.D = s.D; a.ld = DH; a.groups = ng; a.group0 = g0; a.layer = l;
            a.rng = rng;
            a.rng.on = masked ? 1 : 0;
            a.allow_l2_mode = c.l2_mode; a.spread_groups = c.spread; a.diag_short_grid = c.short_grid;
            ProfScope ps(PK_SCAN_BWD, c.st);
            if (const int rc = nsd_scan_bwd_launch(a, H, s.MG, c.st)) return rc;
        }
        if (const int rc = weight_grads(l, at<bf16_t>(c.ws, c.w.da), (long)s.D * G, at<float>(c.ws, c.w.dbp))) return rc;
        if (l > 0) {                                             // gradient w.r.t. the layer's input, both directions in one contraction
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            if (s.residual) {                                    // row-major fp32 + d(linked output) of this layer (extension)
                g.A = at<bf16_t>(c.ws, c.w.da); g.lda = (long)s.D * G; g.B = at<bf16_t>(c.ws, c.w.wxt[l]); g.ldb = (long)s.D * G;
                g.C = at<float>(c.ws, c.w.din[0]); g.ldc = DH; g.M = (int)R; g.N = DH; g.K = (long)s.D * G; g.splits = 1; g.epi = GEMM_EPI_F32;
                g.add = at<float>(c.ws, c.w.din[1]);
            } else {
                // the TRANSPOSED product d_in^T [D*H, T*Bp] = W_ih^T' . da^T: an accumulator tile is then 32 units x the 32 trials of
                // one (batch tile, step) -- registers 4j..4j+3 of a lane are what lane (trial, half) of wave j of the lower layer's
                // backward scan needs -- written as bf16 in that order (the scan's lanes load 8 contiguous bytes)
                g.A = at<bf16_t>(c.ws, c.w.wxt[l]); g.lda = (long)s.D * G; g.B = at<bf16_t>(c.ws, c.w.da); g.ldb = (long)s.D * G;
                g.C = at<bf16_t>(c.ws, c.w.din[0]); g.ldc = 0; g.M = DH; g.N = (int)R; g.K = (long)s.D * G; g.splits = 1; g.epi = GEMM_EPI_TILE_WAVE_BF16;
            }
            ProfScope ps(PK_GEMM_DIN, c.st);
            if (const int rc = nsd_gemm_bf16_launch(g, c.st)) return rc;
        }
    }
    const long hbs = c.w.hb_stride;
    ProfScope ps(PK_HEAD_GRADS, c.st);
    return nsd_head_tm_grads_launch(at<float>(c.ws, c.w.hb), hbs, s.B, DH, s.F, s.K, parts, grads + c.pl.ln_w, grads + c.pl.ln_b, grads + c.pl.attn_w,
                                    grads + c.pl.attn_b, grads + c.pl.fc0_w, grads + c.pl.fc0_b, grads + c.pl.fc3_w, grads + c.pl.fc3_b, c.st);
}

int make_ctx(const nsd_dims *d, uint32_t flags, const float *params, void *ws, int64_t ws_bytes, void *stream, const char *who, Ctx *c) {
    if (const int rc = derive(d, flags, &c->s)) return rc;
    c->pl = nsd_seq_make_layout(c->s.C, c->s.H, c->s.L, c->s.K, c->s.F, c->s.D);
    c->w = make_ws(c->s);
    if (!ws || !params) { nsd_set_error("%s: null pointer", who); return NSD_E_INVALID; }
    if (ws_bytes < c->w.total) {
        nsd_set_error("%s: workspace of %lld bytes is smaller than nsd_seq_workspace_bytes() = %lld", who, (long long)ws_bytes, (long long)c->w.total);
        return NSD_E_WORKSPACE;
    }
    c->ws = ws; c->params = params; c->st = (hipStream_t)stream;
    c->cap = nsd_num_cus() / (c->s.P * c->s.D);
    c->l2_mode = (NSD_DIAG && (flags & NSD_DIAG_FLAG_NO_L2_EXCHANGE)) ? 0 : 1;
    c->spread = (NSD_DIAG && (flags & NSD_DIAG_FLAG_SPREAD_GROUPS)) ? 1 : 0;
    c->short_grid = (NSD_DIAG && (flags & NSD_DIAG_FLAG_LOSE_MEMBER)) ? 1 : 0;
    return NSD_OK;
}

}  // namespace

SeqParamLayout nsd_seq_make_layout(int C, int H, int L, int K, int F, int D) {
    SeqParamLayout o;
    memset(&o, 0, sizeof(o));
    int64_t p = 0;
    const int DH = D * H;
    for (int l = 0; l < L; ++l) {
        const int I = l == 0 ? C : DH;
        for (int d = 0; d < D; ++d) {                            // torch: all four tensors of a direction, then the _reverse ones
            o.w_ih[l][d] = p; p += 4LL * H * I;
            o.w_hh[l][d] = p; p += 4LL * H * H;
            o.b_ih[l][d] = p; p += 4LL * H;
            o.b_hh[l][d] = p; p += 4LL * H;
        }
    }
    o.lstm_total = p;
    o.ln_w = p; p += DH;   o.ln_b = p; p += DH;
    o.attn_w = p; p += DH; o.attn_b = p; p += 1;
    o.fc0_w = p; p += (int64_t)F * DH; o.fc0_b = p; p += F;
    o.fc3_w = p; p += (int64_t)K * F; o.fc3_b = p; p += K;
    o.total = p;
    return o;
}

extern "C" {

int64_t nsd_seq_param_count(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F, int32_t D) {
    if (C < 1 || H < 1 || L < 1 || L > NSD_MAX_LAYERS || K < 1 || F < 1 || D < 1 || D > NSD_SEQ_MAX_DIRS) {
        nsd_set_error("bad model dims C=%d H=%d L=%d K=%d F=%d D=%d", C, H, L, K, F, D);
        return NSD_E_INVALID;
    }
    return nsd_seq_make_layout(C, H, L, K, F, D).total;
}

int nsd_seq_param_layout(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F, int32_t D, int64_t *offsets) {
    if (nsd_seq_param_count(C, H, L, K, F, D) < 0 || !offsets) return NSD_E_INVALID;
    const SeqParamLayout o = nsd_seq_make_layout(C, H, L, K, F, D);
    int64_t *q = offsets;
    for (int l = 0; l < L; ++l)
        for (int d = 0; d < D; ++d) { *q++ = o.w_ih[l][d]; *q++ = o.w_hh[l][d]; *q++ = o.b_ih[l][d]; *q++ = o.b_hh[l][d]; }
    q[0] = o.ln_w; q[1] = o.ln_b; q[2] = o.attn_w; q[3] = o.attn_b; q[4] = o.fc0_w; q[5] = o.fc0_b; q[6] = o.fc3_w; q[7] = o.fc3_b;
    return NSD_OK;
}

int nsd_seq_supported(const nsd_dims *d, uint32_t flags) {
    SeqDims s;
    return derive(d, flags, &s) == NSD_OK ? 1 : 0;
}

int64_t nsd_seq_workspace_bytes(const nsd_dims *d, uint32_t flags) {
    SeqDims s;
    if (const int rc = derive(d, flags, &s)) return rc;
    return make_ws(s).total;
}

int nsd_seq_infer(const nsd_dims *d, const float *params, const float *x, uint32_t flags, float *logits, float *probs, void *workspace,
                  int64_t workspace_bytes, void *stream) {
    Ctx c;
    if (const int rc = make_ctx(d, flags, params, workspace, workspace_bytes, stream, "seq_infer", &c)) return rc;
    if (!x || !logits) { nsd_set_error("seq_infer: null pointer"); return NSD_E_INVALID; }
    if (d->B == 0) return NSD_OK;
    RngArgs off;
    memset(&off, 0, sizeof(off));
    if (const int rc = forward(c, x, off, false)) return rc;
    HeadTmArgs h = head_args(c, logits, probs);
    ProfScope ps(PK_HEAD, c.st);
    return nsd_head_tm_launch(h, c.st);
}

int nsd_seq_train_fwd(const nsd_dims *d, const float *params, const float *x, const nsd_rng *rng, const int32_t *labels, float scale,
                      uint32_t flags, void *workspace, int64_t workspace_bytes, float *logits, void *stream) {
    Ctx c;
    if (const int rc = make_ctx(d, flags, params, workspace, workspace_bytes, stream, "seq_train_fwd", &c)) return rc;
    if (!x || !logits || !labels) { nsd_set_error("seq_train_fwd: null pointer"); return NSD_E_INVALID; }
    if (d->B == 0) return NSD_OK;
    RngArgs r;
    if (const int rc = make_rng_args(rng, &r)) return rc;
    if (const int rc = forward(c, x, r, true)) return rc;
    HeadTmArgs h = head_args(c, logits, nullptr);
    h.train = 1; h.labels = labels; h.scale = scale; h.rng = r;
    h.alpha = at<float>(c.ws, c.w.alpha); h.dscore = at<float>(c.ws, c.w.dscore);
    h.pooled = at<float>(c.ws, c.w.pooled); h.dpooled = at<float>(c.ws, c.w.dpooled); h.loss = at<float>(c.ws, c.w.loss);
    h.hb = at<float>(c.ws, c.w.hb); h.hb_stride = c.w.hb_stride;
    // padding trials: their dpooled / dscore must be zero so that they contribute nothing to any gradient
    const int DH = c.s.D * c.s.H;
    if (c.s.Bp > c.s.B) {
        if (hipMemsetAsync(h.dpooled + (long)c.s.B * DH, 0, (size_t)(c.s.Bp - c.s.B) * DH * 4, c.st) != hipSuccess ||
            hipMemsetAsync(h.alpha, 0, (size_t)c.s.T * c.s.Bp * 4, c.st) != hipSuccess ||
            hipMemsetAsync(h.dscore, 0, (size_t)c.s.T * c.s.Bp * 4, c.st) != hipSuccess) {
            nsd_set_error("seq_train_fwd: memset failed");
            return NSD_E_LAUNCH;
        }
    }
    ProfScope ps(PK_HEAD, c.st);
    return nsd_head_tm_launch(h, c.st);
}

int nsd_seq_train_bwd(const nsd_dims *d, const float *params, const nsd_rng *rng, uint32_t flags, void *workspace, int64_t workspace_bytes,
                      float *grads, void *stream) {
    Ctx c;
    if (const int rc = make_ctx(d, flags, params, workspace, workspace_bytes, stream, "seq_train_bwd", &c)) return rc;
    if (!grads) { nsd_set_error("seq_train_bwd: null pointer"); return NSD_E_INVALID; }
    if (d->B == 0) return NSD_OK;
    RngArgs r;
    if (const int rc = make_rng_args(rng, &r)) return rc;
    return backward(c, r, grads);
}

int nsd_seq_loss_sum(const nsd_dims *d, uint32_t flags, const void *workspace, int64_t workspace_bytes, float *out, void *stream) {
    SeqDims s;
    if (const int rc = derive(d, flags, &s)) return rc;
    const SeqWs w = make_ws(s);
    if (!workspace || !out) { nsd_set_error("seq_loss_sum: null pointer"); return NSD_E_INVALID; }
    if (workspace_bytes < w.total) { nsd_set_error("seq_loss_sum: workspace too small"); return NSD_E_WORKSPACE; }
    return nsd_loss_sum_launch(reinterpret_cast<const float *>(reinterpret_cast<const char *>(workspace) + w.loss), s.B, out, (hipStream_t)stream);
}

// Zero the persistent header of a freshly allocated workspace (the sticky status word).  Call once per allocation; an
// uninitialised header reads as a failure, never as success.
int nsd_seq_workspace_init(void *workspace, int64_t workspace_bytes, void *stream) {
    if (!workspace) { nsd_set_error("seq_workspace_init: null pointer"); return NSD_E_INVALID; }
    if (workspace_bytes < NSD_SEQ_HEADER_BYTES + NSD_SEQ_STATUS_WORDS * 4) { nsd_set_error("seq_workspace_init: workspace too small"); return NSD_E_WORKSPACE; }
    if (hipMemsetAsync(workspace, 0, NSD_SEQ_HEADER_BYTES + NSD_SEQ_STATUS_WORDS * 4, (hipStream_t)stream) != hipSuccess) {
        nsd_set_error("seq_workspace_init: %s", hipGetErrorString(hipGetLastError()));
        return NSD_E_LAUNCH;
    }
    return NSD_OK;
}

// Blocking read of the path's status: status_out[0] = code of the evaluation in flight OR the sticky word (0 = ok, bit 0 / bit 1 =
// a forward / backward scan group gave up waiting for a member: a workgroup of the group was not resident, or the device was
// lost), [1] = the sticky word alone (every code since nsd_seq_workspace_init), [2] / [3] = scan groups on one XCD / spread.
int nsd_seq_status(const void *workspace, int32_t *status_out, void *stream) {
    if (!workspace || !status_out) { nsd_set_error("seq_status: null pointer"); return NSD_E_INVALID; }
    int32_t sticky = 0, cur[4] = {0, 0, 0, 0};
    const char *base = reinterpret_cast<const char *>(workspace);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
        hipMemcpy(&sticky, base, sizeof(sticky), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(cur, base + NSD_SEQ_HEADER_BYTES, sizeof(cur), hipMemcpyDeviceToHost) != hipSuccess) {
        nsd_set_error("seq_status: %s", hipGetErrorString(hipGetLastError()));
        return NSD_E_LAUNCH;
    }
    status_out[0] = cur[0] | sticky; status_out[1] = sticky; status_out[2] = cur[2]; status_out[3] = cur[3];
    return NSD_OK;
}

// flag_out[0] (device, fp32) = 1 if the workspace reports a scan time-out (current or sticky), else 0; enqueued, never blocks.
// The trainer appends the flag to the flat gradient it all-reduces, so that EVERY rank skips the update when ANY rank failed.
int nsd_seq_guard(const void *workspace, float *flag_out, void *stream) {
    if (!workspace || !flag_out) { nsd_set_error("seq_guard: null pointer"); return NSD_E_INVALID; }
    return nsd_seq_guard_launch(reinterpret_cast<const int *>(workspace), NSD_SEQ_HEADER_WORDS, flag_out, (hipStream_t)stream);
}

#if NSD_DIAG
// Diagnostic build only (nsd_diag.h).  enable != 0 starts recording HIP events (on the launch stream) around the kernels of
// every following nsd_seq_* call, 0 stops and discards.  nsd_seq_profile_read sums one kind and forgets its records
// (BLOCKING: waits for those events).  kind: 0 forward scan, 1 backward scan, 2 input-projection GEMM, 3 weight-gradient
// GEMMs (+ their reductions), 4 input-gradient GEMM, 5 head, 6 head parameter gradients, 7 operand preparation.
int nsd_seq_profile(int32_t enable) {
    for (ProfRec &r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    g_prof.on = enable != 0;
    return NSD_OK;
}
int nsd_seq_profile_read(int32_t kind, float *total_ms, int32_t *count) {
    if (kind < 0 || kind >= PK_COUNT || !total_ms || !count) { nsd_set_error("seq_profile_read: bad argument"); return NSD_E_INVALID; }
    float tot = 0.f; int n = 0;
    std::vector<ProfRec> keep;
    for (ProfRec &r : g_prof.recs) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; ++n; }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.recs.swap(keep);
    *total_ms = tot; *count = n;
    return NSD_OK;
}
#endif

}  // extern "C"
