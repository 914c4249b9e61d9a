// nsd_seq.h -- the sequence-batched path for large hidden sizes (BASELINE cfg3: H=256, K=5, B=1024, bf16; cfg5: C=64, T=1000,
// H=512, bidirectional).  Shared declarations of nsd_scan.hip (persistent recurrence kernels), nsd_head_tm.hip (attention
// pooling / LayerNorm / dense head on the time-major bf16 sequence) and the orchestration in nsd_seq.hip.
//
// Same model as the other paths -- EEG_LSTM of Neuro-Alpha-App/Utilities/lstm_eeg_model.py:13-39 with the ctor kwargs of
// :14 (and `bidirectional=True` where :16-22 would take it) -- computed with bf16 GEMM operands and bf16 saved activations,
// fp32 accumulation, fp32 cell state and gate arithmetic.
//
// Layout of the path.  Rows are TILE-MAJOR: r = seq_row(t, b) = ((b / 32) * T + t) * 32 + b % 32 -- the 32 trials of a batch
// tile are adjacent rows and the tile's T time steps follow one another, so every persistent workgroup walks ONE contiguous
// region of each tensor.  (Time-major rows t * Bp + b put consecutive steps of a tile Bp rows apart -- 2 MB in ga at cfg3 --
// and every step of every scan touched a fresh translation in each of ~8 tensors: issuing a step's saved-activation loads
// cost ~3 000 cycles of a 13 000-cycle backward step.)  Bp = B rounded up to the batch tile (padding trials carry zeros and
// produce zero gradients).  The GEMMs see [T*Bp][cols] matrices; the reduction over rows does not care about their order.
//   xbf   [T*Bp][CP]        the EEG windows as bf16 (CP = C rounded up to 16)
//   hs[l] [T*Bp][D*H]       h_t of layer l, direction d in columns d*H..        (recurrent exchange, dW_hh operand)
//   lk[l] [T*Bp][D*H]       what layer l+1 and the head read: h * dropout multiplier (== hs[l] without dropout)
//   cs    [T*Bp][H], ga [T*Bp][4H] per (l, d): cell state and activated gates (unit-major quads) for the backward pass
//   xproj [T*Bp/32][4H/32][64][16] per d (row tile (b/32)*T + t): input projection + bias as MFMA accumulator tiles (the scan's lanes load 32 bytes)
//   da    [T*Bp][D*4H]      gate pre-activation gradients of the layer in flight (unit-major columns c = 4u + g)
//   din   [T*Bp][D*H] fp32  gradient w.r.t. the layer's output coming from the layer above
#pragma once
#include "nsd_bf16.h"

#define NSD_SEQ_MAX_DIRS 2

__host__ __device__ __forceinline__ long seq_row(const int t, const int b, const int T) { return ((long)(b >> 5) * T + t) * 32 + (b & 31); }
#define NSD_SEQ_STATUS_WORDS 32
// The workspace starts with a persistent header of NSD_SEQ_HEADER_BYTES (word 0: STICKY status = OR of every time-out code since
// nsd_seq_workspace_init; never cleared by a forward call), followed by the status words of the evaluation in flight (cleared by
// every forward call): [0] status, [2] / [3] scan groups on one XCD / spread, [4..] diagnostic stamps.
// flag words per scan group: [0,64) one publish counter per wave of every member, [64,80) XCC ids of the members (rendezvous),
// [128,192) one consume counter ("ack") per wave of every member -- the backward partial-sum ring has ONE slot (nsd_scan2.hip)
// status bits: 1 / 2 a forward / backward scan group timed out (also ORed into the sticky word); 4 = a forward scan met a NaN / Inf
// hidden state (per evaluation, NOT sticky: the affected trials' logits are NaN as in the reference, the others are valid)
#define NSD_SEQ_ST_TIMEOUT_MASK 3
#define NSD_SEQ_ST_NONFINITE 4
#define NSD_SEQ_GROUP_WORDS 256
#define NSD_SEQ_ACK_WORD 128
#define NSD_SEQ_HEADER_BYTES 256
#define NSD_SEQ_HEADER_WORDS (NSD_SEQ_HEADER_BYTES / 4)

struct SeqDims {
    int B, T, C, H, L, K, F, D;
    int residual;                   // extension: out_l = LSTM_l(in_l) + in_l for l >= 1
    int fused2;                     // two unidirectional layers advance in one launch (nsd_scan2.hip)
    int Bp, MG, CP, P, groups;      // derived: padded batch, trials per group (32 or 64), padded channels, workgroups per group, groups per direction
};

struct SeqParamLayout {
    int64_t w_ih[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS], w_hh[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS];
    int64_t b_ih[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS], b_hh[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS];
    int64_t ln_w, ln_b, attn_w, attn_b, fc0_w, fc0_b, fc3_w, fc3_b, total, lstm_total;
};
SeqParamLayout nsd_seq_make_layout(int C, int H, int L, int K, int F, int D);

// byte offsets inside the caller's workspace
struct SeqWs {
    int64_t status, flags;                                       // int32[32] behind the persistent header; uint32 [L][D][groups][128] + backward copy
    int64_t xbf;
    int64_t wf[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS], wb[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS], wx[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS];
    int64_t wxt[NSD_MAX_LAYERS], bsum[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS];
    int64_t hs[NSD_MAX_LAYERS], lk[NSD_MAX_LAYERS];
    int64_t cs[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS], ga[NSD_MAX_LAYERS][NSD_SEQ_MAX_DIRS];
    int64_t xproj[NSD_SEQ_MAX_DIRS];
    int64_t da, da2, din[2];
    int64_t alpha, dscore, pooled, dpooled, loss, hb, parts, dbp, xch;
    int64_t total;
    int64_t flags_bytes, hb_stride;
};

// ---- recurrence kernels (nsd_scan.hip) ----------------------------------------------------------------------------------
struct ScanFwdArgs {
    const bf16_t *wf[NSD_SEQ_MAX_DIRS];      // [4H][H] recurrent weights, rows in accumulator-tile order
    const bf16_t *xproj[NSD_SEQ_MAX_DIRS];   // accumulator tiles, bias included (layers whose input is a hidden sequence)
    // layer 0 with at most 64 channels: the projection rides in the scan instead (CP / 16 <= 4 MFMAs per step; xproj unused)
    const bf16_t *wx0[NSD_SEQ_MAX_DIRS];     // [4H][CP] W_ih, rows in accumulator-tile order, or null
    const float *bsum0[NSD_SEQ_MAX_DIRS];    // [4H] b_ih + b_hh, tile order
    const bf16_t *xbf;                       // [T*Bp][CP]
    int CP;
    bf16_t *hs;                              // [T*Bp][ld]
    bf16_t *xch;                             // exchange ring [2][D][groups_total][MG*H] in gate-tile blocks (whole lines per producer wave)
    int groups_total;
    bf16_t *lk;                              // [T*Bp][ld] linked output (h [+ res]) * multiplier, or null (== h: readers use hs)
    const bf16_t *res;                       // [T*Bp][ld] residual input added to h (extension; the layer's own input) or null
    bf16_t *cs[NSD_SEQ_MAX_DIRS];            // [T*Bp][H] or null (inference)
    bf16_t *ga[NSD_SEQ_MAX_DIRS];            // [T*Bp][4H] or null
    unsigned *flags;                         // [D][groups][128] (one word per wave of every member + XCC ids), zeroed before the launch
    int *status;
    int B, Bp, T, D, ld, groups, group0, layer;  // groups of THIS launch, the first of them being batch tile group0
    int allow_l2_mode;                       // 0: always the write-through exchange (tests: both modes must agree)
    int spread_groups;                       // diagnostics: consecutive block ids per group = a group spread over all XCDs
    RngArgs rng;                             // rng.on: multiplier of stream rng.base on this layer's output
    int diag_short_grid;                     // diagnostic build only (forced time-out test): launch this many workgroups fewer
};
struct ScanBwdArgs {
    const bf16_t *wb[NSD_SEQ_MAX_DIRS];      // [H][4H] recurrent weights transposed, k = unit-major gate column
    const bf16_t *cs[NSD_SEQ_MAX_DIRS], *ga[NSD_SEQ_MAX_DIRS];
    bf16_t *da;                              // [T*Bp][D*4H]
    float *dbp;                              // [D][groups_total][4H] bias-gradient partials, one row per batch tile (unit-major columns)
    bf16_t *xch;                             // partial-sum ring [2][D][groups_total][P][P][NT][4][64 x 8 B] (nsd_scan.hip)
    const float *din;                        // [T*Bp][ld] fp32 gradient w.r.t. this layer's (multiplied) output (residual extension), or null
    const bf16_t *din_tiles;                 // the same gradient as bf16 accumulator tiles of the transposed product, register-group major:
                                             // [T*Bp/32][D*H/32][4 consumer waves][64 lanes][4 units] (GEMM_EPI_TILE_WAVE_BF16), or null
    float *dres;                             // [T*Bp][ld] residual extension: d(linked output) * multiplier, added to the input gradient; or null
    const float *alpha, *dscore;             // [T*Bp] (top layer)
    const float *dpooled;                    // [Bp][ld]
    const float *attn_w;                     // [ld]
    unsigned *flags;
    int *status;
    int B, Bp, T, D, ld, groups, group0, groups_total, layer;
    int allow_l2_mode, spread_groups;
    RngArgs rng;
    int diag_short_grid;                     // diagnostic build only (forced time-out test): launch this many workgroups fewer
};
// two unidirectional layers in ONE launch, layer 1 one time step behind layer 0 (nsd_scan2.hip): half the serial steps, the
// input projection of layer 1 and the input gradient of layer 1 ride in the scans (no GEMM, no xproj / din round trip)
struct Scan2FwdArgs {
    const bf16_t *wf0, *wx1, *wf1;           // [4H][H] each, rows in accumulator-tile order: W_hh0, W_ih1, W_hh1
    const float *bsum1;                      // [4H] b_ih1 + b_hh1, tile order
    const bf16_t *wx0;                       // [4H][CP] W_ih0, rows in accumulator-tile order: the layer-0 projection rides in the scan (K = CP <= 64)
    const float *bsum0;                      // [4H] b_ih0 + b_hh0, tile order
    const bf16_t *xbf;                       // [T*Bp][CP] the windows as bf16 (tile-major rows)
    int CP;
    bf16_t *hs0, *lk0, *hs1;                 // [T*Bp][H]; lk0 = h0 * multiplier or null
    bf16_t *xch;                             // exchange ring [2][groups_total][3][MG*H]
    int groups_total;
    bf16_t *cs0, *ga0, *cs1, *ga1;           // saves or null (inference)
    unsigned *flags;
    int *status;
    int B, Bp, T, groups, group0;
    int allow_l2_mode, spread_groups;
    RngArgs rng;
    int diag_short_grid;                     // diagnostic build only (forced time-out test): launch this many workgroups fewer
};
struct Scan2BwdArgs {
    const bf16_t *wb0, *wb1, *wxt1;          // [H][4H] each: W_hh0^T, W_hh1^T, W_ih1^T (k = unit-major gate column)
    const bf16_t *cs0, *ga0, *cs1, *ga1;
    bf16_t *da0, *da1;                       // [T*Bp][4H]
    bf16_t *xch;                             // partial-sum ring [2][groups_total][P][P][NT][4][64 lanes x (16 + 8) B] (PartRing, nsd_scan2.hip)
    float *dbp0, *dbp1;                      // [groups_total][4H]
    const float *alpha, *dscore, *dpooled, *attn_w;
    unsigned *flags;
    int *status;
    int B, Bp, T, groups, group0, groups_total;
    int allow_l2_mode, spread_groups;
    RngArgs rng;                             // rng.on: multiplier of layer 0's output
    int diag_short_grid;                     // diagnostic build only (forced time-out test): launch this many workgroups fewer
};
bool nsd_scan2_supported(int H, int MG);
int nsd_scan2_fwd_launch(const Scan2FwdArgs &a, int H, int MG, hipStream_t st);
int nsd_scan2_bwd_launch(const Scan2BwdArgs &a, int H, int MG, hipStream_t st);
int nsd_scan_fwd_launch(const ScanFwdArgs &a, int H, int MG, hipStream_t st);
int nsd_scan_bwd_launch(const ScanBwdArgs &a, int H, int MG, hipStream_t st);
bool nsd_scan_supported(int H);

struct PrepArgs {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    bf16_t *wf, *wb, *wx, *wxt;              // wxt may be null (layer 0)
    float *bsum;
    int H, I, Ipad, wxt_ld, wxt_off;
};
int nsd_seq_prep_launch(const PrepArgs &a, hipStream_t st);
int nsd_seq_xbf_launch(const float *x, bf16_t *xbf, int B, int Bp, int T, int C, int CP, hipStream_t st);

// ---- head on the time-major sequence (nsd_head_tm.hip) -------------------------------------------------------------------
struct HeadTmArgs {
    const bf16_t *top;                       // [T*Bp][DH]
    const float *ln_w, *ln_b, *attn_w, *attn_b, *fc0_w, *fc0_b, *fc3_w, *fc3_b;
    float eval_slope;
    float *logits, *probs;                   // [B][K]
    // training
    const int32_t *labels;
    float scale;
    RngArgs rng;                             // rng.on: RReLU slopes (stream base+1) and head dropout (base+2) drawn here
    const float *rrelu_slope, *drop_head;    // explicit [B][F] tensors instead (tests); null with rng.on
    float *alpha, *dscore;                   // [T*Bp]
    float *pooled, *dpooled;                 // [Bp][DH]
    float *loss;                             // [Bp]
    float *hb;                               // per-trial rows for the parameter-gradient reductions, stride hb_stride
    long hb_stride;
    int B, Bp, T, DH, F, K, train;
    const int *status;                       // status words of the evaluation (sticky word NSD_SEQ_HEADER_WORDS before): a scan time-out
                                             // poisons logits / probs / loss with NaN -- nothing downstream can mistake garbage for a result
};
int nsd_head_tm_launch(const HeadTmArgs &a, hipStream_t st);
// head parameter gradients from the per-trial rows: grads_head points at ln.weight inside the flat gradient vector
// (scratch: 16 * (3 DH + F DH + F + K F + K + 1) floats)
int nsd_head_tm_grads_launch(const float *hb, long hb_stride, int B, int DH, int F, int K, float *scratch, float *g_ln_w, float *g_ln_b,
                             float *g_attn_w, float *g_attn_b, float *g_fc0_w, float *g_fc0_b, float *g_fc3_w, float *g_fc3_b, hipStream_t st);
// per-trial row layout (floats): ln_out[DH] | dy_xhat[DH] | dy[DH] | dattn[DH] | dpre[F] | act[F] | dlogits[K] | dscore_sum[1]
static inline long nsd_head_tm_row_floats(int DH, int F, int K) { return 4L * DH + 2L * F + K + 1; }
