/*
 * nsd.h -- C ABI of libnsd_hip.so: the MI355X (gfx950) implementation of the
 * EEG_LSTM hot path of aa217/Neural-Speech-Decoding.
 *
 * The reference has NO native interface: its hot path is a torch nn.Module
 * (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:13-39) called through
 * SimplePredictor.predict (lstm_eeg_model.py:86-101).  Each entry point below
 * names the reference lines whose arithmetic it replaces; the Python facade in
 * neural-speech-decoding_amd/ binds them with ctypes (INTEGRATION.md shows the
 * stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C symbols, no C++/torch types; every function returns int:
 *     0 = ok, <0 = error (NSD_E_*); nsd_last_error() gives thread-local text.
 *   - every entry point that reads or writes the training workspace takes `workspace_bytes`, the size of the
 *     caller's buffer, and returns NSD_E_WORKSPACE (launching nothing) when it is smaller than
 *     nsd_workspace_bytes(d) -- a C caller cannot be overrun.
 *   - the CALLER owns every buffer.  All pointers are DEVICE pointers (fp32
 *     unless stated) valid on the current HIP device; the library allocates
 *     nothing and never synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the legacy default stream).
 *   - re-entrant per stream; kernels are deterministic (no float atomics).
 *   - shapes: B trials, T time steps, C channels, H hidden, L layers,
 *     K classes, F = width of the first dense layer (32 in the reference).
 *
 * Flat parameter vector (same order as the reference state_dict, so the
 * reference .pth maps 1:1; see nsd_param_layout):
 *   for l in 0..L-1: lstm.weight_ih_l{l}[4H,I_l] lstm.weight_hh_l{l}[4H,H]
 *                    lstm.bias_ih_l{l}[4H] lstm.bias_hh_l{l}[4H]   (I_0=C, I_l=H)
 *   ln.weight[H] ln.bias[H] attn.weight[H] attn.bias[1]
 *   fc.0.weight[F,H] fc.0.bias[F] fc.3.weight[K,F] fc.3.bias[K]
 */
#ifndef NSD_H
#define NSD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSD_VERSION 300          /* 0.3.0: sequence-batched path: persistent workspace header (nsd_seq_workspace_init), sticky
                                    status, nsd_seq_guard / nsd_adam_step_guarded; diagnostics left the shipped library */
#define NSD_MAX_LAYERS 8

#define NSD_OK            0
#define NSD_E_INVALID    -1      /* bad argument / unsupported shape */
#define NSD_E_LAUNCH     -2      /* HIP launch or runtime error */
#define NSD_E_WORKSPACE  -3      /* workspace_bytes < nsd_workspace_bytes(d): nothing was launched */

/* flags */
#define NSD_FLAG_RESIDUAL   1u   /* extension (not in the reference): out_l = LSTM_l(in_l) + in_l, l>=1 */
#define NSD_FLAG_TRAIN      2u   /* keep activations in the workspace for nsd_*_bwd */
#define NSD_FLAG_BIDIR      8u   /* nsd_seq_* entry points only: bidirectional LSTM (torch.nn.LSTM(bidirectional=True)); the
                                    sequence fed to the attention pooling and the head is 2H wide */
#define NSD_FLAG_BF16       4u   /* large-H batched path only (H % 16 == 0, H >= 64, B >= 16; ignored elsewhere): GEMM operands
                                    rounded to bf16 at the matrix pipe (fp32 accumulate, fp32 storage and cell arithmetic) --
                                    BASELINE cfg3's precision; results differ from fp32 at the 1e-2 level */

typedef struct nsd_dims {
    int32_t B, T, C, H, L, K, F;
} nsd_dims;

/* word offsets (units of float) of the regions inside the training workspace */
typedef struct nsd_ws_layout {
    int64_t hseq;      /* [L,B,T,H]   h_t of every layer (LSTM's own output)          */
    int64_t cseq;      /* [L,B,T,H]   c_t                                              */
    int64_t gact;      /* [L,B,T,H,4] activated gates, unit-major: (i,f,g,o) per unit  */
    int64_t inseq;     /* [L-1,B,T,H] input fed to layer l+1 (after residual+dropout)  */
    int64_t top;       /* [B,T,H]     sequence fed to attention (== hseq[L-1] unless residual) */
    int64_t alpha;     /* [B,T]       attention weights                                */
    int64_t pooled;    /* [B,H]                                                        */
    int64_t fc0_pre;   /* [B,F]       pre-activation of fc.0                           */
    int64_t dscore;    /* [B,T]       d loss / d attention score  (head_bwd -> lstm_bwd) */
    int64_t dpooled;   /* [B,H]       d loss / d pooled           (head_bwd -> lstm_bwd) */
    int64_t loss;      /* [B]         per-trial CE loss                                */
    int64_t adpack;    /* [B,T,4]     {alpha, dscore, 0, 0} per step: 16-byte records for the LSTM backward's LDS-DMA */
    int64_t slabs;     /* [n_slabs,P_lstm] per-workgroup partial gradients of the LSTM stack */
    int64_t n_slabs;
    int64_t hslabs;    /* [B,P_head]  per-trial gradients of ln/attn/fc                  */
    int64_t da_seq;    /* [B,T,4H]    generic path only: pre-activation gradients of the layer in flight */
    int64_t din;       /* [2,B,T,H]   generic path only: gradient w.r.t. a layer's input (ping-pong)      */
    int64_t total;     /* floats */
} nsd_ws_layout;

int         nsd_version(void);
const char *nsd_last_error(void);

/* number of floats in the flat parameter vector; <0 on bad dims */
int64_t nsd_param_count(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F);
/* offsets[4L+8]: per layer w_ih,w_hh,b_ih,b_hh; then ln.w ln.b attn.w attn.b fc0.w fc0.b fc3.w fc3.b */
int     nsd_param_layout(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F, int64_t *offsets);

/* workspace needed by the train-mode calls for these dims, in BYTES; layout optional */
int64_t nsd_workspace_bytes(const nsd_dims *d, nsd_ws_layout *layout_out);

/* 1 if the fused register-resident kernels cover these dims (H in {32,48,64}, L==2, C<=8), else 0: the shape-generic
 * per-layer kernels (nsd_lstm_generic.hip) are used -- same results, not tuned */
int     nsd_fast_path(const nsd_dims *d);

/*
 * Per-channel z-score over time, y = (x - mean_T) / (std_T(ddof=0) + 1e-6).
 * Replaces normalize_eeg, Neuro-Alpha-App/Frontend/app.py:166-170.  x,y [B,T,C]; in-place allowed.
 */
int nsd_zscore_fwd(const float *x, float *y, int32_t B, int32_t T, int32_t C, void *stream);

/*
 * Inference: x[B,T,C] -> logits[B,K] (+ probs[B,K] if non-NULL).
 * Replaces EEG_LSTM.forward in eval mode (lstm_eeg_model.py:32-39) and the class
 * softmax of SimplePredictor.predict (lstm_eeg_model.py:97).
 * scratch: device buffer of nsd_infer_scratch_bytes(d) bytes.
 */
int64_t nsd_infer_scratch_bytes(const nsd_dims *d);
int nsd_infer(const nsd_dims *d, const float *params, const float *x, uint32_t flags,
              float *logits, float *probs, void *scratch, void *stream);

/*
 * Stacked LSTM forward (lstm_eeg_model.py:34, self.lstm(x)), train mode.
 * drop_lstm: NULL, or multiplier masks [L-1,B,T,H] (0 or 1/(1-p)) applied to the output of every
 *            layer but the last (nn.LSTM(dropout=p), lstm_eeg_model.py:21).
 * Fills hseq/cseq/gact/inseq/top of the workspace.
 */
int nsd_lstm_fwd(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm,
                 uint32_t flags, float *workspace, int64_t workspace_bytes, void *stream);

/*
 * Head forward: attention pooling over time, LayerNorm, fc (lstm_eeg_model.py:35-39),
 * optional class softmax (lstm_eeg_model.py:97).  Reads `top` from the workspace.
 * rrelu_slope: NULL -> eval slope (lower+upper)/2, else per-element slopes [B,F] (train-mode RReLU noise)
 * drop_head:   NULL or multiplier mask [B,F] (nn.Dropout, lstm_eeg_model.py:28)
 */
int nsd_head_fwd(const nsd_dims *d, const float *params, const float *rrelu_slope, const float *drop_head,
                 float *workspace, int64_t workspace_bytes, float *logits, float *probs, void *stream);

/*
 * Head backward.  Either dlogits[B,K] is given, or labels[B] (int32) with `scale`:
 * then dlogits = (softmax(logits) - onehot(label)) * scale (mean CE when scale = 1/B_global) and the
 * per-trial CE loss is written to the workspace's `loss` region.  Writes dscore/dpooled for
 * nsd_lstm_bwd and the head's partial gradients into the slabs.
 */
int nsd_head_bwd(const nsd_dims *d, const float *params, const float *rrelu_slope, const float *drop_head,
                 const float *logits, const float *dlogits, const int32_t *labels, float scale,
                 float *workspace, int64_t workspace_bytes, void *stream);

/*
 * Train-step head: nsd_head_fwd + mean cross-entropy + nsd_head_bwd of every trial in ONE launch (sequence and head
 * parameters staged in LDS once).  Same workspace outputs as the two separate calls; logits[B,K] is written too.
 */
int nsd_head_train(const nsd_dims *d, const float *params, const float *rrelu_slope, const float *drop_head,
                   const int32_t *labels, float scale, float *workspace, int64_t workspace_bytes, float *logits, void *stream);

/*
 * nsd_lstm_fwd + nsd_head_train in ONE launch where the shape allows (H = 48, L = 2, T <= 1024, F <= 64, K <= 8: the
 * attention pooling rides along the recurrence and the dense head, loss and head backward run in the kernel's tail);
 * other shapes run the two launches it replaces.  Same outputs, workspace contents and gradient slabs either way --
 * except that the single launch does not write the workspace's `top` region when the residual extension is off (it
 * would duplicate the last layer's hseq, which nsd_lstm_bwd and the kernel's own tail read instead).
 */
int nsd_lstm_head_train(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm,
                        const float *rrelu_slope, const float *drop_head, const int32_t *labels, float scale, uint32_t flags,
                        float *workspace, int64_t workspace_bytes, float *logits, void *stream);

/*
 * Train-mode random streams generated INSIDE the kernels (no mask tensors in HBM): the three streams of one step are
 * value(seed, base_stream + {0: LSTM inter-layer dropout [B,T,H], 1: RReLU slope [B,F], 2: head dropout [B,F]}, index),
 * the same pure function as nsd_train_masks / oracle -- a step run this way is bit-identical to the same step run with
 * the masks of nsd_train_masks passed explicitly.  Only where nsd_rng_path(d) != 0 (the single-launch H = 48 shape).
 */
typedef struct nsd_rng {
    uint64_t seed;
    uint32_t base_stream;
    float p_lstm, p_head;
} nsd_rng;
int nsd_rng_path(const nsd_dims *d);
int nsd_lstm_head_train_rng(const nsd_dims *d, const float *params, const float *x, const nsd_rng *rng, const int32_t *labels,
                            float scale, uint32_t flags, float *workspace, int64_t workspace_bytes, float *logits, void *stream);
int nsd_lstm_bwd_rng(const nsd_dims *d, const float *params, const float *x, const nsd_rng *rng, uint32_t flags,
                     float *workspace, int64_t workspace_bytes, void *stream);

/*
 * Stacked LSTM backward (BPTT) through lstm_eeg_model.py:34 with the activations kept by nsd_lstm_fwd.
 * Partial gradients go to the slabs.  dx (optional, may be NULL): dL/dx [B,T,C], what autograd through self.lstm(x)
 * (lstm_eeg_model.py:34) returns for the EEG window -- formed as da0 . W_ih0 behind the backward pass, for H = 48 (L = 2, C <= 8; the
 * one-trial kernel then runs whatever the batch and leaves da0 IN PLACE of layer 0's saved gates: one backward per forward) and on the
 * shape-generic path; on the other paths a non-NULL dx returns NSD_E_INVALID and launches nothing.  The parameter gradients never need it.
 */
int nsd_lstm_bwd(const nsd_dims *d, const float *params, const float *x, const float *drop_lstm,
                 uint32_t flags, float *workspace, int64_t workspace_bytes, float *dx, void *stream);

/* grads[P] (=|+=) sum over slabs.  accumulate: 0 overwrite, 1 add to existing. */
int nsd_grad_reduce(const nsd_dims *d, const float *workspace, int64_t workspace_bytes, float *grads, int32_t accumulate, void *stream);

/* Single-rank train step tail: nsd_grad_reduce (grads[] is still written) followed, in the same launch, by
 * nsd_adam_step on the reduced gradient -- same arithmetic in the same order as the two separate calls.  With more
 * than one rank the all-reduce sits between the two and the separate entry points are used. */
int nsd_grad_reduce_adam(const nsd_dims *d, const float *workspace, int64_t workspace_bytes, float *grads, float *p, float *m, float *v, float lr,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t step, void *stream);

/* sum of the per-trial losses written by nsd_head_bwd -> out[0] (device) */
int nsd_loss_sum(const nsd_dims *d, const float *workspace, int64_t workspace_bytes, float *out, void *stream);

/* torch.optim.Adam semantics (no amsgrad); step counted from 1; all vectors length n */
int nsd_adam_step(int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, int32_t step, void *stream);
/* the same update, skipped entirely (p, m, v untouched) when the device flag skip[0] != 0: see nsd_seq_guard */
int nsd_adam_step_guarded(int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float grad_scale, int32_t step, const float *skip, void *stream);

/*
 * Counter-based random streams of the trainer (the reference's training RNG is torch's and is not
 * portable): value(seed, stream, index) is a pure function, identical in oracle/nsd_oracle.c.
 *   nsd_dropout_mask: out[i] = keep ? 1/(1-p) : 0       nsd_rrelu_noise: out[i] ~ U(1/8, 1/3)
 */
int nsd_dropout_mask(uint64_t seed, uint32_t stream_id, float p, int64_t n, float *out, void *stream);
/* the three streams of one train step in one launch: drop_lstm[n_lstm] = stream base, rrelu_slope[n_head] = base+1,
 * drop_head[n_head] = base+2 (bit-identical to the three separate calls) */
int nsd_train_masks(uint64_t seed, uint32_t base_stream, float p_lstm, float p_head, int64_t n_lstm, float *drop_lstm,
                    int64_t n_head, float *rrelu_slope, float *drop_head, void *stream);
int nsd_rrelu_noise(uint64_t seed, uint32_t stream_id, int64_t n, float *out, void *stream);

/*
 * hipGraph-friendly variants: whatever changes from step to step (the Adam step number, the random-stream ids) is read
 * from a device-side counter instead of being a kernel argument, so that one captured graph can be replayed.
 *   nsd_step_counter_inc  step_dev[0] += 1 (int64, device); call it first in the captured step
 *   nsd_train_masks_dev   like nsd_train_masks with base_stream = 4 * (step_dev[0] & 0x3fffffff)
 *   nsd_adam_step_dev     like nsd_adam_step with step = step_dev[0]
 */
int nsd_step_counter_inc(int64_t *step_dev, void *stream);
int nsd_train_masks_dev(uint64_t seed, const int64_t *step_dev, float p_lstm, float p_head, int64_t n_lstm, float *drop_lstm,
                        int64_t n_head, float *rrelu_slope, float *drop_head, void *stream);
int nsd_adam_step_dev(int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2, float eps,
                      float weight_decay, float grad_scale, const int64_t *step_dev, void *stream);

/*
 * ---- sequence-batched path for large hidden sizes (BASELINE cfg3: H=256, K=5, B=1024 bf16; cfg5: bidirectional H=512) ----
 *
 * Building block, exported so that it can be tested and timed on its own: C[M,N] = A . B with bf16 operands (device
 * pointers to bf16 bit patterns) and fp32 accumulation on v_mfma_f32_32x32x16_bf16.
 *   a_kmajor == 0: A is [M][lda], k contiguous;  != 0: A is [K][lda], m contiguous
 *   b_kmajor == 0: B is [N][ldb], k contiguous;  != 0: B is [K][ldb], n contiguous; row k is then taken from row
 *                  k + b_shift, rows outside [0, K) read as zero (h_{t-1} for the recurrent weight gradient)
 *   epilogue 0: C fp32 [M][ldc]; splits > 1 writes split z to C + z*M*ldc (the caller sums the parts)
 *            1: C bf16 [M][ldc]
 *            2: C bf16 as 32x32 accumulator tiles [N/32][M/32][64][16] (+ bias[m]): the layout the scan kernels' lanes load
 *            3: the same tiles with the register group first, [N/32][M/32][4][64][4]: element (g, lane, e) is tile row
 *               8 g + 4 (lane >> 5) + e, column lane & 31 (what lane `lane` of wave g of a backward scan owns, 8 contiguous bytes)
 * Contiguous dimensions and leading dimensions must be multiples of 8 elements.
 */
int nsd_gemm_bf16(const void *A, int64_t lda, int32_t a_kmajor, const void *B, int64_t ldb, int32_t b_kmajor, int64_t b_shift,
                  void *C, int64_t ldc, int32_t epilogue, const float *bias, int32_t M, int32_t N, int64_t K, int32_t splits,
                  void *stream);

/*
 * The path itself.  Same model as above -- EEG_LSTM, lstm_eeg_model.py:13-39, with the ctor kwargs of :14 -- for hidden sizes
 * 64, 128, 256, 512 (L <= 8, F, K <= 64), optionally bidirectional (NSD_FLAG_BIDIR; where lstm_eeg_model.py:16-22 would take
 * torch's `bidirectional=True`), computed with bf16 GEMM operands, bf16 saved activations, fp32 accumulation and fp32 cell
 * state / gate arithmetic (BASELINE cfg3 / cfg5 precision; differs from an fp32 run at the 1e-2 level on logits).
 * Per layer: input projection over the whole sequence (one GEMM) -> persistent scan (recurrent weights resident in
 * registers, groups of H/32 workgroups exchanging h_t through the saved sequence) ; backward: persistent scan -> weight /
 * input gradient GEMMs.  x is the same [B,T,C] fp32 tensor as everywhere else; everything in between lives in `workspace`.
 *
 * Flat parameter vector: torch's state_dict order, i.e. per layer the four tensors of the forward direction, then (D = 2)
 * the four `_reverse` tensors; layer l > 0 has input width D*H; ln / attn / fc.0 act on D*H columns.
 *   nsd_seq_param_layout: offsets[4*L*D + 8]
 * Train-mode randomness: counter streams of `rng` as in nsd_lstm_head_train_rng (inter-layer dropout index
 * ((l*B + b)*T + t)*D*H + column, RReLU / head dropout index b*F + f); rng == NULL: no dropout, eval RReLU slope.
 *   nsd_seq_train_fwd   forward + head + mean CE (scale = 1/B_global) + head backward; logits[B,K] written
 *   nsd_seq_train_bwd   BPTT + all parameter gradients -> grads[P] (overwritten), same rng as the forward call
 *   nsd_seq_loss_sum    sum of the per-trial CE losses of the last nsd_seq_train_fwd -> out[0] (device)
 * Failure reporting (the persistent scan kernels assume that all workgroups of a scan group are resident at once -- true on an
 * MI355X this process has to itself; a CU mask, another process or a partition mode can break it -- and bound every wait: a group
 * that cannot assemble gives up after ~1-2 s).  A time-out is never silent:
 *   - the workspace starts with a persistent header whose first word is a STICKY status (OR of every time-out code; bit 0 a
 *     forward, bit 1 a backward scan).  nsd_seq_workspace_init zeroes it: call it once after allocating the workspace (an
 *     uninitialised header reads as a failure).  No forward / backward call ever clears it.
 *   - logits, probs and the per-trial loss of an evaluation on a workspace that reports a time-out are NaN.
 * Non-finite values (a NaN / Inf window, NaN / Inf or diverged weights) propagate as in the reference's torch.nn.LSTM
 * (lstm_eeg_model.py:34): the logits / probs / loss of the affected trials are NaN, the other trials of the batch are
 * untouched, gradients of a batch with such a trial are NaN.  They are reported as status bit 2 (value 4) of the evaluation
 * -- NOT sticky, no time-out, no waiting -- and nsd_seq_guard raises its flag for that step, so the guarded Adam update
 * does not write NaN into the parameters.
 *   nsd_seq_status      BLOCKING: status_out[4] = {code of the last evaluation OR the sticky word (0 = ok; bits 0 / 1 time-outs,
 *                       bit 2 non-finite activations in the last evaluation), the sticky word alone,
 *                       scan groups (over all scan launches since the last nsd_seq_train_fwd / nsd_seq_infer) whose workgroups
 *                       all reported ONE XCD, groups spread over several}
 *   nsd_seq_guard       enqueued: flag_out[0] (device fp32) = 1 if the workspace reports a time-out or non-finite activations, else 0.  Append it to the
 *                       gradient vector that is all-reduced and hand it to nsd_adam_step_guarded: every rank then skips the
 *                       update when any rank's gradient is garbage.
 */
int64_t nsd_seq_param_count(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F, int32_t D);
int     nsd_seq_param_layout(int32_t C, int32_t H, int32_t L, int32_t K, int32_t F, int32_t D, int64_t *offsets);
int     nsd_seq_supported(const nsd_dims *d, uint32_t flags);
int64_t nsd_seq_workspace_bytes(const nsd_dims *d, uint32_t flags);
int nsd_seq_infer(const nsd_dims *d, const float *params, const float *x, uint32_t flags, float *logits, float *probs,
                  void *workspace, int64_t workspace_bytes, void *stream);
int nsd_seq_train_fwd(const nsd_dims *d, const float *params, const float *x, const nsd_rng *rng, const int32_t *labels,
                      float scale, uint32_t flags, void *workspace, int64_t workspace_bytes, float *logits, void *stream);
int nsd_seq_train_bwd(const nsd_dims *d, const float *params, const nsd_rng *rng, uint32_t flags, void *workspace,
                      int64_t workspace_bytes, float *grads, void *stream);
int nsd_seq_loss_sum(const nsd_dims *d, uint32_t flags, const void *workspace, int64_t workspace_bytes, float *out, void *stream);
int nsd_seq_workspace_init(void *workspace, int64_t workspace_bytes, void *stream);
int nsd_seq_status(const void *workspace, int32_t *status_out, void *stream);
int nsd_seq_guard(const void *workspace, float *flag_out, void *stream);
#ifdef __cplusplus
}
#endif
#endif /* NSD_H */
