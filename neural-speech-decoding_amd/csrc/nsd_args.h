// nsd_args.h -- kernel argument blocks and host launchers shared by the .hip files of libnsd_hip.so.
#pragma once
#include "nsd_common.h"

struct Lstm2FwdArgs {
    const float *x;
    const float *w_ih0, *w_hh0, *b_ih0, *b_hh0, *w_ih1, *w_hh1, *b_ih1, *b_hh1;
    const float *mask;
    float *hseq0, *hseq1, *cseq0, *cseq1, *gact0, *gact1, *inseq, *top;
    long long *dbg;                          // diagnostic build only (see nsd_prof.h)
    // fused inference tail (lstm2_fwd48 only; logits_out != null): attention pooling over time as an online softmax,
    // LayerNorm, dense head and class softmax inside the LSTM kernel -- the [B,T,H] sequence never leaves the chip
    const float *attn_w, *attn_b, *ln_w, *ln_b, *fc0_w, *fc0_b, *fc3_w, *fc3_b;
    float *logits_out, *probs_out;
    float eval_slope;
    int K, F;
    int B, T, C, residual;
    int ablate;                              // timing experiments only (env NSD_ABLATE); 0 in production
    // fused TRAIN head (lstm2_fwd48 only; head_train != 0): attention pooling rides along the recurrence (wave 10), then
    // LayerNorm, dense head, mean-CE and the head's backward run in the kernel's tail -- what nsd_head_train does in a
    // second launch.  Outputs and per-trial gradient slabs are those of HeadArgs.
    int head_train;
    // lstm2_fwd48x4 only, set by the launcher when the backward pass of this batch runs lstm2_bwd48x4_kernel: the per-step part of the
    // attention's backward (dL/dscore_t, d attn.weight, d attn.bias) is left to that kernel, which reads the top rows anyway; the
    // forward writes alpha_t and marks the {alpha, dscore} records as open ({alpha_t, 0, 1, 0}: see Lstm2BwdArgs::dsc_pack)
    int defer_att;
    const int32_t *labels;
    const float *rrelu_slope, *drop_head;
    float scale;
    float *logits, *loss, *alpha, *pooled, *fc0_pre, *dscore, *dpooled, *adpack, *hslabs;
    long o_ln_w, o_ln_b, o_attn_w, o_attn_b, o_fc0_w, o_fc0_b, o_fc3_w, o_fc3_b, Ph;
    RngArgs rng;                             // rng.on: dropout multipliers / RReLU slopes are generated in the kernel
};
struct Lstm2BwdArgs {
    const float *x;
    const float *w_hh0, *w_ih1, *w_hh1, *attn_w;
    const float *mask;
    const float *hseq0, *hseq1, *cseq0, *cseq1, *gact0, *gact1;
    const float *in1seq;
    const float *alpha, *dscore, *dpooled;
    const float *dsc_pack;                   // [B,T,4] {alpha, dscore, 0, 0}: 16-byte records for LDS-DMA
                                             // lstm2_bwd48x4: a record {alpha, -, 1, -} is OPEN (left by lstm2_fwd48x4 with defer_att): the kernel
                                             // forms dscore_t = alpha_t dpooled . (top_t - pooled) itself, writes it to dscore_out and adds
                                             // d attn.weight / d attn.bias of the trial to its head slab
    const float *pooled;                     // [B,H] (open records only)
    float *dscore_out;                       // [B,T]
    float *da0_out;                          // null, or [B,T,4H]: the layer-0 pre-activation gradients da0[b][t][gate * H + unit] for the input
                                             // gradient dx = da0 . W_ih0 (lstm2_bwd48_kernel<1> only; may alias gact0: a step's saved gates
                                             // have been staged a chunk earlier)
    float *hslabs;                           // per-trial head-gradient slabs (stride Ph), offsets of attn.weight / attn.bias in them
    long Ph, o_attn_w, o_attn_b;
    float *slabs;
    long slab_stride;
    long o_w_ih0, o_w_hh0, o_b_ih0, o_b_hh0, o_w_ih1, o_w_hh1, o_b_ih1, o_b_hh1;
    long long *dbg;                          // diagnostic build of the schedule: per-wave {work, wait} cycle sums of workgroup 0 (null = off)
    int B, T, C, residual;
    int ablate;                              // timing experiments only (env NSD_ABLATE); 0 in production
    RngArgs rng;                             // rng.on: the inter-layer dropout multipliers are generated in the kernel
};
struct HeadArgs {
    const float *top;
    const float *ln_w, *ln_b, *attn_w, *attn_b, *fc0_w, *fc0_b, *fc3_w, *fc3_b;
    const float *rrelu_slope, *drop_head;
    float eval_slope;
    float *logits, *probs;
    float *alpha, *pooled, *fc0_pre;
    const float *logits_in, *dlogits;
    const int32_t *labels;
    float scale;
    float *loss, *dscore, *dpooled;
    float *adpack;                           // [B,T,4] {alpha, dscore, 0, 0} (backward only, may be null)
    float *hslabs;
    long o_ln_w, o_ln_b, o_attn_w, o_attn_b, o_fc0_w, o_fc0_b, o_fc3_w, o_fc3_b;
    long Ph;
    int stage_stride;                        // set by the launcher: LDS row stride of the staged sequence (0 = read from HBM)
    int B, T, H, F, K;
};
int nsd_lstm2_fwd_launch(const Lstm2FwdArgs &a, int H, hipStream_t st);
int nsd_lstm2_bwd_launch(const Lstm2BwdArgs &a, int H, hipStream_t st);
int nsd_lstm2_bwd_grid(int B);
int nsd_lstm2_bwd48_launch(const Lstm2BwdArgs &a, int nb, int grid, hipStream_t st);
int nsd_lstm2_fwd48_launch(const Lstm2FwdArgs &a, int nb, int grid, hipStream_t st);
bool nsd_lstm2_fwd48_head_train_fits(int T, int F, int K);
// four trials per workgroup, gate products on the matrix pipe (nsd_lstm2_fwd48x4.hip): training launches of the plain stack
bool nsd_lstm2_fwd48x4_ok(const Lstm2FwdArgs &a);
int nsd_lstm2_fwd48x4_launch(const Lstm2FwdArgs &a, int grid, hipStream_t st);
// EXPERIMENTAL one-wave-per-layer forward (nsd_lstm2_fwd48w.hip): diagnostic twin only (nsd_diag_force_fwd48(8))
bool nsd_lstm2_fwd48w_ok(const Lstm2FwdArgs &a);
int nsd_lstm2_fwd48w_launch(const Lstm2FwdArgs &a, int grid, hipStream_t st);
bool nsd_lstm2_bwd48x4_ok(const Lstm2BwdArgs &a);
int nsd_lstm2_bwd48x4_launch(const Lstm2BwdArgs &a, int grid, hipStream_t st);
int nsd_lstm_generic_fwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, float *hseq, float *cseq, float *gact, float *inseq, float *top_out, float *scratch2,
                         hipStream_t st);
int nsd_lstm_generic_bwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, const float *hseq, const float *cseq, const float *gact, const float *inseq,
                         const float *alpha, const float *dscore, const float *dpooled, float *da_seq, float *din_a, float *din_b,
                         float *slab, hipStream_t st);
bool nsd_lstm_batched_ok(const nsd_dims *d, bool training);
int nsd_lstm_batched_infer(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, float *top_out,
                           float *scratch2, float *cstate, bool bf16, hipStream_t st);
int nsd_lstm_batched_fwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, float *hseq, float *cseq, float *gact, float *inseq, float *top_out, bool bf16, hipStream_t st);
int nsd_lstm_batched_bwd(const nsd_dims *d, const ParamLayout &pl, const float *params, const float *x, const float *drop_lstm,
                         int residual, const float *hseq, const float *cseq, const float *gact, const float *inseq,
                         const float *alpha, const float *dscore, const float *dpooled, float *da_seq, float *din_a, float *din_b,
                         float *state, float *slab, bool bf16, hipStream_t st);
int nsd_head_launch(const HeadArgs &a, bool bwd, hipStream_t st);
int nsd_head_train_launch(const HeadArgs &a, hipStream_t st);   // 1 launched, 0 shape does not fit, <0 error
int nsd_zscore_launch(const float *x, float *y, int B, int T, int C, hipStream_t st);
int nsd_grad_reduce_launch(const float *slabs, long slab_stride, int n_slabs, long p_lstm, const float *hslabs,
                           long ph, int n_hslabs, float *grads, int accumulate, hipStream_t st);
int nsd_grad_reduce_adam_launch(const float *slabs, long slab_stride, int n_slabs, long p_lstm, const float *hslabs,
                                long ph, int n_hslabs, float *grads, float *p, float *m, float *v, float lr, float b1,
                                float b2, float eps, float wd, float gscale, int step, hipStream_t st);
int nsd_adam_launch(long n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps,
                    float wd, float gscale, int step, const float *skip, hipStream_t st);
int nsd_seq_guard_launch(const int *header, int status_word, float *flag_out, hipStream_t st);
int nsd_dropout_mask_launch(uint64_t seed, uint32_t stream_id, float p, long n, float *out, hipStream_t st);
int nsd_train_masks_launch(uint64_t seed, uint32_t base, const long long *step_dev, float p_lstm, float p_head, long n_lstm,
                           float *drop_lstm, long n_head, float *rrelu, float *drop_head, hipStream_t st);
int nsd_adam_dev_launch(long n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps,
                        float wd, float gscale, const long long *step_dev, hipStream_t st);
int nsd_step_inc_launch(long long *step_dev, hipStream_t st);
int nsd_rrelu_noise_launch(uint64_t seed, uint32_t stream_id, long n, float *out, hipStream_t st);
int nsd_loss_sum_launch(const float *loss, int B, float *out, hipStream_t st);
// dx[r][c] = sum_k da0[r][k] * w_ih0[k][c]: r = (trial, step), k = gate * H + unit (nn.LSTM weight_ih_l0 is [4H][C] row-major)
int nsd_dx_launch(const float *da0, const float *w_ih0, float *dx, long rows, int G4, int C, hipStream_t st);

