// nsd_misc.hip -- small memory-bound kernels around the LSTM path: per-channel z-score, slab reduction,
// Adam, counter-based dropout / RReLU-noise streams, loss sum.
#include "nsd_args.h"

// ---------------------------------------------------------------------------------------------
// z-score: y = (x - mean_T) / (std_T(ddof=0) + 1e-6) per trial and channel.
// Replaces normalize_eeg (Neuro-Alpha-App/Frontend/app.py:166-170).  One 256-thread workgroup per trial;
// lanes run along the contiguous [t][c] axis (coalesced), each thread owns channel tid % C.
// ---------------------------------------------------------------------------------------------
#define ZS_NT 256
__global__ __launch_bounds__(ZS_NT) void zscore_kernel(const float *x, float *y, int B, int T, int C) {
    extern __shared__ float zl[];          // [ZS_NT] partials + [2*C] stats
    float *stat = zl + ZS_NT;
    const int tid = threadIdx.x;
    const int act = (ZS_NT / C) * C;       // threads taking part; each keeps one channel
    const int ch = tid % C;
    const int rows = act / C;              // time steps covered per sweep
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float *xb = x + (size_t)b * T * C;
        float *yb = y + (size_t)b * T * C;
        float s = 0.f;
        if (tid < act) for (int t = tid / C; t < T; t += rows) s += xb[(size_t)t * C + ch];
        zl[tid] = s;
        __syncthreads();
        if (tid < C) { float m = 0.f; for (int q = 0; q < rows; ++q) m += zl[q * C + tid]; stat[tid] = m / (float)T; }
        __syncthreads();
        const float mu = stat[ch];
        float v = 0.f;
        if (tid < act) for (int t = tid / C; t < T; t += rows) { const float d = xb[(size_t)t * C + ch] - mu; v = fmaf(d, d, v); }
        __syncthreads();
        zl[tid] = v;
        __syncthreads();
        if (tid < C) { float m = 0.f; for (int q = 0; q < rows; ++q) m += zl[q * C + tid]; stat[C + tid] = 1.0f / (sqrtf(m / (float)T) + 1e-6f); }
        __syncthreads();
        const float rs = stat[C + ch];
        if (tid < act) for (int t = tid / C; t < T; t += rows) yb[(size_t)t * C + ch] = (xb[(size_t)t * C + ch] - mu) * rs;
        __syncthreads();
    }
}

int nsd_zscore_launch(const float *x, float *y, int B, int T, int C, hipStream_t st) {
    if (B <= 0) return NSD_OK;
    if (C < 1 || C > ZS_NT || T < 1) { nsd_set_error("zscore: bad shape T=%d C=%d", T, C); return NSD_E_INVALID; }
    const int cap = 8 * nsd_num_cus();
    hipLaunchKernelGGL(zscore_kernel, dim3(B < cap ? B : cap), dim3(ZS_NT), (ZS_NT + 2 * C) * sizeof(float), st, x, y, B, T, C);
    NSD_CHECK_LAUNCH("zscore");
    return NSD_OK;
}

// ---------------------------------------------------------------------------------------------
// gradient reduction: grads[e] (+)= sum over the per-workgroup LSTM slabs (e < P_lstm) or over the
// per-trial head slabs (e >= P_lstm).  Column sums: consecutive threads read consecutive floats.
// ---------------------------------------------------------------------------------------------
// One workgroup = 32 consecutive gradient entries x 8 slab groups: lane (c = tid&31, grp = tid>>5) sums its share
// of the slabs for column c (consecutive lanes read 128 contiguous bytes of a slab row), the 8 partials meet
// in LDS in a fixed order (deterministic).  ~1000 workgroups for the reference model instead of 124.
#ifndef GR_COLS
#define GR_COLS 32
#endif
#ifndef GR_GROUPS
#define GR_GROUPS 8
#endif
#ifndef GR_UNROLL
#define GR_UNROLL 8
#endif
// optional optimizer tail of the reduction (single-rank training: no all-reduce sits between the two)
struct AdamTail { float *p, *m, *v; float lr_over_bc1, rsqrt_bc2, beta1, beta2, eps, wd, gscale; };

template <bool ADAM>
__global__ __launch_bounds__(GR_COLS * GR_GROUPS) void grad_reduce_kernel(
        const float *slabs, long slab_stride, int n_slabs, long p_lstm, const float *hslabs, long ph, int n_hslabs,
        float *grads, int accumulate, AdamTail ad) {
    __shared__ float part[GR_GROUPS][GR_COLS];
    const int c = threadIdx.x & (GR_COLS - 1), grp = threadIdx.x / GR_COLS;
    const long e = (long)blockIdx.x * GR_COLS + c;
    float s0 = 0.f, s1 = 0.f;
    if (e < p_lstm + ph) {
        const float *p; long stride; int n;
        if (e < p_lstm) { p = slabs + e; stride = slab_stride; n = n_slabs; }
        else { p = hslabs + (e - p_lstm); stride = ph; n = n_hslabs; }
        // GR_UNROLL independent loads in flight per thread; fixed association order -> deterministic.  (Measured: 8, 16
        // and 32 in flight all run at ~4.9 TB/s for the 32.6 MB of slabs -- the sum sits at the memory roof.)
        int q = grp;
        float acc[GR_UNROLL];
#pragma unroll
        for (int u = 0; u < GR_UNROLL; ++u) acc[u] = 0.f;
        for (; q + (GR_UNROLL - 1) * GR_GROUPS < n; q += GR_UNROLL * GR_GROUPS) {
            float vload[GR_UNROLL];
#pragma unroll
            for (int u = 0; u < GR_UNROLL; ++u) vload[u] = p[(size_t)(q + u * GR_GROUPS) * stride];
#pragma unroll
            for (int u = 0; u < GR_UNROLL; ++u) acc[u] += vload[u];
        }
        for (; q < n; q += GR_GROUPS) acc[0] += p[(size_t)q * stride];
#pragma unroll
        for (int w = GR_UNROLL / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
        s0 = acc[0];
    }
    part[grp][c] = s0 + s1;
    __syncthreads();
    if (grp == 0 && e < p_lstm + ph) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < GR_GROUPS; ++g) s += part[g][c];
        if (accumulate) s += grads[e];
        grads[e] = s;
        if (ADAM) {                                   // same arithmetic, in the same order, as adam_kernel
            const float pi = ad.p[e];
            const float gi = fmaf(ad.wd, pi, s * ad.gscale);
            const float mi = ad.beta1 * ad.m[e] + (1.f - ad.beta1) * gi;
            const float vi = ad.beta2 * ad.v[e] + (1.f - ad.beta2) * gi * gi;
            ad.m[e] = mi; ad.v[e] = vi;
            const float denom = sqrtf(vi) * ad.rsqrt_bc2 + ad.eps;
            ad.p[e] = pi - ad.lr_over_bc1 * (mi / denom);
        }
    }
}

int nsd_grad_reduce_launch(const float *slabs, long slab_stride, int n_slabs, long p_lstm, const float *hslabs,
                           long ph, int n_hslabs, float *grads, int accumulate, hipStream_t st) {
    const long n = p_lstm + ph;
    hipLaunchKernelGGL((grad_reduce_kernel<false>), dim3((unsigned)((n + GR_COLS - 1) / GR_COLS)), dim3(GR_COLS * GR_GROUPS), 0, st,
                       slabs, slab_stride, n_slabs, p_lstm, hslabs, ph, n_hslabs, grads, accumulate, AdamTail{});
    NSD_CHECK_LAUNCH("grad_reduce");
    return NSD_OK;
}

int nsd_grad_reduce_adam_launch(const float *slabs, long slab_stride, int n_slabs, long p_lstm, const float *hslabs,
                                long ph, int n_hslabs, float *grads, float *p, float *m, float *v, float lr, float b1,
                                float b2, float eps, float wd, float gscale, int step, hipStream_t st) {
    const long n = p_lstm + ph;
    if (step < 1) { nsd_set_error("grad_reduce_adam: step must be >= 1"); return NSD_E_INVALID; }
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    AdamTail ad{p, m, v, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), b1, b2, eps, wd, gscale};
    hipLaunchKernelGGL((grad_reduce_kernel<true>), dim3((unsigned)((n + GR_COLS - 1) / GR_COLS)), dim3(GR_COLS * GR_GROUPS), 0, st,
                       slabs, slab_stride, n_slabs, p_lstm, hslabs, ph, n_hslabs, grads, 0, ad);
    NSD_CHECK_LAUNCH("grad_reduce_adam");
    return NSD_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, amsgrad=False).  g is multiplied by grad_scale first (1/world_size after a
// SUM all-reduce).
// ---------------------------------------------------------------------------------------------
// skip: null, or a device flag -- a non-zero value leaves p, m, v untouched (nsd_adam_step_guarded: the gradient is known to be
// garbage, e.g. a scan group of the sequence-batched path timed out on some rank)
__global__ void adam_kernel(long n, float *p, const float *g, float *m, float *v, float lr_over_bc1, float rsqrt_bc2,
                            float beta1, float beta2, float eps, float wd, float gscale, const float *skip) {
    if (skip != nullptr && skip[0] != 0.f) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i] * gscale);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        p[i] = pi - lr_over_bc1 * (mi / denom);
    }
}

// Device-side step counter variants (hipGraph replay: nothing that changes from step to step may be a kernel
// argument).  step_dev[0] is the 1-based step number; bias corrections are formed in double like on the host.
__global__ void adam_dev_kernel(long n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                                float eps, float wd, float gscale, const long long *step_dev) {
    const double st = (double)step_dev[0];
    const double bc1 = 1.0 - pow((double)beta1, st), bc2 = 1.0 - pow((double)beta2, st);
    const float lr_over_bc1 = (float)((double)lr / bc1), rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i] * gscale);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        p[i] = pi - lr_over_bc1 * (mi / denom);
    }
}
int nsd_adam_dev_launch(long n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps,
                        float wd, float gscale, const long long *step_dev, hipStream_t st) {
    if (n <= 0) return NSD_OK;
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, p, g, m, v, lr, b1, b2, eps, wd, gscale, step_dev);
    NSD_CHECK_LAUNCH("adam_dev");
    return NSD_OK;
}
__global__ void step_inc_kernel(long long *step_dev) { if (threadIdx.x == 0 && blockIdx.x == 0) step_dev[0] += 1; }
int nsd_step_inc_launch(long long *step_dev, hipStream_t st) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, st, step_dev);
    NSD_CHECK_LAUNCH("step_inc");
    return NSD_OK;
}

__global__ void seq_guard_kernel(const int *header, int status_word, float *flag_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) flag_out[0] = (header[0] | header[status_word]) != 0 ? 1.f : 0.f;
}
int nsd_seq_guard_launch(const int *header, int status_word, float *flag_out, hipStream_t st) {
    hipLaunchKernelGGL(seq_guard_kernel, dim3(1), dim3(64), 0, st, header, status_word, flag_out);
    NSD_CHECK_LAUNCH("seq_guard");
    return NSD_OK;
}

int nsd_adam_launch(long n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2, float eps,
                    float wd, float gscale, int step, const float *skip, hipStream_t st) {
    if (n <= 0) return NSD_OK;
    if (step < 1) { nsd_set_error("adam: step must be >= 1"); return NSD_E_INVALID; }
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, p, g, m, v, (float)(lr / bc1),
                       (float)(1.0 / sqrt(bc2)), b1, b2, eps, wd, gscale, skip);
    NSD_CHECK_LAUNCH("adam");
    return NSD_OK;
}

// ---------------------------------------------------------------------------------------------
// counter-based random streams (bit-identical to oracle/nsd_oracle.c)
// ---------------------------------------------------------------------------------------------
__global__ void dropout_mask_kernel(uint64_t seed, uint32_t stream_id, uint32_t thr, float keep, long n, float *out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = nsd_rand_u32(seed, stream_id, (uint64_t)i) >= thr ? keep : 0.f;
}
__global__ void rrelu_noise_kernel(uint64_t seed, uint32_t stream_id, long n, float *out) {
    const float lower = 0.125f, upper = (float)(1.0 / 3.0);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float u = (float)(nsd_rand_u32(seed, stream_id, (uint64_t)i) >> 8) * (1.0f / 16777216.0f);
        out[i] = lower + (upper - lower) * u;
    }
}

// all three train-mode streams of one step in ONE launch (stream ids base, base+1, base+2: bit-identical to
// the separate calls): LSTM inter-layer dropout [n_lstm], RReLU slopes [n_head], head dropout [n_head]
__global__ void train_masks_kernel(uint64_t seed, uint32_t base_arg, const long long *step_dev, uint32_t thr_lstm, float keep_lstm, uint32_t thr_head,
                                   float keep_head, long n_lstm, float *drop_lstm, long n_head, float *rrelu, float *drop_head) {
    const float lower = 0.125f, upper = (float)(1.0 / 3.0);
    // stream ids of this step: explicit, or 4 * (device step counter) as nsd_amd.trainer numbers them
    const uint32_t base = step_dev ? (uint32_t)(step_dev[0] & 0x3FFFFFFF) * 4u : base_arg;
    const long total = n_lstm + 2 * n_head;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < n_lstm) {
            drop_lstm[i] = nsd_rand_u32(seed, base, (uint64_t)i) >= thr_lstm ? keep_lstm : 0.f;
        } else if (i < n_lstm + n_head) {
            const long k = i - n_lstm;
            const float u = (float)(nsd_rand_u32(seed, base + 1u, (uint64_t)k) >> 8) * (1.0f / 16777216.0f);
            rrelu[k] = lower + (upper - lower) * u;
        } else {
            const long k = i - n_lstm - n_head;
            drop_head[k] = nsd_rand_u32(seed, base + 2u, (uint64_t)k) >= thr_head ? keep_head : 0.f;
        }
    }
}
static uint32_t drop_threshold(float p) { return nsd_drop_threshold(p); }
int nsd_train_masks_launch(uint64_t seed, uint32_t base, const long long *step_dev, float p_lstm, float p_head, long n_lstm,
                           float *drop_lstm, long n_head, float *rrelu, float *drop_head, hipStream_t st) {
    if (!(p_lstm >= 0.f && p_lstm < 1.f) || !(p_head >= 0.f && p_head < 1.f)) { nsd_set_error("train_masks: p out of [0,1)"); return NSD_E_INVALID; }
    const long total = n_lstm + 2 * n_head;
    if (total <= 0) return NSD_OK;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(train_masks_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, base, step_dev, drop_threshold(p_lstm),
                       1.0f / (1.0f - p_lstm), drop_threshold(p_head), 1.0f / (1.0f - p_head), n_lstm, drop_lstm, n_head, rrelu, drop_head);
    NSD_CHECK_LAUNCH("train_masks");
    return NSD_OK;
}

int nsd_dropout_mask_launch(uint64_t seed, uint32_t stream_id, float p, long n, float *out, hipStream_t st) {
    if (n <= 0) return NSD_OK;
    if (!(p >= 0.f && p < 1.f)) { nsd_set_error("dropout: p=%f out of [0,1)", p); return NSD_E_INVALID; }
    double t = (double)p * 4294967296.0; if (t > 4294967295.0) t = 4294967295.0;
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, stream_id, (uint32_t)t,
                       1.0f / (1.0f - p), n, out);
    NSD_CHECK_LAUNCH("dropout_mask");
    return NSD_OK;
}
int nsd_rrelu_noise_launch(uint64_t seed, uint32_t stream_id, long n, float *out, hipStream_t st) {
    if (n <= 0) return NSD_OK;
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rrelu_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, st, seed, stream_id, n, out);
    NSD_CHECK_LAUNCH("rrelu_noise");
    return NSD_OK;
}

// ---------------------------------------------------------------------------------------------
// loss sum (deterministic single workgroup)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_sum_kernel(const float *loss, int B, float *out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += loss[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}
// ------------------------------------------------------------------------------------------------
// Input gradient of the stack: dL/dx[b][t][c] = sum_k da0[b][t][k] W_ih0[k][c] (what autograd through self.lstm(x) returns for x,
// lstm_eeg_model.py:34).  Off the training path (the parameter gradients do not need it): a plain kernel, one wave per 64 rows,
// the weight matrix staged in LDS, each lane one row x all channels.
// ------------------------------------------------------------------------------------------------
constexpr int DX_CMAX = 64;
__global__ __launch_bounds__(256) void dx_kernel(const float *da0, const float *w_ih0, float *dx, long rows, int G4, int C) {
    extern __shared__ __align__(16) float wl[];                    // [G4][C]
    for (int e = threadIdx.x; e < G4 * C; e += 256) wl[e] = w_ih0[e];
    __syncthreads();
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *dr = da0 + r * G4;
    for (int c0 = 0; c0 < C; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = 0; k < G4; ++k) {
            const float v = dr[k];
#pragma unroll
            for (int i = 0; i < 8; ++i) if (c0 + i < C) acc[i] = fmaf(v, wl[k * C + c0 + i], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) if (c0 + i < C) dx[r * C + c0 + i] = acc[i];
    }
}
int nsd_dx_launch(const float *da0, const float *w_ih0, float *dx, long rows, int G4, int C, hipStream_t st) {
    if (rows <= 0) return NSD_OK;
    // (the staged weight matrix: dynamic LDS without an opt-in attribute)
    if (C > DX_CMAX || (long)G4 * C * 4 > 64 * 1024) { nsd_set_error("dx: C = %d, 4H = %d outside the kernel's domain", C, G4); return NSD_E_INVALID; }
    hipLaunchKernelGGL(dx_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), (size_t)G4 * C * 4, st, da0, w_ih0, dx, rows, G4, C);
    NSD_CHECK_LAUNCH("dx");
    return NSD_OK;
}

int nsd_loss_sum_launch(const float *loss, int B, float *out, hipStream_t st) {
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(256), 0, st, loss, B, out);
    NSD_CHECK_LAUNCH("loss_sum");
    return NSD_OK;
}
