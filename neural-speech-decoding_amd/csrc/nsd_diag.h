// nsd_diag.h -- what only the DIAGNOSTIC build of the library has (make diag -> libnsd_hip_diag.so, compiled with -DNSD_DIAG=1).
// None of this is part of include/nsd.h or of libnsd_hip.so: the product library rejects these flag bits (NSD_E_INVALID) and
// does not export the entry points.  Users: tests/ (exchange-mode agreement, forced time-out), bench.py's per-kernel timing
// loop (outside the timed region), tools/.
#pragma once
#include <stdint.h>

#ifndef NSD_DIAG
#define NSD_DIAG 0
#endif

// nsd_seq_* flag bits, honoured by the diagnostic build only
#define NSD_DIAG_FLAG_NO_L2_EXCHANGE  16u   // scan groups always use the write-through exchange, also when all their workgroups report one XCD
#define NSD_DIAG_FLAG_SPREAD_GROUPS   32u   // consecutive block ids per group: every group spread over all XCDs (exercises write-through for real)
#define NSD_DIAG_FLAG_NO_FUSED_LAYERS 64u   // two unidirectional layers as two scans + GEMMs (the general route) instead of the skewed launch
#define NSD_DIAG_FLAG_LOSE_MEMBER    128u   // every scan launch misses its last workgroup: its group must time out (bounded spin) and report it
#define NSD_DIAG_FLAG_ALL (NSD_DIAG_FLAG_NO_L2_EXCHANGE | NSD_DIAG_FLAG_SPREAD_GROUPS | NSD_DIAG_FLAG_NO_FUSED_LAYERS | NSD_DIAG_FLAG_LOSE_MEMBER)

#if NSD_DIAG
extern "C" {
// Opt-in launch timing: nsd_seq_profile(1) records HIP events on the launch stream around the kernels of every following
// nsd_seq_* call, nsd_seq_profile(0) stops and discards; nsd_seq_profile_read sums the records of one kind and forgets them
// (BLOCKING).  kind: 0 forward scan, 1 backward scan, 2 input-projection GEMM, 3 weight-gradient GEMMs, 4 input-gradient GEMM,
// 5 head, 6 head parameter gradients, 7 operand preparation.
int nsd_seq_profile(int32_t enable);
int nsd_seq_profile_read(int32_t kind, float *total_ms, int32_t *count);
// Pin the H = 48 forward instantiation of the fp32 fast path: 1 / 2 / 4 trials per workgroup (4 = nsd_lstm2_fwd48x4.hip where it
// applies), 0 = the product's own choice.  Process-wide; tests compare the instantiations on the same inputs.
int nsd_diag_force_fwd48(int32_t nb);       // 0 = the product's choice, 1 / 2 / 4 trials per workgroup, 8 = the experimental one-wave-per-layer kernel (nsd_lstm2_fwd48w.hip: unfused training launches)
int nsd_diag_force_bwd48(int32_t nb);       // 0 = the product's choice, 2 = nsd_lstm2_bwd48.hip, 4 = nsd_lstm2_bwd48x4.hip where it applies
}
#endif
