#!/bin/bash
# build both libraries; exit non-zero on any compile error (never ship a stale .so to the GPU box)
set -e -o pipefail
cd "$(dirname "$0")/../neural-speech-decoding_amd/csrc"
make -s -j4 2>&1 | grep -v "^\s*$" | grep -E "error:|warning:" -A4 || true
make -s -j4 >/dev/null 2>&1
make -s prof >/dev/null 2>&1
test ../libnsd_hip.so -nt nsd_lstm2_fwd48.hip
echo "build ok: $(ls -la ../libnsd_hip.so | awk '{print $6,$7,$8}')"
