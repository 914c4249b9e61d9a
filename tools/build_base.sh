#!/bin/bash
# A/B helper: build the kernels of a git ref (default HEAD) as libnsd_hip_base.so next to the working-tree library, so that
# both can be timed in ONE gpurun call (box-to-box variation is larger than most single kernel changes):
#   tools/build.sh && tools/build_base.sh && gpurun -- 'for l in libnsd_hip_base.so libnsd_hip.so libnsd_hip_base.so libnsd_hip.so; do NSD_LIB=$l python tools/kbench.py; done'
set -e -o pipefail
ref="${1:-HEAD}"
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp="$(mktemp -d)"
git -C "$root" archive "$ref" neural-speech-decoding_amd/csrc include | tar -x -C "$tmp"
make -s -j4 -C "$tmp/neural-speech-decoding_amd/csrc" >/dev/null 2>&1
cp "$tmp/neural-speech-decoding_amd/libnsd_hip.so" "$root/neural-speech-decoding_amd/libnsd_hip_base.so"
rm -rf "$tmp"
echo "base lib built from $ref"
