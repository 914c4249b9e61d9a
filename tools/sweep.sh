#!/bin/bash
# Build variants of one kernel file with -D switches and time them all in ONE gpurun call (box-to-box variation is ~1-5 %).
#   tools/sweep.sh nsd_lstm2_fwd48.hip "-DNSD_P_SLEEP=1 -DNSD_S_SLEEP=2" "-DNSD_P_SLEEP=3" ...
# Writes libnsd_hip_v<i>.so next to the shipped library and prints the command to run.
set -e
cd "$(dirname "$0")/../neural-speech-decoding_amd/csrc"
src=$1; shift
make -s -j4 >/dev/null 2>&1
i=0
objs="nsd_abi.o nsd_lstm2.o nsd_lstm2_fwd48.o nsd_lstm2_bwd48.o nsd_lstm_generic.o nsd_lstm_batched.o nsd_head.o nsd_misc.o"
libs="libnsd_hip.so"
for flags in "$@"; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $flags -c $src -o /tmp/sweep_$i.o
    o=$(echo $objs | sed "s#${src%.hip}.o#/tmp/sweep_$i.o#")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnsd_hip_v$i.so $o
    echo "v$i: $flags"
    libs="$libs libnsd_hip_v$i.so"
    i=$((i+1))
done
echo "run: gpurun -- 'for l in $libs libnsd_hip.so; do echo \$l; NSD_LIB=\$l python tools/kbench.py | grep ablate=; done'"
