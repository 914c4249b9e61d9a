#!/bin/bash
# Build variants of nsd_scan2.hip (-D switches) as libnsd_hip_var_<tag>.so next to the product library (never shipped; `make clean` removes them).
#   tools/micro/scan2_variants.sh tagA "-DNSD_LOOK_POS=0" tagB "-DNSD_LOOK_POS=2 -DNSD_LOOK_DELAY=4" ...
set -e
cd "$(dirname "$0")/../../neural-speech-decoding_amd/csrc"
make -s
OBJS=$(ls *.o | grep -v '^diag_\|^stamps_\|^prof_\|^abl_\|^var_' | grep -v nsd_scan2.o)
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -I../../include $defs -c nsd_scan2.hip -o var_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnsd_hip_var_$tag.so $OBJS var_$tag.o
  echo built libnsd_hip_var_$tag.so "($defs)"
done
