#!/bin/bash
# rocprofv3 passes for profiles/: (1) kernel trace + stats of the default bench, (2) FETCH_SIZE, (3) WRITE_SIZE
# (separate --pmc passes: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md).
# Usage on the GPU box:  bash tools/pmc_run.sh <tag>
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-kernel-timing > $OUT.trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT.write.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $F > $OUT.gaps.txt
cat $OUT.gaps.txt
