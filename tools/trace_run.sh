#!/bin/bash
# kernel trace of a long bench run (steady clocks), then durations + gaps: bash tools/trace_run.sh <tag>
set -e
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-kernel-timing > $OUT.trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py $F > $OUT.gaps.txt
cat $OUT.gaps.txt
tail -1 $OUT.trace.log | cut -c1-200
