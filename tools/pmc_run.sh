#!/bin/bash
# rocprofv3 passes for profiles/: (1) kernel trace + stats of the bench, (2) FETCH_SIZE, (3) WRITE_SIZE
# (separate --pmc passes: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md; no other tracing with --pmc).
# Usage on the GPU box:  bash tools/pmc_run.sh <tag> [config]      then, in the build container: python tools/pmc_summarize.py <tag> [config]
set -e
TAG=${1:-r02}
CFG=${2:-cfg2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "cfg2" ] || [ "$CFG" = "cfg4" ]; then LONG="--steps 200 --warmup 50"; SHORT="--steps 4 --warmup 2 --preheat-steps 4"; else LONG="--steps 40 --warmup 5"; SHORT="--steps 2 --warmup 1 --preheat-steps 1"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config $CFG $LONG --no-cpu-baseline --no-kernel-timing > $OUT.trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --config $CFG $SHORT --no-cpu-baseline --no-kernel-timing > $OUT.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --config $CFG $SHORT --no-cpu-baseline --no-kernel-timing > $OUT.write.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_gaps.py $F > $OUT.gaps.txt || true
tail -3 $OUT.gaps.txt
tail -1 $OUT.trace.log | cut -c1-300
