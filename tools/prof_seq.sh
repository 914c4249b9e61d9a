#!/bin/bash
# rocprofv3 kernel stats of tools/seq_step.py; usage: tools/prof_seq.sh <tag> [seq_step args]   (run on the GPU box via gpurun)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/tools/seq_step.py --iters 3 "$@" > $root/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$root/gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:8.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} {r['Percentage']}%")
PY
