#!/bin/bash
# GPU box: WRITE_SIZE / FETCH_SIZE per variant of tools/micro/l2_writeback (separate --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf $ROOT/gpurun_out/l2wb_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/l2wb_$c -- $ROOT/tools/micro/l2_writeback > $ROOT/gpurun_out/l2wb_$c.log 2>&1
done
python3 - <<PY
import csv, glob
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob("$ROOT/gpurun_out/l2wb_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            print(c, r["Kernel_Name"][:60], "%.1f MB" % (float(r["Counter_Value"]) / 1024))
PY
