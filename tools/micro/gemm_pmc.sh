#!/bin/bash
# counter passes over one GEMM shape (separate --pmc runs, nothing else traced):  bash tools/micro/gemm_pmc.sh xproj
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  d=$root/gpurun_out/gpmc_$(echo $set | cut -d' ' -f1)
  rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $root/tools/micro/gemm_one.py $1 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} {sum(v) / len(v):16.0f}  ({len(v)} launches)")
PY
done
