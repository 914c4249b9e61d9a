#!/bin/bash
# Real-data training runs on the packed recorded trials (tests/golden/recorded_trials.npz); logs -> gpurun_out/real/
set -e
mkdir -p gpurun_out/real
run() {  # name, extra args
  name=$1; shift
  rm -f gpurun_out/real/$name.jsonl
  python -m nsd_amd.train --data tests/golden/recorded_trials.npz --out gpurun_out/real/$name.pth --log-jsonl gpurun_out/real/$name.jsonl --log-every 10 "$@" > gpurun_out/real/$name.out 2>&1
  tail -1 gpurun_out/real/$name.jsonl | cut -c1-160
}
run c3_lr3e-3_b32   --classes 3 --epochs 120 --batch 32 --lr 0.003 --seed 1
run c3_lr1e-3_b32   --classes 3 --epochs 200 --batch 32 --lr 0.001 --seed 1
run c3_lr3e-3_b16_n --classes 3 --epochs 120 --batch 16 --lr 0.003 --seed 1 --normalize
run c3_lr3e-3_d03   --classes 3 --epochs 120 --batch 32 --lr 0.003 --seed 1 --dropout 0.3
run c3_lr3e-3_s2    --classes 3 --epochs 120 --batch 32 --lr 0.003 --seed 2
run c3_wd           --classes 3 --epochs 120 --batch 32 --lr 0.003 --seed 1 --weight-decay 0.01
