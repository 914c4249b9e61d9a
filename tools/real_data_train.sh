#!/bin/bash
# Real-data training runs (GPU box): 5-fold cross-validation on the recorded 3-class trials AS THE REFERENCE'S PreProcessor HANDS THEM TO
# THE MODEL (tests/golden/recorded_trials_filtered.npz, x_filt), a few recipes; logs -> gpurun_out/real/.  The recipe with the best
# k-fold mean is the one whose all-trials model is shipped at neural-speech-decoding_amd/LSTM_Model/ (copy it by hand).
set -e
mkdir -p gpurun_out/real
run() {  # name, extra args
  name=$1; shift
  rm -f gpurun_out/real/$name.jsonl
  python -m nsd_amd.train --data tests/golden/recorded_trials_filtered.npz --npz-key x_filt --classes 3 --kfold 5 --out gpurun_out/real/$name.pth \
      --log-jsonl gpurun_out/real/$name.jsonl --log-every 20 "$@" > gpurun_out/real/$name.out 2>&1
  tail -1 gpurun_out/real/$name.jsonl | cut -c1-220
}
run f_lr3e-3_b32_e120   --epochs 120 --batch 32 --lr 0.003 --seed 1
run f_lr1e-3_b32_e200   --epochs 200 --batch 32 --lr 0.001 --seed 1
run f_lr3e-3_b16_e100   --epochs 100 --batch 16 --lr 0.003 --seed 1
run f_lr3e-3_b32_e120_n --epochs 120 --batch 32 --lr 0.003 --seed 1 --normalize
run f_lr3e-3_b32_e80_d3 --epochs 80  --batch 32 --lr 0.003 --seed 1 --dropout 0.3
run f_lr2e-3_b32_e150_wd --epochs 150 --batch 32 --lr 0.002 --seed 1 --weight-decay 0.01
# the unfiltered windows with the first recipe, for the record (what round 2 shipped was trained on these)
name=raw_lr3e-3_b32_e120; rm -f gpurun_out/real/$name.jsonl
python -m nsd_amd.train --data tests/golden/recorded_trials.npz --classes 3 --kfold 5 --out gpurun_out/real/$name.pth --log-jsonl gpurun_out/real/$name.jsonl \
    --log-every 20 --epochs 120 --batch 32 --lr 0.003 --seed 1 > gpurun_out/real/$name.out 2>&1
tail -1 gpurun_out/real/$name.jsonl | cut -c1-220
