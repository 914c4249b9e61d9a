#!/bin/bash
# Round-4 recipe pass on the recorded 3-class windows as the reference's PreProcessor hands them to the model
# (tests/golden/recorded_trials_filtered.npz, x_filt): z-score on / off, H in {32, 48}, weight decay, dropout, FIVE seeds per recipe,
# every run a 5-fold cross-validation with the last-epoch model (no epoch selection).  Summary: gpurun_out/real4/summary.jsonl
# (one line per recipe: mean +- sd over the 25 fold accuracies, and the per-seed k-fold means).
mkdir -p gpurun_out/real4
rm -f gpurun_out/real4/summary.jsonl
run() {  # name, extra args
  name=$1; shift
  for seed in 1 2 3 4 5; do
    rm -f gpurun_out/real4/${name}_s$seed.jsonl
    python -m nsd_amd.train --data tests/golden/recorded_trials_filtered.npz --npz-key x_filt --classes 3 --kfold 5 --out gpurun_out/real4/${name}_s$seed.pth \
        --log-jsonl gpurun_out/real4/${name}_s$seed.jsonl --log-every 50 --seed $seed "$@" > gpurun_out/real4/${name}_s$seed.out 2>&1 || echo "FAILED $name seed $seed"
  done
  python - "$name" "$@" <<'PY'
import json, sys, glob, statistics
name = sys.argv[1]
folds, means = [], []
for f in sorted(glob.glob(f"gpurun_out/real4/{name}_s*.jsonl")):
    for ln in open(f):
        j = json.loads(ln)
        if j.get("done"):
            folds += j["acc_val_folds"]; means.append(j["acc_val_mean"])
out = {"recipe": name, "args": sys.argv[2:], "seeds": len(means), "folds": len(folds),
       "acc_val_mean": round(statistics.mean(folds), 4) if folds else None, "acc_val_sd_over_folds": round(statistics.pstdev(folds), 4) if folds else None,
       "kfold_means_per_seed": means, "sd_of_seed_means": round(statistics.pstdev(means), 4) if len(means) > 1 else None}
print(json.dumps(out)); open("gpurun_out/real4/summary.jsonl", "a").write(json.dumps(out) + "\n")
PY
}
run h48_lr1e-3_e200          --epochs 200 --batch 32 --lr 0.001
run h48_lr1e-3_e200_n        --epochs 200 --batch 32 --lr 0.001 --normalize
run h48_lr3e-3_e120_n        --epochs 120 --batch 32 --lr 0.003 --normalize
run h48_lr3e-3_e80_d3        --epochs 80  --batch 32 --lr 0.003 --dropout 0.3
run h48_lr3e-3_e80_d3_n      --epochs 80  --batch 32 --lr 0.003 --dropout 0.3 --normalize
run h48_lr1e-3_e200_wd1e-4   --epochs 200 --batch 32 --lr 0.001 --weight-decay 0.0001
run h48_lr1e-3_e200_wd1e-3_n --epochs 200 --batch 32 --lr 0.001 --weight-decay 0.001 --normalize
run h32_lr1e-3_e200          --epochs 200 --batch 32 --lr 0.001 --hidden 32
run h32_lr3e-3_e120_n        --epochs 120 --batch 32 --lr 0.003 --hidden 32 --normalize
run h32_lr3e-3_e80_d3_n      --epochs 80  --batch 32 --lr 0.003 --hidden 32 --dropout 0.3 --normalize
