"""Command-line trainer standing in for the reference's missing notebook (DeepLearning/lstm_trainer.ipynb,
.MISSING_LARGE_BLOBS:1).  Trains EEG_LSTM on recorded trials (or synthetic windows) on 1..8 MI355X and writes a
checkpoint the reference's SimplePredictor loads unchanged (lstm_eeg_model.py:77-81).

    python -m nsd_amd.train --data /path/to/EEG_data_collection --classes 3 --epochs 60 --out model.pth
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m nsd_amd.train --synthetic 8192 ...

Every rank holds the whole data set in HBM (6.5 MB); each global batch is split contiguously over the ranks
(shard_range) and the flat gradient is summed with one RCCL all-reduce per step (trainer.py).
"""
from __future__ import annotations

import argparse
import json
import sys
import time

import numpy as np
import torch

from . import data as D
from .lstm_eeg_model import EEG_LSTM
from .trainer import Trainer, init_distributed, save_reference_checkpoint, shard_range


def evaluate(model: EEG_LSTM, x: torch.Tensor, y: torch.Tensor, batch: int = 512) -> float:
    was_training = model.training
    model.eval()
    correct = 0
    with torch.no_grad():
        for lo in range(0, x.shape[0], batch):
            pred = model(x[lo:lo + batch]).argmax(-1)
            correct += int((pred == y[lo:lo + batch].long()).sum().item())
    model.train(was_training)
    return correct / max(int(x.shape[0]), 1)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", help="directory with <prefix>_*.csv trials (reference: EEG_data_collection/), or a packed .npz "
                                   "of them (data.load_trials_npz)")
    ap.add_argument("--log-jsonl", default=None, help="also append the per-epoch JSON lines to this file (rank 0)")
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic windows x = 2.7*N(0,1) instead of --data")
    ap.add_argument("--classes", type=int, default=3, choices=(3, 5))
    ap.add_argument("--label-order", default="checkpoint", choices=("checkpoint", "code"))
    ap.add_argument("--T", type=int, default=625)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch size")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--weight-decay", type=float, default=0.0)
    ap.add_argument("--hidden", type=int, default=48)
    ap.add_argument("--dropout", type=float, default=0.60)
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"),
                    help="bf16: the sequence-batched path (hidden 64/128/256/512; BASELINE cfg3 = --hidden 256 --classes 5 --precision bf16)")
    ap.add_argument("--bidirectional", action="store_true", help="bidirectional LSTM (needs --precision bf16)")
    ap.add_argument("--val-fraction", type=float, default=0.2)
    ap.add_argument("--normalize", action="store_true", help="per-channel z-score of each window (app.py:166-170)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="eeg_lstm.pth")
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--npz-key", default="x", help="array of a packed --data .npz holding the windows (`x_filt`: the windows as the "
                                                   "reference's PreProcessor hands them to the model, tests/golden/recorded_trials_filtered.npz)")
    ap.add_argument("--kfold", type=int, default=0, help="K > 1: K-fold cross-validation (mean +- sd of the LAST-epoch validation accuracy, "
                                                         "no epoch selection), then --out is trained on ALL trials with the same recipe")
    args = ap.parse_args(argv)

    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        print("train: needs an MI355X (no CPU training path)", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.synthetic:
        rs = np.random.RandomState(args.seed)
        y_np = rs.randint(0, args.classes, args.synthetic).astype(np.int32)
        # class-dependent mean shift on one channel so that there is something to learn
        x_np = (2.7 * rs.standard_normal((args.synthetic, args.T, 8))).astype(np.float32)
        x_np[np.arange(args.synthetic), :, y_np % 8] += 1.5
        tr_idx, va_idx = D.stratified_split(y_np, args.val_fraction, args.seed)
    else:
        if not args.data:
            ap.error("give --data or --synthetic")
        lm = D.LABELS_5CLASS if args.classes == 5 else (D.LABELS_3CLASS_CHECKPOINT if args.label_order == "checkpoint" else D.LABELS_3CLASS_CODE)
        ts = D.load_trials_npz(args.data, lm, x_key=args.npz_key) if args.data.endswith(".npz") else D.load_trials(args.data, lm, samples=args.T)
        if ts.x.shape[1] != args.T:
            ap.error(f"--T {args.T} but the trials have {ts.x.shape[1]} samples")
        x_np, y_np = ts.x, ts.y
        tr_idx, va_idx = D.stratified_split(y_np, args.val_fraction, args.seed)
    x_all = torch.from_numpy(x_np).to(dev)
    y_all = torch.from_numpy(y_np).to(dev)

    def emit(rec: dict) -> None:
        if rank != 0:
            return
        line = json.dumps(rec)
        print(line, flush=True)
        if args.log_jsonl:
            with open(args.log_jsonl, "a") as f:
                f.write(line + "\n")

    def fit(tr_idx, va_idx, seed: int, tag: str, out_path, keep_best: bool):
        """Train one model on tr_idx for args.epochs epochs.  keep_best: write the checkpoint of the best validation epoch
        (model selection ON the validation set: its accuracy is then optimistic); otherwise the model after the LAST epoch is
        what is written / scored -- the number of epochs is fixed beforehand, nothing is selected."""
        torch.manual_seed(seed)           # (Trainer also broadcasts rank 0's parameters when world > 1)
        model = EEG_LSTM(8, args.hidden, 2, args.classes, args.dropout, normalize=args.normalize, precision=args.precision,
                         bidirectional=args.bidirectional).to(dev).train()
        trainer = Trainer(model, lr=args.lr, weight_decay=args.weight_decay, seed=seed + 1)
        tr_dev = torch.from_numpy(tr_idx).to(dev)
        best = (-1.0, -1)
        t0 = time.time()
        acc_tr = acc_va = float("nan")
        for epoch in range(args.epochs):
            for idx in D.epoch_batches(len(tr_idx), args.batch, seed, epoch, drop_last=len(tr_idx) >= args.batch):
                # every rank takes part in every step (the step ends in a collective): a rank whose shard of a short tail
                # batch is empty passes zero trials and contributes a zero gradient; the mean is over the GLOBAL batch
                lo, hi = shard_range(len(idx), rank, world)
                sel = tr_dev[torch.from_numpy(idx[lo:hi]).to(dev)]
                trainer.step(x_all[sel].contiguous(), y_all[sel].contiguous(), global_batch=len(idx))
            last = epoch == args.epochs - 1
            if epoch % args.log_every == 0 or last:
                trainer.check()                  # every rank: raises (non-zero exit) if a scan group timed out anywhere
            if rank == 0 and (epoch % args.log_every == 0 or last):
                acc_tr = evaluate(model, x_all[tr_dev], y_all[tr_dev])
                acc_va = evaluate(model, x_all[va_idx], y_all[va_idx]) if len(va_idx) else float("nan")
                if keep_best and out_path and acc_va > best[0]:
                    best = (acc_va, epoch)
                    save_reference_checkpoint(model, out_path)
                emit({"run": tag, "epoch": epoch, "loss_last_batch": round(trainer.last_loss(), 5), "acc_train": round(acc_tr, 4),
                      "acc_val": round(acc_va, 4), "elapsed_s": round(time.time() - t0, 2)})
        if rank == 0 and out_path and (not keep_best or best[1] < 0):
            save_reference_checkpoint(model, out_path)
        return {"acc_train_last": acc_tr, "acc_val_last": acc_va, "best_val_acc": best[0], "best_epoch": best[1]}

    if args.kfold > 1:
        # accuracy estimate: k stratified folds, every model trained for the SAME, pre-set number of epochs and scored after its
        # last epoch (no epoch picked on the validation data); then the shipped model: all trials, same recipe, no held-out set
        folds = D.stratified_folds(y_np, args.kfold, args.seed)
        accs = []
        for f, va in enumerate(folds):
            tr = np.setdiff1d(np.arange(len(y_np)), va)
            r = fit(tr, va, args.seed + 101 * f, f"fold{f}", None, keep_best=False)
            accs.append(r["acc_val_last"])
            emit({"fold": f, "n_train": int(len(tr)), "n_val": int(len(va)), "acc_val_last_epoch": round(r["acc_val_last"], 4),
                  "acc_train_last_epoch": round(r["acc_train_last"], 4)})
        everything = np.arange(len(y_np))
        r = fit(everything, everything[:0], args.seed + 7777, "all", args.out, keep_best=False)
        emit({"done": True, "kfold": args.kfold, "acc_val_mean": round(float(np.mean(accs)), 4) if rank == 0 else None,
              "acc_val_sd": round(float(np.std(accs, ddof=1)), 4) if rank == 0 else None,
              "acc_val_folds": [round(float(a), 4) for a in accs], "selection": "none: fixed epoch count, last-epoch model",
              "shipped": {"checkpoint": args.out, "trained_on": int(len(everything)), "acc_train_last_epoch": r["acc_train_last"]},
              "world": world, "args": {k: v for k, v in vars(args).items()}})
        return 0

    r = fit(tr_idx, va_idx, args.seed, "split", args.out, keep_best=True)
    emit({"done": True, "best_val_acc": r["best_val_acc"], "best_epoch": r["best_epoch"], "acc_val_last_epoch": r["acc_val_last"],
          "checkpoint": args.out, "world": world,
          "n_train": int(len(tr_idx)), "n_val": int(len(va_idx)), "args": {k: v for k, v in vars(args).items()}})
    return 0


if __name__ == "__main__":
    sys.exit(main())
