#!/usr/bin/env python3
"""Write tests/golden/recorded_trials.npz: the reference's recorded data set as DATA (no reference source).

    python tests/golden/make_trials_fixture.py --reference /root/reference

Every `EEG_data_collection/<prefix>_*.csv` trial ([625,8] `%.7f` floats, format written by the reference's
Neural_decoding_data_collector.py:129-139) parsed by THIS repository's loader (nsd_amd.data.load_trials, so the
fixture also pins the loader), plus the reference model's own logits on every window fed raw (no MindsAI filter), from
the reference class imported where it lies and the reference checkpoint.  The fixture travels to the GPU box, where
/root/reference does not exist: the real-data training test and the 324-window inference parity test read it.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "recorded_trials.npz"))
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    import torch
    import nsd_amd  # noqa: F401  (import shim)
    from nsd_amd import data as D

    ts = D.load_trials(os.path.join(args.reference, "EEG_data_collection"), D.LABELS_5CLASS)
    assert ts.x.shape == (324, 625, 8), ts.x.shape
    sys.path.insert(0, os.path.join(args.reference, "Neuro-Alpha-App", "Utilities"))
    import lstm_eeg_model as ref  # the reference module, imported where it lies
    torch.set_num_threads(1)
    state = torch.load(os.path.join(args.reference, "DeepLearning", "LSTM_Model", "lstm_classifier_Water_Food_Bg_Noise.pth"),
                       map_location="cpu", weights_only=True)
    model = ref.EEG_LSTM(input_size=8, hidden_size=48, num_layers=2, num_classes=3, dropout=0.60)
    model.load_state_dict(state, strict=True)
    model.eval()
    with torch.no_grad():
        logits = torch.cat([model(torch.from_numpy(ts.x[i:i + 1])) for i in range(len(ts))]).numpy()   # B=1, like predict()
    np.savez_compressed(args.out, x=ts.x, prefix=np.array(ts.prefix), stem=np.array([os.path.basename(f)[:-4] for f in ts.files]),
                        ref_logits_raw=logits.astype(np.float32))
    counts = {p: int(sum(q == p for q in ts.prefix)) for p in D.PREFIXES}
    print("wrote", args.out, os.path.getsize(args.out), "bytes", counts)


if __name__ == "__main__":
    main()
