#!/usr/bin/env python3
"""Write tests/golden/recorded_trials_filtered.npz: the reference's SimplePredictor.predict, WITH its preprocessing, as DATA.

    python tests/golden/make_predict_fixture.py --reference /root/reference          (build container only, ~7 min of CPU)

For every recorded window of tests/golden/recorded_trials.npz (the 324 trials of EEG_data_collection/, same order) the
reference's own `SimplePredictor` (Neuro-Alpha-App/Utilities/lstm_eeg_model.py:42-101, imported where it lies, reference
checkpoint, default tailoring_lambda) is run exactly as tester.py:83-88 runs it.  Stored, outputs only:

  x_filt     [324,625,8] fp32   what `PreProcessor.transform` (preprocessor.py:21-36: the third-party MindsAI filter) handed to the
                                model for that window -- the filter itself is out of scope and is never restated
  ref_probs  [324,3]     fp32   the probabilities predict() returned (lstm_eeg_model.py:95-98)
  ref_label  [324]       str    the label predict() returned (CLASS_NAMES[argmax], lstm_eeg_model.py:99-101)
  stem       [324]       str    file stem of the trial (same order as recorded_trials.npz)

Uses: the GPU parity test of `SimplePredictor.predict` with the reference's preprocessing (a replay preprocessor returns x_filt
for the window it is given), and the training set of the shipped checkpoint (trained on what the predictor feeds the model).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "recorded_trials_filtered.npz"))
    args = ap.parse_args()
    sys.dont_write_bytecode = True                 # the reference tree is read-only
    sys.path.insert(0, os.path.join(args.reference, "Neuro-Alpha-App", "Utilities"))
    import torch
    import lstm_eeg_model as ref                   # the reference module, imported where it lies
    torch.set_num_threads(1)
    rec = np.load(os.path.join(HERE, "recorded_trials.npz"))
    x = rec["x"]
    pred = ref.SimplePredictor(os.path.join(args.reference, "DeepLearning", "LSTM_Model", "lstm_classifier_Water_Food_Bg_Noise.pth"), sr=125)
    xf = np.zeros_like(x)
    probs = np.zeros((len(x), 3), np.float32)
    labels = []
    for i in range(len(x)):
        xf[i] = pred.pre.transform(x[i])
        p, lab = pred.predict(x[i])
        probs[i] = p
        labels.append(lab)
        if i % 20 == 0:
            print(i, rec["stem"][i], p, lab, flush=True)
    # predict() == model(filtered window): the stored filtered windows really are what the model saw
    with torch.no_grad():
        chk = torch.softmax(pred.model(torch.from_numpy(xf[:8])), -1).numpy()
    assert np.abs(chk - probs[:8]).max() < 1e-5, np.abs(chk - probs[:8]).max()
    np.savez_compressed(args.out, x_filt=xf, ref_probs=probs, ref_label=np.array(labels), stem=rec["stem"])
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
