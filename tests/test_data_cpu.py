"""Trial loader (EEG_data_collection/*.csv format) -- CPU only."""
import os

import numpy as np
import pytest

from nsd_amd import data as D

REF_DATA = "/root/reference/EEG_data_collection"


def _write(d, name, a):
    np.savetxt(os.path.join(d, name), a, fmt="%.7f", delimiter=",")


def test_load_trials_labels_shapes_and_cache(tmp_path):
    rs = np.random.RandomState(0)
    arrs = {}
    for i, p in enumerate(["food", "water", "backgroundnoise", "yes", "food"]):
        a = rs.standard_normal((625, 8)).astype(np.float32)
        arrs[f"{p}_{i:04d}-uuid.csv"] = a
        _write(tmp_path, f"{p}_{i:04d}-uuid.csv", a)
    ts = D.load_trials(str(tmp_path), cache=True)                      # default 3-class map: 'yes' is ignored
    assert len(ts) == 4 and ts.x.shape == (4, 625, 8) and ts.x.dtype == np.float32
    assert ts.num_classes == 3 and sorted(ts.prefix) == ["backgroundnoise", "food", "food", "water"]
    for f, lab, x in zip(ts.files, ts.y, ts.x):
        name = os.path.basename(f)
        assert lab == D.LABELS_3CLASS_CHECKPOINT[D.prefix_of(name)]
        assert np.abs(x - arrs[name]).max() < 1e-6                        # %.7f round trip
    again = D.load_trials(str(tmp_path), cache=True)                     # served from the .npz cache
    assert np.array_equal(again.x, ts.x) and np.array_equal(again.y, ts.y)
    five = D.load_trials(str(tmp_path), D.LABELS_5CLASS)
    assert len(five) == 5 and five.num_classes == 5
    assert D.LABELS_3CLASS_CODE["food"] == 0 and D.LABELS_3CLASS_CHECKPOINT["water"] == 0


def test_bad_inputs(tmp_path):
    with pytest.raises(FileNotFoundError):
        D.load_trials(str(tmp_path))
    _write(tmp_path, "food_x.csv", np.zeros((600, 8)))
    with pytest.raises(ValueError, match="expected"):
        D.load_trials(str(tmp_path))


def test_stratified_split_and_batches():
    y = np.array([0] * 40 + [1] * 69 + [2] * 70)
    tr, va = D.stratified_split(y, 0.2, seed=1)
    assert len(set(tr) & set(va)) == 0 and len(tr) + len(va) == len(y)
    assert [int((y[va] == c).sum()) for c in range(3)] == [8, 14, 14]
    b1 = [i.tolist() for i in D.epoch_batches(10, 4, seed=3, epoch=0)]
    b2 = [i.tolist() for i in D.epoch_batches(10, 4, seed=3, epoch=0)]
    assert b1 == b2 and sorted(sum(b1, [])) == list(range(10)) and [len(b) for b in b1] == [4, 4, 2]
    assert [len(b) for b in D.epoch_batches(10, 4, 3, 1, drop_last=True)] == [4, 4]


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data set not present (GPU box)")
def test_reference_dataset_counts():
    ts = D.load_trials(REF_DATA, D.LABELS_5CLASS)
    counts = {p: ts.prefix.count(p) for p in D.PREFIXES}
    assert counts == {"backgroundnoise": 40, "food": 69, "no": 71, "water": 70, "yes": 74}     # SURVEY 2 (#9)
    assert ts.x.shape == (324, 625, 8) and np.isfinite(ts.x).all()
    assert abs(float(ts.x.std()) - 2.73) < 0.05


FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "recorded_trials.npz")


def test_packed_fixture_is_the_recorded_data_set():
    """tests/golden/recorded_trials.npz (the form in which the trials reach the GPU box) == what load_trials parses
    from the reference's CSV directory, for the 3-class and the 5-class label maps."""
    five = D.load_trials_npz(FIXTURE, D.LABELS_5CLASS)
    assert five.x.shape == (324, 625, 8) and five.x.dtype == np.float32 and five.num_classes == 5
    assert {p: five.prefix.count(p) for p in D.PREFIXES} == {"backgroundnoise": 40, "food": 69, "no": 71, "water": 70, "yes": 74}
    three = D.load_trials_npz(FIXTURE)
    assert len(three) == 179 and sorted(set(three.prefix)) == ["backgroundnoise", "food", "water"]
    assert [int((three.y == c).sum()) for c in range(3)] == [70, 69, 40]               # water, food, backgroundnoise
    with pytest.raises(FileNotFoundError):
        D.load_trials_npz(FIXTURE, {"nothing": 0})
    if os.path.isdir(REF_DATA):
        for lm, packed in ((None, three), (D.LABELS_5CLASS, five)):
            ts = D.load_trials(REF_DATA, lm)
            assert np.array_equal(ts.x, packed.x) and np.array_equal(ts.y, packed.y)
            assert [os.path.basename(f) for f in ts.files] == packed.files
