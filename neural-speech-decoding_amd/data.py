"""Trial loader for the reference's recorded data set (EEG_data_collection/*.csv).

On-disk format (written by the reference's Neural_decoding_data_collector.py:129-139): one file per trial,
no header, comma-separated `%.7f` floats, 625 rows (5 s @ 125 Hz) x 8 channels, already band-limited and
zero-mean.  The class is the file-name prefix before the first '_'
(backgroundnoise / food / no / water / yes) -- the prefix is the only label carrier.

The whole data set is 324 x 20 KB = 6.5 MB: it is parsed once, kept as one [N,625,8] fp32 array (optionally
cached as .npz next to the CSVs) and moved to HBM in full; there is no per-step I/O.
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PREFIXES = ("backgroundnoise", "food", "no", "water", "yes")

# Explicit label maps (the reference's own naming is inconsistent, SURVEY 6 / BASELINE.md 2):
#  * LABELS_3CLASS_CHECKPOINT: the order the shipped checkpoint was evidently trained with (its file name
#    says Water_Food_Bg_Noise; with this order the reference reproduces its ~70 % claim)
#  * LABELS_3CLASS_CODE: the order of CLASS_NAMES in the reference code (Food, Water, BG-Noise)
LABELS_3CLASS_CHECKPOINT: Dict[str, int] = {"water": 0, "food": 1, "backgroundnoise": 2}
LABELS_3CLASS_CODE: Dict[str, int] = {"food": 0, "water": 1, "backgroundnoise": 2}
LABELS_5CLASS: Dict[str, int] = {"yes": 0, "no": 1, "food": 2, "water": 3, "backgroundnoise": 4}


@dataclass
class TrialSet:
    x: np.ndarray            # [N, T, C] float32
    y: np.ndarray            # [N] int32 class index under `label_map`
    prefix: List[str]        # file-name prefix of each trial
    files: List[str]
    label_map: Dict[str, int]

    def __len__(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_classes(self) -> int:
        return len(set(self.label_map.values()))


def parse_trial_csv(path: str) -> np.ndarray:
    """[T, C] float32 from one trial file."""
    a = np.loadtxt(path, delimiter=",", dtype=np.float64, ndmin=2)
    return a.astype(np.float32)


def prefix_of(path: str) -> str:
    return os.path.basename(path).split("_", 1)[0].lower()


def load_trials(directory: str, label_map: Optional[Dict[str, int]] = None, *, samples: int = 625, channels: int = 8,
                cache: bool = False) -> TrialSet:
    """Parse every `<prefix>_*.csv` under `directory` whose prefix is in `label_map` (default: the 3-class map of
    the shipped checkpoint).  Files with another shape than [samples, channels] are rejected loudly."""
    label_map = dict(LABELS_3CLASS_CHECKPOINT if label_map is None else label_map)
    files = sorted(f for f in glob.glob(os.path.join(directory, "*.csv")) if prefix_of(f) in label_map)
    if not files:
        raise FileNotFoundError(f"no trial CSVs with prefixes {sorted(label_map)} under {directory!r}")
    cache_path = os.path.join(directory, f".nsd_cache_{samples}x{channels}_{len(files)}.npz")
    x = None
    if cache and os.path.exists(cache_path):
        z = np.load(cache_path, allow_pickle=False)
        if list(z["files"]) == [os.path.basename(f) for f in files]:
            x = z["x"]
    if x is None:
        x = np.empty((len(files), samples, channels), np.float32)
        for i, f in enumerate(files):
            a = parse_trial_csv(f)
            if a.shape != (samples, channels):
                raise ValueError(f"{f}: expected [{samples},{channels}] got {a.shape}")
            x[i] = a
        if cache:
            try:
                np.savez(cache_path, x=x, files=np.array([os.path.basename(f) for f in files]))
            except OSError:
                pass                      # read-only data directory: just skip the cache
    prefix = [prefix_of(f) for f in files]
    y = np.array([label_map[p] for p in prefix], np.int32)
    return TrialSet(x=x, y=y, prefix=prefix, files=files, label_map=label_map)


def load_trials_npz(path: str, label_map: Optional[Dict[str, int]] = None, x_key: str = "x") -> TrialSet:
    """The same TrialSet from a packed copy of the data set (`x` [N,T,C] float32, `stem` [N] and optionally `prefix` [N]; written
    by tests/golden/make_trials_fixture.py with load_trials itself) -- the form in which the recorded trials travel to
    machines that do not hold the CSV directory.  `x_key`: the array holding the windows (`x_filt` in
    tests/golden/recorded_trials_filtered.npz: the same trials as the reference's PreProcessor hands them to the model)."""
    label_map = dict(LABELS_3CLASS_CHECKPOINT if label_map is None else label_map)
    z = np.load(path, allow_pickle=False)
    if x_key not in z.files:
        raise KeyError(f"{path!r} has no array {x_key!r} (arrays: {z.files})")
    prefix_all = [str(p) for p in z["prefix"]] if "prefix" in z.files else [prefix_of(str(s)) for s in z["stem"]]
    keep = np.array([p in label_map for p in prefix_all])
    if not keep.any():
        raise FileNotFoundError(f"no trials with prefixes {sorted(label_map)} in {path!r}")
    prefix = [p for p, k in zip(prefix_all, keep) if k]
    files = [str(s) + ".csv" for s, k in zip(z["stem"], keep) if k]
    x = np.ascontiguousarray(z[x_key][keep], dtype=np.float32)
    y = np.array([label_map[p] for p in prefix], np.int32)
    return TrialSet(x=x, y=y, prefix=prefix, files=files, label_map=label_map)


def stratified_folds(y: Sequence[int], k: int, seed: int = 0) -> List[np.ndarray]:
    """k disjoint validation index sets covering every trial once, each with (nearly) the class proportions of the whole set."""
    y = np.asarray(y)
    rs = np.random.RandomState(seed)
    folds: List[List[int]] = [[] for _ in range(k)]
    for cls in np.unique(y):
        idx = np.flatnonzero(y == cls)
        rs.shuffle(idx)
        for i, v in enumerate(idx):
            folds[i % k].append(int(v))
    return [np.sort(np.array(f, dtype=np.int64)) for f in folds]


def stratified_split(y: Sequence[int], val_fraction: float = 0.2, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Indices (train, val): every class contributes round(val_fraction * count) trials to the validation set."""
    y = np.asarray(y)
    rs = np.random.RandomState(seed)
    tr, va = [], []
    for cls in np.unique(y):
        idx = np.flatnonzero(y == cls)
        rs.shuffle(idx)
        k = int(round(val_fraction * idx.size))
        va.append(idx[:k]); tr.append(idx[k:])
    return np.sort(np.concatenate(tr)), np.sort(np.concatenate(va))


def epoch_batches(n: int, batch: int, seed: int, epoch: int, drop_last: bool = False):
    """Shuffled index batches of one epoch (same permutation on every rank: seed + epoch)."""
    perm = np.random.RandomState(seed + 1000003 * epoch).permutation(n)
    for lo in range(0, n, batch):
        idx = perm[lo:lo + batch]
        if drop_last and idx.size < batch:
            break
        yield idx
