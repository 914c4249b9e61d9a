"""Live harness with the contract of the reference's Neuro-Alpha-App/Utilities/tester.py:23-110 — pull N windows
from a producer process, classify each on the MI355X, hand back the averaged probabilities and window.

What is kept from the reference is the *contract*: `run_trials` signature and defaults (tester.py:30-37), the
`TrialResult` fields (:23-27), the two module constants (:17-20), the lines printed in verbose mode (:66,:95,:101-105),
`RuntimeError("Producer exited unexpectedly")` (:60), the 3-slot probability accumulator (:54), the label list
handed to the predictor (:85) and the predictor being built only after the producer is running (:49 before :73 —
here that also keeps HIP un-initialised across the fork).  The body is organised differently: a window source that
owns the producer's lifetime, a running-mean accumulator, and a reporting helper.
"""
from __future__ import annotations

import os
import queue as _queue
import time
from contextlib import contextmanager
from dataclasses import dataclass
from multiprocessing import Queue, freeze_support
from pathlib import Path
from typing import Iterator, Optional, Tuple

import numpy as np

from .streaming_process import StreamingProcess
from .lstm_eeg_model import SimplePredictor

DEFAULT_SERIAL = "/dev/cu.usbserial-FTB6SPL3"          # tester.py:17 (a real board; see streaming_process.py)
# tester.py:18-20: <this package>/LSTM_Model/<checkpoint>.  The file shipped there is a checkpoint written by THIS
# repository's trainer on the reference's recorded trials (profiles/r02_real_data_train.jsonl); NSD_MODEL_PATH overrides.
DEFAULT_MODEL = str(Path(__file__).resolve().parent / "LSTM_Model" / "lstm_classifier_Water_Food_Bg_Noise.pth")

_HARNESS_LABELS = ("Food", "Water", "None")             # tester.py:85
_QUEUE_DEPTH = 8                                        # tester.py:42
_JOIN_SECONDS = 5.0                                     # tester.py:110


@dataclass
class TrialResult:
    trials: int
    avg_probs: Optional[np.ndarray]
    avg_chunk: Optional[np.ndarray] = None


def resolve_model_path(model_path: str) -> str:
    """`model_path` as given, except that the built-in default may be redirected with NSD_MODEL_PATH; a missing file is
    reported before any GPU work starts."""
    path = model_path
    if model_path == DEFAULT_MODEL and os.environ.get("NSD_MODEL_PATH"):
        path = os.environ["NSD_MODEL_PATH"]
    if not os.path.isfile(path):
        raise FileNotFoundError(f"run_trials: checkpoint {path!r} does not exist (pass model_path=..., or set "
                                "NSD_MODEL_PATH; `python -m nsd_amd.train --out <path>` writes one)")
    return path


class _RunningMean:
    """Sum of probability vectors and of windows, divided on demand."""

    def __init__(self, classes: int):
        self.n = 0
        self._p = np.zeros(classes, dtype=np.float32)
        self._w: Optional[np.ndarray] = None

    def add(self, probs: np.ndarray, window: np.ndarray) -> None:
        self._p += probs
        self._w = window if self._w is None else self._w + window
        self.n += 1

    def result(self) -> TrialResult:
        if self.n == 0:
            return TrialResult(trials=0, avg_probs=None, avg_chunk=None)
        return TrialResult(trials=self.n, avg_probs=self._p / self.n,
                           avg_chunk=None if self._w is None else self._w / self.n)


@contextmanager
def _recording(producer) -> Iterator[None]:
    """Producer running and recording inside the block; flag cleared, stopped and joined on the way out, whatever
    happened inside (tester.py:49-50,107-110)."""
    producer.start()
    producer.recording_flag.value = True
    try:
        yield
    finally:
        producer.recording_flag.value = False
        producer.stop()
        producer.join(timeout=_JOIN_SECONDS)
        if producer.is_alive():            # a stuck child must not outlive the call
            producer.terminate()


def _windows(producer, q: Queue, timeout: float, verbose: bool) -> Iterator[Tuple[np.ndarray, int, Optional[list]]]:
    """Endless stream of (window [T,C], sampling rate, channel list) taken off the queue.  A dead producer is an error
    (tester.py:59-60); an empty queue is only reported (tester.py:62-67)."""
    while True:
        if not producer.is_alive():
            raise RuntimeError("Producer exited unexpectedly")
        try:
            payload = q.get(timeout=timeout)
        except _queue.Empty:
            if verbose:
                print("Waiting for chunk...", flush=True)
            continue
        yield np.asarray(payload["data"]), payload["sr"], payload.get("channels")


def _report(res: TrialResult) -> None:
    if res.avg_probs is None:
        print("No trials completed; no average available.")
        return
    print(f"\nAveraged over {res.trials} trials: {np.round(res.avg_probs, 3)}")
    if res.avg_chunk is not None:
        print(f"Averaged chunk shape: {res.avg_chunk.shape}")


def run_trials(trials: int = 10, serial_port: str = DEFAULT_SERIAL, num_channels: int = 8, window_seconds: float = 5.0,
               model_path: str = DEFAULT_MODEL, verbose: bool = True, *, producer_factory=None,
               queue_timeout: float = 6.5, predictor_kwargs: Optional[dict] = None) -> TrialResult:
    """Collect `trials` windows, run SimplePredictor on each, return the averages.

    Positional / keyword interface: the reference's.  Keyword-only additions: `producer_factory`
    (callable(serial_port=, num_channels=, window_seconds=, out_queue=) -> process; default StreamingProcess),
    `queue_timeout` (the reference hard-codes 6.5 s), `predictor_kwargs` (extra SimplePredictor kwargs, e.g.
    {"preprocess": "identity"} when the reference's MindsAI filter is not importable).
    """
    q = Queue(maxsize=_QUEUE_DEPTH)
    producer = (producer_factory or StreamingProcess)(serial_port=serial_port, num_channels=num_channels,
                                                      window_seconds=window_seconds, out_queue=q)
    mean = _RunningMean(classes=len(_HARNESS_LABELS))
    predictor = None
    with _recording(producer):
        source = _windows(producer, q, queue_timeout, verbose)
        while mean.n < trials:
            window, sr, channels = next(source)
            if predictor is None:          # built lazily: the sampling rate and channel list come with the first window
                predictor = SimplePredictor(pth_path=resolve_model_path(model_path), sr=sr, channel_order=channels,
                                            input_size=num_channels, hidden_size=48, num_layers=2, num_classes=3,
                                            dropout=0.60, device="cpu", tailoring_lambda=1.25e-29,
                                            class_names=list(_HARNESS_LABELS), **(predictor_kwargs or {}))
            probs, label = predictor.predict(window)
            mean.add(probs, window)
            if verbose:
                print(f"[Trial {mean.n:02d} @ {time.strftime('%H:%M:%S')}] pred={label} probs={np.round(probs, 3)}")
        res = mean.result()
        if verbose:
            _report(res)
        return res


def main():
    run_trials()


if __name__ == "__main__":
    freeze_support()
    main()
