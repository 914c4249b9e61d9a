"""Counterpart of the reference harness Neuro-Alpha-App/Utilities/tester.py: collect N windows from a
producer process, classify each on the MI355X, average probabilities and windows.

`run_trials` / `TrialResult` keep the reference's signature, defaults, printed lines and exception
surface (tester.py:23-110), so Frontend/app.py:141-160 consumes the result unchanged.
"""
from dataclasses import dataclass
from multiprocessing import Queue, freeze_support
from pathlib import Path
import time
from typing import Optional

import numpy as np

from .streaming_process import StreamingProcess
from .lstm_eeg_model import SimplePredictor

DEFAULT_SERIAL = "/dev/cu.usbserial-FTB6SPL3"          # tester.py:17 (a real board; see streaming_process.py)
DEFAULT_MODEL = str(Path(__file__).resolve().parent / "LSTM_Model" / "lstm_classifier_Water_Food_Bg_Noise.pth")


@dataclass
class TrialResult:
    trials: int
    avg_probs: Optional[np.ndarray]
    avg_chunk: Optional[np.ndarray] = None


def run_trials(trials: int = 10, serial_port: str = DEFAULT_SERIAL, num_channels: int = 8, window_seconds: float = 5.0,
               model_path: str = DEFAULT_MODEL, verbose: bool = True, *, producer_factory=None,
               queue_timeout: float = 6.5, predictor_kwargs: Optional[dict] = None) -> TrialResult:
    """Collect `trials` windows, run SimplePredictor on each, return the averages.

    The positional/keyword interface is the reference's.  Keyword-only additions: `producer_factory`
    (callable(serial_port, num_channels, window_seconds, out_queue) -> process; default StreamingProcess),
    `queue_timeout` (the reference hard-codes 6.5 s), `predictor_kwargs` (extra SimplePredictor kwargs).
    """
    q = Queue(maxsize=8)
    make = producer_factory or StreamingProcess
    producer = make(serial_port=serial_port, num_channels=num_channels, window_seconds=window_seconds, out_queue=q)
    producer.start()                      # before any HIP initialisation in this process (fork safety)
    producer.recording_flag.value = True

    predictor = None
    collected = 0
    sum_probs = np.zeros(3, dtype=np.float32)      # 3-class accumulator, as tester.py:54
    sum_chunk = None
    try:
        while collected < trials:
            if not producer.is_alive():
                raise RuntimeError("Producer exited unexpectedly")
            try:
                item = q.get(timeout=queue_timeout)
            except Exception:
                if verbose:
                    print("Waiting for chunk...", flush=True)
                continue
            chunk = np.asarray(item["data"])
            sr = item["sr"]
            channels = item.get("channels")
            if predictor is None:
                predictor = SimplePredictor(pth_path=model_path, sr=sr, channel_order=channels, input_size=num_channels,
                                            hidden_size=48, num_layers=2, num_classes=3, dropout=0.60, device="cpu",
                                            tailoring_lambda=1.25e-29, class_names=["Food", "Water", "None"],
                                            **(predictor_kwargs or {}))
            probs, label = predictor.predict(chunk)
            sum_probs += probs
            sum_chunk = chunk if sum_chunk is None else sum_chunk + chunk
            collected += 1
            if verbose:
                stamp = time.strftime("%H:%M:%S")
                print(f"[Trial {collected:02d} @ {stamp}] pred={label} probs={np.round(probs, 3)}")
        avg_probs = (sum_probs / collected) if collected else None
        avg_chunk = (sum_chunk / collected) if (collected and sum_chunk is not None) else None
        if verbose:
            if avg_probs is not None:
                print(f"\nAveraged over {collected} trials: {np.round(avg_probs, 3)}")
                if avg_chunk is not None:
                    print(f"Averaged chunk shape: {avg_chunk.shape}")
            else:
                print("No trials completed; no average available.")
        return TrialResult(trials=collected, avg_probs=avg_probs, avg_chunk=avg_chunk)
    finally:
        producer.recording_flag.value = False
        producer.stop()
        producer.join(timeout=5.0)
        if producer.is_alive():
            producer.terminate()


def main():
    run_trials()


if __name__ == "__main__":
    freeze_support()
    main()
