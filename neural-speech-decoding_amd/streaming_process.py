"""Window producer with the reference's StreamingProcess protocol (Utilities/streaming_process.py:10-76),
fed from recorded trials or a synthetic generator instead of a BrainFlow serial session.

The reference's producer needs the NeuroPawn board + brainflow (hardware I/O, out of scope here).  This
stand-in keeps the process / queue contract `run_trials` relies on:
  * a multiprocessing.Process with `recording_flag` (Value('b')), `stop()`, constructor
    (serial_port, num_channels, window_seconds, out_queue, start_recording, buffer_size)
  * while recording, emits payload dicts {"sr", "channels", "data" float32[T,C], "t_emit"} with
    put_nowait, dropping the oldest item when the queue is full (streaming_process.py:58-69)
`serial_port` selects the source:
  "replay:<dir-or-glob>"   recorded trial CSVs ([625,8] rows x channels, format of
                           Neural_decoding_data_collector.py:129-139), cycled in sorted order
  "synthetic:" / "synthetic:<seed>"   sinusoid + noise windows (like Frontend/app.py:58-66 mock EEG)
  anything else            treated as a real serial device: the child exits with an error, which
                           run_trials reports exactly like the reference ("Producer exited unexpectedly")
"""
from __future__ import annotations

import glob
import os
import time
from multiprocessing import Event, Process, Queue, Value

import numpy as np

SAMPLING_RATE = 125   # Hz (readme.md:52, Frontend/app.py:38)


def load_trial_csv(path: str) -> np.ndarray:
    """One recorded trial: comma-separated floats, rows = samples, columns = channels -> float32 [T,C]."""
    return np.loadtxt(path, delimiter=",", dtype=np.float64, ndmin=2).astype(np.float32)


def synthetic_window(T: int, C: int, rng: np.random.RandomState, sr: int = SAMPLING_RATE) -> np.ndarray:
    t = np.arange(T, dtype=np.float64) / sr
    freqs = rng.uniform(6.0, 30.0, size=C)
    phase = rng.uniform(0, 2 * np.pi, size=C)
    x = 2.0 * np.sin(2 * np.pi * freqs[None, :] * t[:, None] + phase[None, :]) + 1.8 * rng.standard_normal((T, C))
    return x.astype(np.float32)


class StreamingProcess(Process):
    def __init__(self, serial_port: str, num_channels: int = 8, window_seconds: float = 5.0, out_queue: Queue = None,
                 start_recording: bool = False, buffer_size: int = 450000, *args, realtime: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.serial_port = serial_port
        self.num_channels = int(num_channels)
        self.window_seconds = float(window_seconds)
        self.buffer_size = int(buffer_size)
        self.out_queue = out_queue or Queue(maxsize=8)
        self.recording_flag = Value('b', start_recording)
        self.realtime = bool(realtime)
        self._running = Event()
        self._running.set()

    def _windows(self):
        sr = SAMPLING_RATE
        T = max(1, int(self.window_seconds * sr))
        src = self.serial_port or ""
        if src.startswith("replay:"):
            pat = src[len("replay:"):]
            files = sorted(glob.glob(os.path.join(pat, "*.csv")) if os.path.isdir(pat) else glob.glob(pat))
            if not files:
                raise RuntimeError(f"no trial CSVs match {pat!r}")
            i = 0
            while True:
                w = load_trial_csv(files[i % len(files)])
                i += 1
                if w.shape[0] >= T and w.shape[1] >= self.num_channels:
                    yield sr, w[-T:, :self.num_channels]
        elif src.startswith("synthetic:"):
            seed = int(src[len("synthetic:"):] or 0)
            rng = np.random.RandomState(seed)
            while True:
                yield sr, synthetic_window(T, self.num_channels, rng, sr)
        else:
            raise RuntimeError(f"serial acquisition from {src!r} needs the NeuroPawn board + brainflow, which this "
                               "build does not drive; use 'replay:<dir>' or 'synthetic:'")

    def run(self):
        gen = self._windows()
        channels = list(range(1, self.num_channels + 1))
        last_emit_ts = 0.0
        while self._running.is_set():
            if not self.recording_flag.value:
                time.sleep(0.01)
                continue
            now = time.time()
            if self.realtime and now - last_emit_ts < self.window_seconds:
                time.sleep(0.01)
                continue
            sr, chunk = next(gen)
            payload = {"sr": sr, "channels": channels, "data": np.asarray(chunk, dtype=np.float32), "t_emit": now}
            try:
                self.out_queue.put(payload, timeout=0.05) if not self.realtime else self.out_queue.put_nowait(payload)
                last_emit_ts = now
            except Exception:
                if self.realtime:       # drop-oldest, as streaming_process.py:63-69
                    try:
                        _ = self.out_queue.get_nowait()
                        self.out_queue.put_nowait(payload)
                        last_emit_ts = now
                    except Exception:
                        pass

    def stop(self):
        self._running.clear()
