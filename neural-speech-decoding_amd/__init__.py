"""MI355X-native EEG imagined-speech LSTM path (drop-in for the reference's Utilities package surface).

Public names mirror the reference: EEG_LSTM, SimplePredictor, CLASS_NAMES (Utilities/lstm_eeg_model.py),
run_trials, TrialResult (Utilities/tester.py), StreamingProcess (Utilities/streaming_process.py).
"""
from ._lib import NsdError, build as build_library, lib as load_library
from .lstm_eeg_model import CLASS_NAMES, EEG_LSTM, IdentityPreProcessor, SimplePredictor, resolve_reference_preprocessor
from .ops import ModelSpec
from .streaming_process import StreamingProcess
from .tester import DEFAULT_MODEL, DEFAULT_SERIAL, TrialResult, run_trials

__all__ = ["EEG_LSTM", "SimplePredictor", "CLASS_NAMES", "run_trials", "TrialResult", "StreamingProcess",
           "ModelSpec", "NsdError", "IdentityPreProcessor", "resolve_reference_preprocessor", "build_library", "load_library", "DEFAULT_MODEL", "DEFAULT_SERIAL"]
