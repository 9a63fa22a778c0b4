"""ORACLE (test infrastructure, never shipped): CPU restatement of the log-mel path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates, in vectorised numpy float64, the numpy branch of the reference's feature extractor:
  * reference `src/utilities/feature_extractors.py:14-61` (CustomFeatureExtractor, utterance or
    global normalisation) which subclasses
  * transformers (pinned 4.39.3, requirements.txt:17; 5.15.0 installed here)
    `models/speech_to_text/feature_extraction_speech_to_text.py:104-138` (_extract_fbank_features),
    `:141-163` (utterance_cmvn) and `audio_utils.py` `spectrogram` (:809-1018),
    `mel_filter_bank` (:638-731, kaldi scale, triangularised in mel space), `window_function` (povey).
The torchaudio Kaldi branch the authors ran cannot be imported offline.
Pinned by tests/golden/fbank_*.npz, generated from the imported reference by tests/golden/make_golden.py.
"""
from __future__ import annotations

import numpy as np

FRAME_LENGTH = 400
HOP_LENGTH = 160
FFT_LENGTH = 512
NUM_BINS = 257
PREEMPH = 0.97
MEL_FLOOR = 1.192092955078125e-07


def povey_window() -> np.ndarray:
    # audio_utils.py:786-787 : np.power(np.hanning(400), 0.85), symmetric (periodic=False)
    return np.power(np.hanning(FRAME_LENGTH), 0.85)


def kaldi_mel_filters(num_mel: int = 80, fmin: float = 20.0, fmax: float = 8000.0, sr: int = 16000) -> np.ndarray:
    """(257, num_mel) float64; audio_utils.py:700-731 with mel_scale='kaldi', norm=None."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    mel_freqs = np.linspace(mel(fmin), mel(fmax), num_mel + 2)
    bin_width = sr / ((NUM_BINS - 1) * 2)
    fft_mels = mel(bin_width * np.arange(NUM_BINS))
    diff = np.diff(mel_freqs)
    slopes = mel_freqs[None, :] - fft_mels[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def num_frames(num_samples: int) -> int:
    # audio_utils.py:955 (center=False, "snip edges")
    return int(1 + np.floor((num_samples - FRAME_LENGTH) / HOP_LENGTH))


def fbank(waveform: np.ndarray, num_mel: int = 80) -> np.ndarray:
    """waveform (N,) float32 in [-1,1] -> (T, num_mel) float32 log-mel (no normalisation)."""
    x = waveform.astype(np.float32) * np.float32(2**15)  # feature_extraction_speech_to_text.py:111
    x = x.astype(np.float64)
    T = num_frames(x.size)
    idx = np.arange(FRAME_LENGTH)[None, :] + HOP_LENGTH * np.arange(T)[:, None]
    frames = x[idx]                                                  # (T, 400)
    frames = frames - frames.mean(axis=1, keepdims=True)             # remove_dc_offset, :970-971
    pre = frames.copy()
    pre[:, 1:] -= PREEMPH * frames[:, :-1]                           # :973-975 (RHS uses original samples)
    pre[:, 0] *= 1.0 - PREEMPH
    pre *= povey_window()[None, :]
    spec = np.fft.rfft(pre, n=FFT_LENGTH, axis=1).astype(np.complex64)   # stored as complex64, :958,979
    power = np.abs(spec, dtype=np.float64) ** 2.0                    # :984
    mel = np.maximum(MEL_FLOOR, power @ kaldi_mel_filters(num_mel))  # :989
    return np.log(mel).astype(np.float32)                            # :993, dtype float32


def utterance_cmvn(x: np.ndarray, input_length: int, normalize_means=True, normalize_vars=True,
                   padding_value: float = 0.0) -> np.ndarray:
    """feature_extraction_speech_to_text.py:141-163 — per-bin mean / population-std over valid frames."""
    x = x.astype(np.float32)
    if normalize_means:
        x = x - x[:input_length].mean(axis=0)
    if normalize_vars:
        x = x / x[:input_length].std(axis=0)
    if input_length < x.shape[0]:
        x[input_length:] = padding_value
    return x.astype(np.float32)


def global_normalize(x: np.ndarray, means: np.ndarray, stds: np.ndarray) -> np.ndarray:
    """reference src/utilities/feature_extractors.py:47-49."""
    return (x - means) / stds


def extract(waveform: np.ndarray, norm_type: str = "utterance", means=None, stds=None) -> np.ndarray:
    """Full CustomFeatureExtractor.__call__ for one un-padded clip -> (T, 80) float32."""
    f = fbank(waveform)
    if norm_type == "utterance":
        return utterance_cmvn(f, f.shape[0])
    return global_normalize(f, np.asarray(means), np.asarray(stds)).astype(np.float32)
