"""Host side of the log-mel front end: float64 tables + the two execution paths.

* `FbankTables` builds the povey window, FFT twiddles and Kaldi mel filterbank exactly as transformers'
  `window_function` / `mel_filter_bank` do (audio_utils.py:638-731, :745-806) for the settings hard-wired in
  `Speech2TextFeatureExtractor.__init__` (feature_extraction_speech_to_text.py:86-102).
* `fbank_gpu` runs the HIP kernels (csrc/fbank.hip) on device waveforms — the GPU pre-stage of `forward`.
* `fbank_numpy` is the fork-safe, GPU-free implementation used by `CustomFeatureExtractor` inside dataloader
  workers (reference src/utilities/callbacks.py:100-118 runs the extractor there); vectorised numpy float64.
"""
from __future__ import annotations

import numpy as np

FRAME, HOP, NFFT, NBINS = 400, 160, 512, 257
MEL_FLOOR = 1.192092955078125e-07
PREEMPH = 0.97


class FbankTables:
    def __init__(self, num_mel: int = 80, sr: int = 16000, fmin: float = 20.0, fmax: float | None = None):
        fmax = float(sr // 2) if fmax is None else fmax
        self.num_mel = num_mel
        self.window = np.power(np.hanning(FRAME), 0.85)                      # "povey", periodic=False
        k = np.arange(NFFT // 2)
        self.twiddle = np.stack([np.cos(2 * np.pi * k / NFFT), -np.sin(2 * np.pi * k / NFFT)], 1)   # exp(-2 pi i k/512)
        mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
        mel_freqs = np.linspace(mel(fmin), mel(fmax), num_mel + 2)
        fft_mels = mel((sr / ((NBINS - 1) * 2)) * np.arange(NBINS))
        diff = np.diff(mel_freqs)
        slopes = mel_freqs[None, :] - fft_mels[:, None]
        self.filters = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))   # (257, nmel)
        nz = self.filters > 0
        self.lo = np.array([int(np.argmax(nz[:, f])) if nz[:, f].any() else 0 for f in range(num_mel)], np.int32)
        self.hi = np.array([int(NBINS - np.argmax(nz[::-1, f])) if nz[:, f].any() else 0 for f in range(num_mel)], np.int32)
        self._dev = {}

    def device(self, device):
        import torch
        key = str(device)
        if key not in self._dev:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev[key] = dict(window=t(self.window), twiddle=t(self.twiddle), mel_t=t(self.filters.T.copy()),
                                  lo=t(self.lo), hi=t(self.hi))
        return self._dev[key]


def num_frames(num_samples: int) -> int:
    return 1 + (num_samples - FRAME) // HOP if num_samples >= FRAME else 0


def fbank_numpy(waveform: np.ndarray, tables: FbankTables) -> np.ndarray:
    """(N,) float32 -> (T, nmel) float32, numerically the numpy branch of the reference extractor."""
    x = (np.asarray(waveform, dtype=np.float32) * np.float32(2 ** 15)).astype(np.float64)
    T = num_frames(x.size)
    idx = np.arange(FRAME)[None, :] + HOP * np.arange(T)[:, None]
    fr = x[idx]
    fr = fr - fr.mean(axis=1, keepdims=True)
    pre = fr.copy()
    pre[:, 1:] -= PREEMPH * fr[:, :-1]
    pre[:, 0] *= 1.0 - PREEMPH
    pre *= tables.window[None, :]
    spec = np.fft.rfft(pre, n=NFFT, axis=1).astype(np.complex64)
    power = np.abs(spec, dtype=np.float64) ** 2.0
    return np.log(np.maximum(MEL_FLOOR, power @ tables.filters)).astype(np.float32)


def fbank_gpu(wave, tables: FbankTables, num_samples=None, pad_frames_to: int | None = None, *, normalize="utterance",
              normalize_means=True, normalize_vars=True, global_means=None, global_stds=None, padding_value=0.0):
    """wave (B,N) float32 device tensor -> (features (B,T,nmel) float32, frames (B,) int32).

    T = frames of the longest clip, optionally rounded up (`pad_frames_to` = collator's pad_to_multiple_of);
    frames beyond a clip's own count hold `padding_value` (the collator's right padding)."""
    import torch

    from . import _lib
    B, N = wave.shape
    tb = tables.device(wave.device)
    if num_samples is None:
        key = ("frames", str(wave.device), B, N)             # the same full-length batch every step: one fill kernel less per call
        frames = tb.get(key)
        if frames is None:
            frames = tb[key] = torch.full((B,), num_frames(N), dtype=torch.int32, device=wave.device)
        T = num_frames(N)
    else:
        frames = torch.clamp((num_samples.to(torch.int32) - FRAME) // HOP + 1, min=0)
        T = num_frames(N)
    if pad_frames_to:
        T = (T + pad_frames_to - 1) // pad_frames_to * pad_frames_to
    # utterance CMVN rewrites every element of the buffer, padding rows included (csrc/fbank.hip cmvn kernels): no fill pass in front of it
    out = (torch.empty((B, T, tables.num_mel), dtype=torch.float32, device=wave.device) if normalize == "utterance"
           else torch.full((B, T, tables.num_mel), float(padding_value), dtype=torch.float32, device=wave.device))
    st = torch.cuda.current_stream().cuda_stream
    L = _lib.lib()
    _lib.check(L.mi_fbank_f64(wave.data_ptr(), wave.stride(0), 0 if num_samples is None else num_samples.to(torch.int32).data_ptr(),
                              N, tb["window"].data_ptr(), tb["twiddle"].data_ptr(), tb["mel_t"].data_ptr(), tb["lo"].data_ptr(),
                              tb["hi"].data_ptr(), out.data_ptr(), T, B, tables.num_mel, MEL_FLOOR, PREEMPH, st), "mi_fbank_f64")
    if normalize == "utterance":
        _lib.check(L.mi_cmvn_utterance(out.data_ptr(), frames.data_ptr(), B, T, tables.num_mel, int(normalize_means),
                                       int(normalize_vars), float(padding_value), st), "mi_cmvn_utterance")
    elif normalize == "global":
        _lib.check(L.mi_cmvn_global(out.data_ptr(), out.numel(), tables.num_mel, global_means.data_ptr(), global_stds.data_ptr(), st),
                   "mi_cmvn_global")
    return out, frames


def strip_zeros_pad_gpu(wave, num_samples=None, min_len: int = 8000):
    """The array handling of the reference's `default_transform` (callbacks.py:108-118) for a device batch: every clip stripped of leading / trailing
    zero samples (`np.trim_zeros`, data_utils.py:173-177) and zero-padded to at least `min_len` samples.
    wave (B,N) float32 device tensor -> (stripped (B, max(N, min_len)) float32, eff_len (B,) int32 = samples the feature extractor sees)."""
    import torch

    from . import _lib
    B, N = wave.shape
    n_out = max(N, int(min_len))
    out = torch.empty((B, n_out), dtype=torch.float32, device=wave.device)
    meta = torch.empty((3, B), dtype=torch.int32, device=wave.device)
    ns = None if num_samples is None else num_samples.to(device=wave.device, dtype=torch.int32).contiguous()
    _lib.check(_lib.lib().mi_trim_zeros_pad_f32(wave.data_ptr(), wave.stride(0), 0 if ns is None else ns.data_ptr(), N, B, int(min_len), out.data_ptr(), out.stride(0),
                                                n_out, meta[0].data_ptr(), meta[1].data_ptr(), meta[2].data_ptr(), torch.cuda.current_stream().cuda_stream),
               "mi_trim_zeros_pad_f32")
    return out, meta[2]
