"""ORACLE (test infrastructure, never shipped): CPU restatement of the Whisper front end + encoder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference reaches Whisper through HuggingFace only (`configs/default_data_preprocessing_whisper.json:20-29` ->
`WhisperFeatureExtractor`; `src/utilities/model_utils.py:183` / `src/trainers/train_enc_dec_asr.py:82-83` ->
`WhisperForConditionalGeneration`), i.e. the arithmetic lives in transformers (pinned 4.39.3, numpy feature path):
  * `models/whisper/feature_extraction_whisper.py` `_np_extract_fbank_features` (installed 5.15.0 :105-133) + padding to 30 s,
    `audio_utils.spectrogram` (center=True reflect padding, periodic hann(400), 400-point rFFT stored as complex64, power 2,
    Slaney mel 80x201 with slaney norm, log10 with floor 1e-10), drop the last frame, clamp to max-8, (x+4)/4;
  * `models/whisper/modeling_whisper.py` WhisperEncoder (:540-646) / WhisperEncoderLayer (:360-413) / WhisperAttention
    (k_proj has no bias, scaling head_dim**-0.5).
Pinned by tests/golden/whisper_*.npz (made by tests/golden/make_golden.py from the installed transformers classes).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

N_FFT, HOP, N_BINS = 400, 160, 201


def slaney_mel_filters(num_mel=80, sr=16000, fmin=0.0, fmax=8000.0):
    """audio_utils.mel_filter_bank(..., norm='slaney', mel_scale='slaney') -> (201, num_mel)."""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        mels = 3.0 * f / 200.0
        logstep = 27.0 / np.log(6.4)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mels)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        logstep = np.log(6.4) / 27.0
        return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), 200.0 * m / 3.0)

    mel_freqs = np.linspace(hz2mel(fmin), hz2mel(fmax), num_mel + 2)
    filter_freqs = mel2hz(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, N_BINS)
    diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    enorm = 2.0 / (filter_freqs[2: num_mel + 2] - filter_freqs[:num_mel])
    return fb * enorm[None, :]


def log_mel(waveform: np.ndarray, n_samples: int = 480000, num_mel: int = 80) -> np.ndarray:
    """(N,) float32 -> (num_mel, n_samples // 160) float32, clip zero-padded / truncated to n_samples first."""
    w = np.zeros(n_samples, dtype=np.float32)
    w[: min(len(waveform), n_samples)] = waveform[:n_samples]
    x = np.pad(w.astype(np.float64), (N_FFT // 2, N_FFT // 2), mode="reflect")
    T = 1 + (x.size - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(T)[:, None]
    n = np.arange(N_FFT)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)             # np.hanning(401)[:-1]
    spec = np.fft.rfft(x[idx] * window[None, :], axis=1).astype(np.complex64)
    power = np.abs(spec, dtype=np.float64) ** 2.0
    mel = np.maximum(1e-10, power @ slaney_mel_filters(num_mel))
    ls = np.log10(mel).astype(np.float32).T                           # (num_mel, T)
    ls = ls[:, :-1]
    ls = np.maximum(ls, ls.max() - 8.0)
    return ((ls + 4.0) / 4.0).astype(np.float32)


def encoder_forward(sd: dict, cfg: dict, feats: torch.Tensor, q=None) -> torch.Tensor:
    """WhisperEncoder.forward (eval): feats (B, num_mel, 2*P) -> (B, P, d)."""
    q = q or (lambda t: t)
    d, H, L = cfg["d_model"], cfg["encoder_attention_heads"], cfg["encoder_layers"]
    hd = d // H
    x = q(F.gelu(F.conv1d(q(feats), q(sd["conv1.weight"]), sd["conv1.bias"], padding=1)))
    x = q(F.gelu(F.conv1d(x, q(sd["conv2.weight"]), sd["conv2.bias"], stride=2, padding=1)))
    x = x.permute(0, 2, 1) + sd["embed_positions.weight"][None]
    B, T, _ = x.shape
    for l in range(L):
        p = f"layers.{l}."
        h = q(F.layer_norm(x, (d,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"]))
        qq = q(F.linear(h, q(sd[p + "self_attn.q_proj.weight"]), sd[p + "self_attn.q_proj.bias"])).view(B, T, H, hd).transpose(1, 2)
        kk = q(F.linear(h, q(sd[p + "self_attn.k_proj.weight"]))).view(B, T, H, hd).transpose(1, 2)
        vv = q(F.linear(h, q(sd[p + "self_attn.v_proj.weight"]), sd[p + "self_attn.v_proj.bias"])).view(B, T, H, hd).transpose(1, 2)
        a = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(hd), -1) @ vv
        a = q(a.transpose(1, 2).reshape(B, T, d))
        x = x + F.linear(a, q(sd[p + "self_attn.out_proj.weight"]), sd[p + "self_attn.out_proj.bias"])
        h = q(F.layer_norm(x, (d,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"]))
        m = q(F.gelu(F.linear(h, q(sd[p + "fc1.weight"]), sd[p + "fc1.bias"])))
        x = x + F.linear(m, q(sd[p + "fc2.weight"]), sd[p + "fc2.bias"])
    return F.layer_norm(x, (d,), sd["layer_norm.weight"], sd["layer_norm.bias"])
