"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of `torchaudio.transforms.SpeedPerturbation` — the first step of the reference's training pre-processing
(configs/default_data_preprocessing2d.json:3-19: orig_freq 16000, factors [0.9, 1.0, 1.1]).

PARITY UNPINNED: the arithmetic lives in a third-party dependency that is absent from /root/reference and from this image
(torchaudio==2.5.0, requirements.txt:15; `import torchaudio` fails here), so no golden vector could be generated.  What follows restates
torchaudio's published algorithm:
  * transforms/_transforms.py `SpeedPerturbation.forward`: one `torch.randint(len(factors), ())` per call picks the factor;
    `Speed.forward` -> functional.speed (factor 1.0 returns the input unchanged);
  * functional/functional.py `speed`: source = int(factor * orig_freq), target = int(orig_freq), both divided by their gcd;
    out_lengths = ceil(lengths * target / source); waveform -> resample(waveform, source, target);
  * functional/functional.py `_get_sinc_resample_kernel` (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; index arithmetic in
    float64, kernel cast to float32) and `_apply_sinc_resample_kernel` (zero-pad width left / width + orig right, conv1d with stride orig,
    interleave the `new` phases, cut to ceil(new * length / orig)).
The tests pin it through properties instead (identity at factor 1, length formula, a resampled sine keeps its shape at the scaled
frequency, DC gain) and compare the HIP kernel with it."""
import math

import numpy as np


def resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (kernel (new, 2*width + orig) float32, width); orig / new already divided by their gcd."""
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t *= base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * scale
    return k.astype(np.float32), width


def resample(wave: np.ndarray, orig: int, new: int) -> np.ndarray:
    """wave (B, N) float32 -> (B, ceil(new * N / orig)) float32"""
    g = math.gcd(int(orig), int(new))
    orig, new = int(orig) // g, int(new) // g
    if orig == new:
        return wave.copy()
    k, width = resample_kernel(orig, new)
    B, N = wave.shape
    x = np.pad(wave.astype(np.float32), ((0, 0), (width, width + orig)))
    nj = (x.shape[1] - k.shape[1]) // orig + 1
    out = np.zeros((B, nj, new), np.float32)
    for j in range(nj):
        seg = x[:, j * orig:j * orig + k.shape[1]]                     # (B, kw)
        out[:, j, :] = seg @ k.T
    out = out.reshape(B, nj * new)
    return out[:, :math.ceil(new * N / orig)]


def speed(wave: np.ndarray, orig_freq: int, factor: float, lengths=None):
    """torchaudio.functional.speed -> (wave', lengths')"""
    src, tgt = int(factor * orig_freq), int(orig_freq)
    g = math.gcd(src, tgt)
    src, tgt = src // g, tgt // g
    out_len = None if lengths is None else np.ceil(np.asarray(lengths) * tgt / src).astype(np.int64)
    if src == tgt:
        return wave.copy(), out_len
    return resample(wave, src, tgt), out_len
