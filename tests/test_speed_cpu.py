"""CPU: the speed-perturbation oracle (oracle/speed_ref.py — PARITY UNPINNED, torchaudio is absent) checked through properties of the
published algorithm, and the host-side kernel table of the product against the oracle's."""
import math

import numpy as np
import pytest
import torch

from oracle import speed_ref as S


def test_kernel_table_matches_between_host_and_oracle_and_has_unit_dc_gain():
    from huggingface_asr_amd.augment import _sinc_resample_kernel
    for orig, new in ((9, 10), (11, 10), (2, 1), (1, 2)):
        k, w = S.resample_kernel(orig, new)
        kt, wt = _sinc_resample_kernel(orig, new)
        assert w == wt and k.shape == (new, 2 * w + orig)
        np.testing.assert_array_equal(k, kt.numpy())
        # every phase is a low-pass interpolator: its taps sum to ~1 (DC passes unchanged)
        assert np.all(np.abs(k.sum(1) - 1.0) < 2e-2), k.sum(1)
    assert S.resample_kernel(9, 10)[1] == 7 and S.resample_kernel(11, 10)[1] == 7           # ceil(6 * orig / (0.99 * min))


@pytest.mark.parametrize("factor", [0.9, 1.0, 1.1])
def test_speed_lengths_identity_and_sine_frequency(factor):
    sr, f0, n = 16000, 440.0, 8000
    t = np.arange(n) / sr
    x = np.sin(2 * np.pi * f0 * t).astype(np.float32)[None]
    y, ol = S.speed(x, sr, factor, lengths=[n, n // 2])
    src, tgt = int(factor * sr), sr
    g = math.gcd(src, tgt)
    assert y.shape[1] == math.ceil(n * (tgt // g) / (src // g))
    assert ol.tolist() == [math.ceil(n * tgt / src), math.ceil(n // 2 * tgt / src)]
    if factor == 1.0:
        np.testing.assert_array_equal(y, x)
        return
    # played at `factor` speed the tone moves to f0 * factor... in samples: the SAME waveform stretched by 1/factor, i.e. a sine of f0 * factor at rate sr
    m = y.shape[1]
    want = np.sin(2 * np.pi * f0 * factor * np.arange(m) / sr).astype(np.float32)
    core = slice(200, m - 200)                                      # away from the zero-padded edges
    assert np.abs(y[0, core] - want[core]).max() < 5e-3
    # DC gain ~ 1
    dc, _ = S.speed(np.ones((1, 4000), np.float32), sr, factor)
    assert np.abs(dc[0, 100:-100] - 1.0).max() < 2e-2
