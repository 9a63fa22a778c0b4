"""GPU: device speed perturbation (csrc/speed.hip behind huggingface_asr_amd.augment.Speed / SpeedPerturbation) against the oracle restatement
of torchaudio's resampler (oracle/speed_ref.py; parity unpinned — torchaudio is not installed — see its header)."""
import numpy as np
import pytest
import torch

from oracle import speed_ref as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("factor", [0.9, 1.1, 1.0])
def test_speed_matches_oracle(factor):
    from huggingface_asr_amd.augment import Speed
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, 16000 + 37)).astype(np.float32)
    lens = torch.tensor([16037, 12000, 16037, 801, 9], dtype=torch.int32)
    x[1, 12000:] = 0; x[3, 801:] = 0; x[4, 9:] = 0
    want, wl = S.speed(x, 16000, factor, lens.numpy())
    got, gl = Speed(16000, factor)(torch.from_numpy(x).to(DEV), lens.to(DEV))
    assert got.shape == want.shape and gl.cpu().tolist() == wl.tolist()
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=3e-6 * max(1.0, np.abs(want).max()), rtol=0)
    one, _ = Speed(16000, factor)(torch.from_numpy(x[2]).to(DEV))               # 1-D input, no lengths
    np.testing.assert_allclose(one.cpu().numpy(), want[2], atol=3e-6 * max(1.0, np.abs(want).max()), rtol=0)


def test_speed_perturbation_draws_like_torchaudio_and_refuses_cpu():
    from huggingface_asr_amd.augment import SpeedPerturbation
    sp = SpeedPerturbation(16000, [0.9, 1.0, 1.1])
    x = torch.randn(2, 4000, device=DEV)
    torch.manual_seed(11)
    picks = [int(torch.randint(3, ())) for _ in range(6)]
    torch.manual_seed(11)
    for k in picks:
        y, _ = sp(x)
        want_n = {0: -(-10 * 4000 // 9), 1: 4000, 2: -(-10 * 4000 // 11)}[k]
        assert y.shape == (2, want_n)
    with pytest.raises(RuntimeError):
        SpeedPerturbation(16000, [0.9])(torch.randn(1, 100))
