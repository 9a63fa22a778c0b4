"""GPU parity of the device SpecAugment (csrc/specaug.hip + huggingface_asr_amd/augment.py) against the reference's
src/augmentations/spec_aug.py run with the recipe parameters (tests/golden/specaug.npz): seeding torch alike reproduces the warp centre /
target and every mask, the bicubic warp matches torch.nn.functional.interpolate to fp32 rounding."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from huggingface_asr_amd import synth

pytestmark = pytest.mark.gpu
RECIPE = dict(apply_time_warp=True, time_warp_window=5, time_warp_mode="bicubic", apply_freq_mask=True, freq_mask_width_range=[0, 27],
              num_freq_mask=2, apply_time_mask=True, time_mask_width_ratio_range=[0, 0.05], num_time_mask=5)


@pytest.mark.parametrize("name", ["equal", "single", "ragged", "short"])
def test_specaug_matches_reference(name):
    from huggingface_asr_amd.augment import SpecAug
    g = load_golden("specaug")
    B, T, seed = [int(v) for v in g[name + "/shape"]]
    lens = [int(v) for v in g[name + "/lens"]]
    x = torch.from_numpy(synth.normal(seed, "specaug/" + name, (B, T, 80), 1.0))
    aug = SpecAug(**RECIPE)
    torch.manual_seed(100 + seed)
    y, _ = aug(x.to("cuda:0"), torch.tensor(lens) if lens else None)
    want = g[name + "/out"]
    got = y.cpu().numpy()
    assert got.shape == want.shape
    assert ((got == 0) == (want == 0)).mean() > 0.9999          # identical masks / padding
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
    assert np.abs(want - x.numpy()).max() > 0.1 or name == "short"   # the fixture really augments


def test_specaug_constructor_errors_and_device_only():
    from huggingface_asr_amd.augment import SpecAug
    with pytest.raises(ValueError):
        SpecAug(apply_time_warp=False, apply_freq_mask=False, apply_time_mask=False)
    with pytest.raises(ValueError):
        SpecAug(time_mask_width_range=5, time_mask_width_ratio_range=0.05)
    with pytest.raises(RuntimeError):
        SpecAug(**RECIPE)(torch.zeros(1, 50, 80))
