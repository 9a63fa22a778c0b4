"""GPU: SURVEY §8a rows 4-5 on the device — the GPU pre-stage (zero strip + pad to >= 8000 samples + log-mel + CMVN + collator padding) against the batch the
reference's own dataloader harness produced (tests/golden/harness.npz = callbacks.py:100-118 + collators.py:65-106 run by make_golden.py), and the HIP
model on that batch against the reference model's loss / logits."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from huggingface_asr_amd import shapes, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_device_prestage_reproduces_the_reference_harness_batch():
    from huggingface_asr_amd.feature_extraction import CustomFeatureExtractor
    g = load_golden("harness")
    fe = CustomFeatureExtractor(feature_size=80, norm_type="utterance")
    waves = torch.from_numpy(g["waves"]).to(DEV)
    waves[1, int(g["lens"][1]):] = 7.0           # samples behind a clip's own length are not the clip: garbage there must not leak in
    feats, mask = fe.extract_on_device(waves, torch.from_numpy(g["lens"]).to(DEV), pad_to_multiple_of=100)
    torch.cuda.synchronize()
    assert tuple(feats.shape) == g["input_values"].shape
    np.testing.assert_array_equal(mask.cpu().numpy(), g["attention_mask"])
    np.testing.assert_allclose(feats.cpu().numpy(), g["input_values"], atol=2e-5, rtol=0)
    # the strip kernel itself: offsets / lengths / zero fill
    from huggingface_asr_amd.fbank import strip_zeros_pad_gpu
    out, eff = strip_zeros_pad_gpu(waves, torch.from_numpy(g["lens"]).to(DEV))
    assert eff.cpu().tolist() == [26000, 8000, 24000, 40001]
    o = out.cpu().numpy()
    for i, n in enumerate(g["lens"]):
        t = np.trim_zeros(g["waves"][i, : int(n)])
        np.testing.assert_array_equal(o[i, : len(t)], t)
        assert not o[i, len(t):].any()
    # an all-zero clip strips to nothing and is padded to 8000 zeros; without the transform the buffer's own length is kept
    z, ez = strip_zeros_pad_gpu(torch.zeros(2, 12000, device=DEV))
    assert ez.cpu().tolist() == [8000, 8000] and not z.cpu().numpy().any()
    f2, m2 = fe.extract_on_device(torch.from_numpy(g["waves"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV), default_transform=False, trim_to_longest=False)
    assert f2.shape[1] == 298 and m2.sum(-1).cpu().tolist() == [186, 28, 148, 248]


def test_hip_model_on_the_reference_collated_batch():
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    from huggingface_asr_amd.modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
    g = load_golden("harness")
    base = dict(shapes.TINY); base.pop("num_fbanks")
    cfg = Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    model = Wav2Vec2EBranchformerForCTC(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(dict(shapes.TINY)), int(g["seed"])).items()}
    model.load_state_dict(sd, strict=False)
    model.to(DEV).eval()
    with torch.no_grad():
        out = model(input_values=torch.from_numpy(g["input_values"]).to(DEV), attention_mask=torch.from_numpy(g["attention_mask"]).to(DEV),
                    labels=torch.from_numpy(g["labels"]).to(DEV))          # labels with -100 mid-row (masked unk) and at the end (padding)
    d = np.abs(out.logits.float().cpu().numpy() - g["logits"])
    assert d.max() < 0.06 and d.mean() < 0.009, (d.max(), d.mean())
    assert abs(float(out.loss) - float(g["loss"])) < 1e-3 * float(g["loss"]), (float(out.loss), float(g["loss"]))
