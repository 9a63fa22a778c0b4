"""CPU: SURVEY §8a rows 4-5 — the reference's dataloader-side harness (zero strip, pad to >= 8000 samples, feature extractor, SpeechCollatorWithPadding)
captured in tests/golden/harness.npz (make_golden.py `run_harness_cases`, which runs the reference's own callbacks.py / collators.py).  Here: the oracle and the
drop-in CPU extractor reproduce it through the same steps."""
import numpy as np
import torch
from transformers import BatchFeature

from helpers import load_golden
from huggingface_asr_amd.feature_extraction import CustomFeatureExtractor
from oracle import fbank_ref


def _clips(g):
    return [g["waves"][i, : int(n)] for i, n in enumerate(g["lens"])]


def _default_transform(w, min_len=8000):
    """callbacks.py:108-118 restated: np.trim_zeros (data_utils.py:173-177), zero-pad to >= 8000 samples"""
    w = np.trim_zeros(w)
    return np.pad(w, (0, min_len - w.shape[0])) if w.shape[0] < min_len else w


def test_oracle_and_cpu_extractor_reproduce_the_transform_chain():
    g = load_golden("harness")
    fe = CustomFeatureExtractor(feature_size=80, norm_type="utterance")
    off = 0
    for w, T in zip(_clips(g), g["frames"]):
        a = _default_transform(w)
        want = g["feats_cat"][off: off + int(T)]
        off += int(T)
        got_o = fbank_ref.extract(a)
        assert got_o.shape == want.shape
        np.testing.assert_allclose(got_o, want, atol=2e-5, rtol=0)
        got = fe(a, sampling_rate=16000, padding=False, return_attention_mask=False, return_tensors="np")["input_features"][0]
        np.testing.assert_allclose(np.asarray(got, np.float32), want, atol=2e-5, rtol=0)
    assert off == g["feats_cat"].shape[0]
    assert list(g["frames"]) == [161, 48, 148, 248]          # 26000 / 8000 (padded from 4800) / 24000 / 40001 samples


def test_drop_in_extractor_pads_like_the_reference_collator():
    """collators.py:82-88 calls `feature_extractor.pad(list of BatchFeature, padding=True, pad_to_multiple_of=100, return_tensors='pt')` on OUR class too"""
    g = load_golden("harness")
    fe = CustomFeatureExtractor(feature_size=80, norm_type="utterance")
    feats, off = [], 0
    for T in g["frames"]:
        feats.append(torch.from_numpy(g["feats_cat"][off: off + int(T)]))
        off += int(T)
    batch = fe.pad([BatchFeature({fe.model_input_names[0]: f}) for f in feats], padding=True, pad_to_multiple_of=100, return_tensors="pt")
    np.testing.assert_array_equal(batch["input_features"].numpy(), g["input_values"])
    np.testing.assert_array_equal(batch["attention_mask"].numpy(), g["attention_mask"])
    lab = g["labels"]
    assert lab.shape == (4, 6) and lab[1].tolist() == [5, -100, 4, -100, -100, -100]      # "b zzz a": unk -> -100 (mask_unks), pad -> -100


def test_oracle_model_on_the_collated_batch():
    from oracle import ebranchformer_ref as R
    from huggingface_asr_amd import shapes, synth
    g = load_golden("harness")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), int(g["seed"])).items()}
    loss, logits = R.ctc_forward(sd, cfg, torch.from_numpy(g["input_values"]), torch.from_numpy(g["attention_mask"]), torch.from_numpy(g["labels"]))
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-4, rtol=0)
    assert abs(float(loss) - float(g["loss"])) < 2e-4 * float(g["loss"])       # labels with -100 in the MIDDLE of a row (masked unk)
