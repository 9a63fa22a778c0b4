"""GPU parity for the Whisper-style path (BASELINE config 4): HIP log-mel against transformers' WhisperFeatureExtractor
outputs, HIP encoder against transformers' WhisperEncoder outputs (tests/golden/whisper.npz) and the bf16-modelled oracle."""
import ast

import numpy as np
import pytest
import torch

from helpers import load_golden
from huggingface_asr_amd import synth
from oracle import whisper_ref as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("wave", ["noise", "tone"])
def test_whisper_logmel_matches_transformers(wave):
    from huggingface_asr_amd.whisper import WhisperFrontend
    g = load_golden("whisper")
    fe = WhisperFrontend(80)
    w = torch.from_numpy(g[f"fe/{wave}_wave"])[None].to(DEV)
    feats, cl = fe(w)
    assert feats.shape == (1, 80, 3000) and cl.shape == (1, 3000, 80)
    np.testing.assert_allclose(feats[0].cpu().numpy(), g[f"fe/{wave}_logmel"], atol=3e-5, rtol=0)
    np.testing.assert_allclose(cl[0].float().cpu().numpy().T, g[f"fe/{wave}_logmel"], atol=1e-2, rtol=0)


def test_whisper_encoder_matches_transformers_and_oracle():
    from huggingface_asr_amd.whisper import WhisperEncoderEngine
    g = load_golden("whisper")
    seed = int(g["seed"])
    sd = {str(n): torch.from_numpy(synth.init_param(seed, str(n), ast.literal_eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])}
    cfg = dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256)
    x = torch.from_numpy(synth.normal(seed, "wh_feats", (2, 80, 200), 0.5))
    eng = WhisperEncoderEngine(cfg, DEV)
    eng.load_state_dict(sd)
    out = eng.forward(x.to(DEV)).cpu().numpy()
    d = np.abs(out - g["enc_out"])
    assert d.max() < 0.08 and d.mean() < 0.01, (d.max(), d.mean())
    with torch.no_grad():
        oq = W.encoder_forward(sd, cfg, x, q=lambda t: t.to(torch.bfloat16).float()).numpy()
    dq = np.abs(out - oq)
    assert dq.max() < 0.04 and dq.mean() < 0.004, (dq.max(), dq.mean())
    with pytest.raises(ValueError):
        eng.forward(x[:, :, :100].to(DEV))
