"""CPU: the decoding-loop oracle (oracle/generate_ref.py) against the reference's OWN `generate()` — fixtures tests/golden/gen_*.npz, written by
tests/golden/make_golden.py `gen` from `JointCTCAttentionEncoderDecoder.generate` (src/models/ctc_encoder_plus_autoregressive_decoder.py:450-482) called the way
`do_generate` calls it (src/utilities/general_utils.py:198-218) — and the argument contract of the drop-in `generate()` that needs no GPU."""
import numpy as np
import pytest
import torch

import gen_model as GM
from helpers import AED_JCFG, gen_case_inputs
from huggingface_asr_amd import shapes
from oracle import generate_ref as G

ENC = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")


@pytest.mark.parametrize("name", list(GM.CASES))
def test_oracle_loop_reproduces_the_reference_generate(name):
    """Greedy and beam search (3 / 5 beams; length penalties 0.6 / 1.0 / 1.6; early_stopping False / True / "never"; max_length 14 — some hypotheses close on EOS, the
    rest when the length runs out — and 5 / 6): the oracle in plain fp32 must return the reference's sequences token for token, in its order, with its scores."""
    torch.set_num_threads(8)
    g, sd, x, am, dec_cfg = gen_case_inputs(name)
    n_eos = n_open = 0
    for W, lp, es, ml in GM.SETTINGS:
        key = GM.setting_key(W, lp, es, ml)
        want = g[key + "/sequences"]
        fn, B = G.joint_score_fn(sd, ENC, dec_cfg, AED_JCFG, x, am, W, 0.3)
        if W == 1:
            seq = G.greedy(fn, B, max_length=ml, eos=GM.EOS, pad=GM.PAD, start=GM.START)
        else:
            seq, sc = G.beam_search(fn, B, W, GM.V, max_length=ml, eos=GM.EOS, pad=GM.PAD, start=GM.START, length_penalty=lp, early_stopping=es)
            assert np.abs(sc - g[key + "/sequences_scores"]).max() < 1e-5, key
            n_eos += int((want == GM.EOS).any(1).sum())
            n_open += int((~(want == GM.EOS).any(1)).sum())
        assert seq.shape == want.shape and (seq == want).all(), (key, seq, want)
    assert n_eos >= 20 and n_open >= 20, (n_eos, n_open)          # the fixture holds both kinds of closure


def test_beam_loop_with_one_beam_yields_the_greedy_tokens():
    """transformers decodes num_beams = 1 with its greedy loop; the HIP loop runs its beam kernel with one beam.  Same tokens (the fixture's greedy rows)."""
    g, sd, x, am, dec_cfg = gen_case_inputs("gen_tiny")
    for W, lp, es, ml in GM.SETTINGS:
        if W != 1:
            continue
        fn, B = G.joint_score_fn(sd, ENC, dec_cfg, AED_JCFG, x, am, 1, 0.3)
        seq, _ = G.beam_search(fn, B, 1, GM.V, max_length=ml, eos=GM.EOS, pad=GM.PAD, start=GM.START)
        want = g[GM.setting_key(W, lp, es, ml) + "/sequences"]
        assert (seq == want[:, : seq.shape[1]]).all() and (want[:, seq.shape[1]:] == GM.PAD).all()


def _cpu_model():
    from test_surface_cpu import _joint_model
    return _joint_model(False).eval()


def test_generate_refuses_what_it_does_not_implement():
    """The reference's generate() accepts every transformers option; the HIP loop implements the ones the reference's call sites use and RAISES for the rest —
    before touching the device, so this runs without one."""
    from huggingface_asr_amd.decoding import GenerationConfigCustom
    m = _cpu_model()
    x = torch.zeros(1, 200, 80)
    m.generation_config = GenerationConfigCustom(pad_token_id=50, eos_token_id=1, decoder_start_token_id=2, num_beams=3, max_length=8, ctc_weight=0.3)
    for bad in (dict(do_sample=True), dict(repetition_penalty=1.2), dict(no_repeat_ngram_size=3), dict(min_length=4), dict(num_beam_groups=3, diversity_penalty=0.5),
                dict(forced_eos_token_id=1), dict(suppress_tokens=[3]), dict(lm_weight=0.5)):
        with pytest.raises(NotImplementedError):
            m.generate(input_values=x, **bad)
    with pytest.raises(ValueError, match="not used by the model"):
        m.generate(input_values=x, attention_maks=torch.ones(1, 200))
    with pytest.raises(ValueError, match="num_return_sequences"):
        m.generate(input_values=x, num_return_sequences=4)
    with pytest.raises(ValueError, match="num_return_sequences"):
        m.generate(input_values=x, num_beams=1, num_return_sequences=2)
    with pytest.raises(NotImplementedError):
        m.generate(input_values=x, decoder_input_ids=torch.zeros(1, 2, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        m.generate(input_values=x, synced_gpus=True)
    with pytest.raises(ValueError, match="num_beams"):            # the reference builds its CTC scorer for the MODEL configuration's beam count
        m.generate(input_values=x, num_beams=5)
    with pytest.raises(RuntimeError, match="GPU"):                # a valid request reaches the device check: no CPU fallback
        m.generate(input_values=x, attention_mask=torch.ones(1, 200, dtype=torch.long), labels=torch.zeros(1, 3, dtype=torch.long), output_hidden_states=True)
