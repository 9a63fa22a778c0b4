"""CPU: pin the oracle (oracle/*.py) against golden vectors made by importing the reference
(tests/golden/make_golden.py).  fp32, tolerance 1e-4 abs on logits (observed ~1e-5)."""
import numpy as np
import pytest
import torch

from helpers import case_inputs, load_golden
from huggingface_asr_amd import shapes
from oracle import ebranchformer_ref as R
from oracle import fbank_ref


def _cfg(base, **kw):
    c = dict(base)
    c.update(ctc_zero_infinity=True, ctc_loss_reduction="mean")
    c.update(kw)
    return c


TINY_CASES = [
    ("tiny_rel", _cfg(shapes.TINY)),
    ("tiny_rotary", _cfg(shapes.TINY, position_embeddings_type="rotary")),
    ("tiny_causal", _cfg(shapes.TINY, is_causal=True)),
    ("tiny_nomacaron", _cfg(shapes.TINY, csgu_activation="gelu", csgu_use_linear_after_conv=True)),
    # context-aware front ends (extractors.py:23-65); the third is the recipes' misspelt `shared_gated`, which the reference resolves to the plain conv
    ("tiny_gated", _cfg(shapes.TINY, context_awareness_type="gated")),
    ("tiny_gated_shared", _cfg(shapes.TINY, context_awareness_type="gated_shared")),
    ("tiny_shared_gated_fallthrough", _cfg(shapes.TINY, context_awareness_type="shared_gated")),
]


@pytest.mark.parametrize("name,cfg", TINY_CASES, ids=[c[0] for c in TINY_CASES])
def test_tiny_full_tensors(name, cfg):
    g = load_golden(name)
    sd, x, am, lab = case_inputs(g, cfg)
    with torch.no_grad():
        hidden, layers = R.encoder_forward(sd, cfg, x, am, return_layers=True)
        loss, logits = R.ctc_forward(sd, cfg, x, am, lab)
    np.testing.assert_allclose(hidden.numpy(), g["last_hidden"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=2e-4, rtol=0)
    # hidden_states[i] of the reference = input of layer i; layer_in_{i+1} = output of layer i
    for i in range(1, cfg["num_hidden_layers"]):
        np.testing.assert_allclose(layers[i - 1].numpy(), g[f"layer_in_{i}"], atol=2e-4, rtol=0)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    # quirk 8': outer (CTC) lengths differ from inner (mask) lengths
    np.testing.assert_array_equal(R.conv_out_lengths_outer(am.sum(-1), cfg).numpy(), g["outer_lens"])
    np.testing.assert_array_equal(R.conv_out_lengths_inner(am.sum(-1), cfg).numpy(), g["inner_lens"])


BIG_CASES = [
    ("small_rel", _cfg(shapes.SMALL)),
    ("small_causal", _cfg(shapes.SMALL, is_causal=True)),
    ("base_rel", _cfg(shapes.BASE)),
    ("base_rotary", _cfg(shapes.BASE, position_embeddings_type="rotary")),
    ("small_gated", _cfg(shapes.SMALL, context_awareness_type="gated")),
]


@pytest.mark.parametrize("name,cfg", BIG_CASES, ids=[c[0] for c in BIG_CASES])
def test_small_base_slices(name, cfg):
    g = load_golden(name)
    sd, x, am, lab = case_inputs(g, cfg)
    torch.set_num_threads(8)
    with torch.no_grad():
        loss, logits = R.ctc_forward(sd, cfg, x, am, lab)
    lg = logits.numpy()
    np.testing.assert_allclose(lg[:, ::25, :64], g["logits_slice"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(lg[:, :, -1], g["logits_blank"], atol=5e-4, rtol=0)
    assert abs(float(lg.std()) - float(g["logits_std"])) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))


def test_lengths_table():
    g = load_golden("lengths")
    L = torch.from_numpy(g["L"])
    cfg = dict(shapes.TINY)
    np.testing.assert_array_equal(R.conv_out_lengths_inner(L, cfg).numpy(), g["inner"])
    np.testing.assert_array_equal(R.conv_out_lengths_outer(L, cfg).numpy(), g["outer"])
    np.testing.assert_array_equal(R.conv_out_lengths_inner(L, dict(cfg, is_causal=True)).numpy(), g["inner_causal"])
    assert int(R.conv_out_lengths_inner(torch.tensor([998]), cfg)) == 250
    assert int(R.conv_out_lengths_outer(torch.tensor([998]), cfg)) == 248


@pytest.mark.parametrize("wave", ["sweep", "noise", "silence_padded"])
def test_fbank_matches_reference(wave):
    g = load_golden("fbank")
    raw = fbank_ref.fbank(g[f"{wave}_wave"])
    assert raw.shape == g[f"{wave}_raw"].shape
    np.testing.assert_allclose(raw, g[f"{wave}_raw"], atol=2e-5, rtol=0)
    cm = fbank_ref.utterance_cmvn(raw, raw.shape[0])
    np.testing.assert_allclose(cm, g[f"{wave}_cmvn"], atol=2e-5, rtol=0)


def test_fbank_global_norm():
    g = load_golden("fbank")
    out = fbank_ref.extract(g["noise_wave"], "global", g["global_means"], g["global_stds"])
    np.testing.assert_allclose(out, g["noise_global"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("case", ["basic", "repeat", "infeasible", "empty_target"])
def test_ctc_known_answers(case):
    g = load_golden("ctc_known")
    logits = torch.from_numpy(g[f"{case}/logits"])
    labels = torch.from_numpy(g[f"{case}/labels"])
    in_len = torch.from_numpy(g[f"{case}/in_len"])
    tl = (labels >= 0).sum(-1)
    lp = torch.log_softmax(logits, -1)
    for zi in (0, 1):
        for red in ("mean", "sum", "none"):
            want = g[f"{case}/{red}/{zi}"]
            got = R.ctc_loss_ref(lp, labels, in_len, tl, blank=logits.shape[-1] - 1, reduction=red, zero_infinity=bool(zi))
            np.testing.assert_allclose(got.numpy(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", ["a", "b", "c", "m"])          # "m": ctc_margin 6 in the reference = no effect (its processor never passes att_w)
def test_ctc_prefix_scorer_matches_reference(case):
    """oracle/ctc_prefix_ref.py against 4 decoding steps of the reference's CTCRescorerLogitsProcessor."""
    from oracle import ctc_prefix_ref as P
    g = load_golden("ctc_prefix")
    B, W, T, O, blank, eos, space, trick = [int(v) for v in g[f"{case}/meta"]][:8]
    logp = torch.log_softmax(torch.from_numpy(g[f"{case}/enc_logits"]), -1).numpy()
    sc = P.PrefixScorer(logp, g[f"{case}/lens"], blank, W)
    for step in range(4):
        ctc = sc.step(g[f"{case}/step{step}/input_ids"])
        want = g[f"{case}/step{step}/ctc"]
        live = want > -1e9
        np.testing.assert_array_equal(ctc > -1e9, live)
        np.testing.assert_allclose(ctc[live], want[live], atol=2e-4, rtol=1e-5)
        if not trick:
            out = P.rescore(g[f"{case}/step{step}/att"], ctc, blank, 0.3)
            wo = g[f"{case}/step{step}/out"]
            ok = wo > -1e9
            np.testing.assert_allclose(out[ok], wo[ok], atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("name,fixed", [("aed_tiny", False), ("aed_tiny_fixedpos", True)])
def test_joint_aed_forward_matches_reference(name, fixed):
    """oracle/aed_ref.py against JointCTCAttentionEncoderDecoder.forward of the reference (3 losses, both logits)."""
    from helpers import AED_JCFG, TINY_DEC, aed_case_inputs
    from oracle import aed_ref as A
    g = load_golden(name)
    sd, x, am, lab = aed_case_inputs(g)
    enc_cfg = _cfg(shapes.TINY)
    dec_cfg = dict(TINY_DEC, pos_emb_fixed=fixed)
    with torch.no_grad():
        out = A.joint_forward(sd, enc_cfg, dec_cfg, AED_JCFG, x, am, lab)
    np.testing.assert_allclose(out["encoder_logits"].numpy(), g["encoder_logits"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out["encoder_hidden"].numpy(), g["encoder_hidden"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(out["logits"].numpy(), g["logits"], atol=3e-4, rtol=0)
    for k in ("loss", "enc_loss", "dec_loss"):
        assert abs(float(out[k]) - float(g[k])) < 1e-4 * abs(float(g[k])), k


def test_whisper_frontend_and_encoder_match_transformers():
    import ast
    from huggingface_asr_amd import synth
    from oracle import whisper_ref as W
    g = load_golden("whisper")
    for k in ("noise", "tone"):
        got = W.log_mel(g[f"fe/{k}_wave"])
        assert got.shape == (80, 3000)
        np.testing.assert_allclose(got, g[f"fe/{k}_logmel"], atol=2e-5, rtol=0)
    seed = int(g["seed"])
    sd = {str(n): torch.from_numpy(synth.init_param(seed, str(n), ast.literal_eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])}
    cfg = dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256)
    x = torch.from_numpy(synth.normal(seed, "wh_feats", (2, 80, 200), 0.5))
    with torch.no_grad():
        out = W.encoder_forward(sd, cfg, x)
    np.testing.assert_allclose(out.numpy(), g["enc_out"], atol=2e-4, rtol=0)


def test_bestrq_oracle_matches_reference_fixture():
    """BEST-RQ (src/models/bestrq.py): quantizer targets bit-exact, loss, last hidden state and every gradient of the oracle (autograd) vs
    the imported reference model run with the same injected masking noise."""
    import torch
    from helpers import BESTRQ_CFG, bestrq_case_inputs
    from huggingface_asr_amd import shapes, synth
    from oracle import bestrq_ref as Bq
    g = load_golden("bestrq_tiny")
    sd, x, am, mask = bestrq_case_inputs(g)
    cfg = dict(shapes.TINY, **BESTRQ_CFG)
    B, T2 = mask.shape
    L, d = cfg["num_hidden_layers"], cfg["hidden_size"]
    noise = torch.from_numpy(synth.mask_noise(int(g["seed"]), ((0 * 64 + L) * 16 + 3), (B, T2, d), 0.1))
    sdr = {k: (v.clone().requires_grad_(True) if not k.startswith("rpq.") else v) for k, v in sd.items()}
    out = Bq.forward(sdr, cfg, x, am, mask, noise)
    tg = Bq.rpq_targets(x.reshape(B, T2, -1), sd["rpq.P"], sd["rpq.CB"])
    assert np.array_equal(tg.numpy(), g["targets"])
    assert abs(float(out["loss"].detach()) - float(g["loss"])) <= 1e-4 * float(g["loss"])
    np.testing.assert_allclose(out["last_hidden"].detach().numpy(), g["last_hidden"], atol=2e-4, rtol=0)
    out["loss"].backward()
    for k in g.files:
        if k.startswith("grad:"):
            got, want = sdr[k[5:]].grad.numpy(), g[k]
            assert np.abs(got - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), k


@pytest.mark.parametrize("name", ["finetune_tiny_mix_extra", "finetune_tiny_mix", "finetune_tiny_extra"])
def test_finetune_head_oracle_matches_reference(name):
    """oracle.finetune_ctc_forward (layer mixing / additional layer of BestRQEBranchformerForCTC, bestrq.py:212-322) against the reference's eval outputs"""
    g = load_golden(name)
    extra, mix = (bool(v) for v in g["flags"])
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean", finetune_with_additional_layer=extra, finetune_with_layer_mixing=mix)
    sd, x, am, lab = case_inputs(g, cfg)
    loss, logits = R.finetune_ctc_forward(sd, cfg, x, am, lab)
    assert float((logits - torch.from_numpy(g["eval_logits"])).abs().max()) < 2e-4
    assert abs(float(loss) - float(g["eval_loss"])) < 2e-4 * abs(float(g["eval_loss"]))
