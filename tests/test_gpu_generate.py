"""GPU parity of `JointCTCAttentionEncoderDecoder.generate` (HIP) with the reference's own `generate()`: fixtures tests/golden/gen_*.npz, written by
tests/golden/make_golden.py `gen` from the reference joint model (src/models/ctc_encoder_plus_autoregressive_decoder.py:450-482, processors :360-404) decoded the way
`do_generate` decodes (src/utilities/general_utils.py:198-218) — greedy, 3 and 5 beams, three length penalties, early_stopping False / True / "never", hypotheses closed by
EOS and by max_length.

How "token for token" is held.  Beam search compares sums of real numbers a few hundred times per utterance; the reference computes them in fp32, the HIP path through
bf16 GEMMs, and no choice of weights keeps every one of those comparisons wider than bf16 noise (tests/gen_model.py).  So the comparison is exact up to a decision the
REFERENCE'S OWN numbers certify as a near tie:
  1. the device loop's bookkeeping is exact: the candidates the kernel walked, replayed through the pinned CPU loop (oracle/generate_ref.py, which reproduces the fixtures
     exactly on the CPU — tests/test_generate_cpu.py), must give the device's kept hypotheses, scores and order — every setting, no tolerance on tokens;
  2. the device's candidate VALUES are the fp32 oracle's for the same prefixes within TOL, rank by rank;
  3. an utterance's decode is followed candidate by candidate along the reference's trajectory: as long as the ranked candidate lists agree the final hypotheses must be
     the fixture's, token for token; where the lists first differ, the reference's own gap between the swapped candidates must be below 2 TOL (otherwise the test fails),
     and from there on only the weaker statement is checked that the device's best hypothesis scores no worse than the reference's best."""
import numpy as np
import pytest
import torch

import gen_model as GM
from helpers import AED_JCFG, gen_case_inputs
from huggingface_asr_amd import shapes
from oracle import generate_ref as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ENC = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
TOL = 0.06          # bf16 path vs fp32 reference on an accumulated candidate value (sums of <= 13 log-probabilities of magnitude <= ~8)


def _engine(sd, dec_cfg):
    from huggingface_asr_amd.decoder import JointAEDEngine
    eng = JointAEDEngine(ENC, dec_cfg, AED_JCFG, DEV)
    eng.load_state_dict(sd)
    return eng


def certified_decode(eng, sd, enc_cfg, dec_cfg, jcfg, x, am, W, lp, es, ml, eos, ref_q=None, tol=TOL, ctc_weight=0.3, stepwise=False):
    """Steps 1-3 of the module docstring for one generation setting.  Returns (device result, reference-trajectory result (sequences, scores), diverged flags)."""
    from huggingface_asr_amd.decoder import generate
    V, pad, start = dec_cfg["vocab_size"], jcfg["pad_token_id"], jcfg["decoder_start_token_id"]
    B = x.shape[0]
    tr = []
    got = generate(eng, x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=W, max_length=ml, ctc_weight=ctc_weight, length_penalty=lp, early_stopping=es,
                   eos_token_id=eos, trace=tr)
    dev = [(s.cpu().numpy(), i.cpu().numpy().astype(np.int64), d.cpu().numpy().astype(bool)) for s, i, d in tr]
    # the reference's trajectory: the oracle loop on its own candidates (fp32, or the storage model `ref_q` the caller pins it with)
    fn, _ = G.joint_score_fn(sd, enc_cfg, dec_cfg, jcfg, x, am, W, ctc_weight, q=ref_q)
    free = {}
    ref_seq, ref_sc = G.beam_search(fn, B, W, V, max_length=ml, eos=eos, pad=pad, start=start, length_penalty=lp, early_stopping=es, trace=free)
    # 1 + 2: replay the device's candidates through the pinned rules, with the oracle's values for the same prefixes alongside
    fn2, _ = G.joint_score_fn(sd, enc_cfg, dec_cfg, jcfg, x, am, W, ctc_weight, q=ref_q)

    def cand_fn(step, running, open_):
        if step < len(dev):
            s, i, was_done = dev[step]
            assert (open_ == ~was_done).all(), (step, open_, was_done)       # the kernel closes an utterance exactly when the pinned rules freeze it
            return s, i
        return np.zeros((B, 2 * W), np.float32), np.zeros((B, 2 * W), np.int64)            # the device stopped enqueuing: every utterance is closed
    rep = {}
    rep_seq, rep_sc = G.beam_search(fn2, B, W, V, max_length=ml, eos=eos, pad=pad, start=start, length_penalty=lp, early_stopping=es, trace=rep, cand_fn=cand_fn)
    for b in range(B):
        hyps = got[b]["hypotheses"]
        assert len(hyps) == W
        for k, (s, toks) in enumerate(hyps):
            want = rep_seq[b * W + k]
            assert toks == want[: len(toks)].tolist() and (want[len(toks):] == pad).all(), ("bookkeeping", b, k, toks, want)
            assert abs(s - float(rep_sc[b * W + k])) < 1e-6 * max(1.0, abs(s)), ("bookkeeping score", b, k, s, rep_sc[b * W + k])
    worst = 0.0
    for t, (s, i, was_done) in enumerate(dev[: len(rep["acc"])]):        # (run-ahead: the device may have enqueued steps after every utterance was closed)
        acc = rep["acc"][t]                                     # oracle values of EVERY candidate of the device's prefixes (its running scores are the device's)
        for b in range(B):
            if was_done[b] or not rep["open"][t][b]:
                continue
            own = np.sort(acc[b])[::-1][: 2 * W]
            # rank by rank the device's candidate values are the oracle's (a swap of two near-equal candidates leaves the ranked values in place), and the candidates it
            # picked carry the oracle's value for THAT candidate
            d_rank = np.abs(own - s[b]).max()
            d_cand = np.abs(acc[b][i[b]] - s[b]).max()
            worst = max(worst, float(d_rank), float(d_cand))
            assert d_rank < tol and d_cand < tol, ("candidate values", t, b, d_rank, d_cand)
    # 3: follow the reference's trajectory
    diverged = [False] * B
    for t in range(min(len(dev), len(free["cands"]))):
        fv, fi = free["cands"][t]
        s, i, was_done = dev[t]
        for b in range(B):
            if diverged[b] or was_done[b] or not free["open"][t][b]:
                continue
            assert (free["running"][t][b] == rep["running"][t][b]).all()                 # same prefixes so far
            if (fi[b] == i[b]).all():
                continue
            r = int(np.argmax(fi[b] != i[b]))                   # first rank that differs: the reference's own gap between what it has there and what the device has there
            acc_ref = free["acc"][t][b]
            gap = abs(float(acc_ref[fi[b, r]]) - float(acc_ref[i[b, r]]))
            assert gap < 2 * tol, ("decision differs from the reference outside a near tie", t, b, r, gap)
            diverged[b] = True
    for b in range(B):
        dev_h = got[b]["hypotheses"]
        if not diverged[b]:
            for k in range(W):
                want = ref_seq[b * W + k]
                toks = dev_h[k][1]
                assert toks == want[: len(toks)].tolist() and (want[len(toks):] == pad).all(), ("tokens", b, k, toks, want)
                assert abs(dev_h[k][0] - float(ref_sc[b * W + k])) < tol, ("score", b, k, dev_h[k][0], ref_sc[b * W + k])
        else:
            assert dev_h[0][0] >= float(ref_sc[b * W]) - tol, ("after a near tie the best hypothesis is worse than the reference's", b, dev_h[0], ref_sc[b * W])
    return got, (ref_seq, ref_sc), diverged, worst


@pytest.mark.parametrize("name", list(GM.CASES))
def test_hip_generate_against_the_reference_generate(name):
    torch.set_num_threads(8)
    g, sd, x, am, dec_cfg = gen_case_inputs(name)
    eng = _engine(sd, dec_cfg)
    exact = total = 0
    for W, lp, es, ml in GM.SETTINGS:
        key = GM.setting_key(W, lp, es, ml)
        got, (ref_seq, ref_sc), diverged, worst = certified_decode(eng, sd, ENC, dec_cfg, AED_JCFG, x, am, W, lp, es, ml, GM.EOS)
        want = g[key + "/sequences"]
        # the trajectory the certification followed IS the reference's: the fp32 oracle reproduces the fixture (also asserted on the CPU, tests/test_generate_cpu.py)
        if W > 1:
            assert ref_seq.shape == want.shape and (ref_seq == want).all() and np.abs(ref_sc - g[key + "/sequences_scores"]).max() < 1e-5
        for b in range(x.shape[0]):
            total += 1
            if not diverged[b]:
                exact += 1
                for k in range(W):                                  # token for token against the REFERENCE's output
                    row = want[b * W + k] if W > 1 else want[b]
                    toks = got[b]["hypotheses"][k][1]
                    n = len(toks)
                    assert toks == row[:n].tolist() and (row[n:] == GM.PAD).all(), (key, b, k, toks, row)
        if W == 1:
            assert not any(diverged), key                           # greedy margins of the fixture are > 1: nothing to certify away
    print(f"{name}: {exact} of {total} utterance decodes equal the reference's token for token; the rest diverge at a certified near tie")
    assert exact >= total // 2


def test_model_generate_follows_the_do_generate_call_sequence():
    """`do_generate` (general_utils.py:198-218): the trainer's GenerationConfigCustom assigned to the model (train_enc_dec_asr.py:61-85), then
    `model.generate(generation_config=gen_config, **sample)` with `num_return_sequences`, `return_dict_in_generate`, `output_scores` set and the batch's `labels` riding
    along; reads `.sequences` (B * k, L) and `.sequences_scores`.  Against the fixture (rows of utterances that stay on the reference's trajectory, see above) and against
    the engine's own hypotheses; plus the tensor return, `num_return_sequences` < beams, greedy's return type, `max_new_tokens`, and the `prediction_step` form."""
    from test_surface_cpu import _joint_model
    from huggingface_asr_amd.decoding import GenerationConfigCustom
    from huggingface_asr_amd.decoder import generate
    g, sd, x, am, dec_cfg = gen_case_inputs("gen_tiny")
    model = _joint_model(False)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    model = model.to(DEV).eval()
    W, lp, es, ml = 5, 1.0, False, 14
    gen_config = GenerationConfigCustom(bos_token_id=GM.START, pad_token_id=GM.PAD, decoder_start_token_id=GM.START, length_penalty=lp, early_stopping=es,
                                        eos_token_id=GM.EOS, max_length=ml, num_beams=W, ctc_weight=0.3, ctc_margin=0, lm_weight=0, lm_model=None, space_token_id=-1,
                                        apply_eos_space_trick=False, eos_space_trick_weight=1.0)
    model.generation_config = gen_config
    gen_config.num_return_sequences, gen_config.return_dict_in_generate, gen_config.output_scores = W, True, True
    sample = dict(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=torch.tensor([[5, 6, 7], [8, 9, -100]], device=DEV))
    out = model.generate(generation_config=gen_config, **sample)
    ref = generate(model._get_engine(DEV), x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=W, max_length=ml, ctc_weight=0.3, length_penalty=lp, eos_token_id=GM.EOS)
    B = x.shape[0]
    assert out.sequences.shape[0] == B * W and out.sequences_scores.shape == (B * W,) and out.sequences.dtype == torch.long and out.sequences.is_cuda
    L = out.sequences.shape[1]
    assert L == max(len(t) for h in ref for _, t in h["hypotheses"])
    for b in range(B):
        for k in range(W):
            s, toks = ref[b]["hypotheses"][k]
            row = out.sequences[b * W + k].tolist()
            assert row[: len(toks)] == toks and all(v == GM.PAD for v in row[len(toks):])
            assert float(out.sequences_scores[b * W + k]) == pytest.approx(s, abs=1e-6)
        sc = out.sequences_scores[b * W:(b + 1) * W]
        assert bool((sc[:-1] >= sc[1:]).all())
    key = GM.setting_key(W, lp, es, ml)
    want, want_sc = g[key + "/sequences"], g[key + "/sequences_scores"]
    same = [b for b in range(B) if out.sequences[b * W:(b + 1) * W, : want.shape[1]].cpu().numpy().tolist() == want[b * W:(b + 1) * W, :L].tolist()]
    for b in same:
        assert np.abs(out.sequences_scores[b * W:(b + 1) * W].cpu().numpy() - want_sc[b * W:(b + 1) * W]).max() < 0.03
    # fewer returned than beams; the tensor return; the keyword form of Seq2SeqTrainer.prediction_step (do_evaluate passes output_hidden_states=True, general_utils.py:151-154)
    import copy
    g2 = copy.deepcopy(gen_config)
    g2.num_return_sequences, g2.return_dict_in_generate = 2, False
    t2 = model.generate(generation_config=g2, **sample)
    assert isinstance(t2, torch.Tensor) and t2.shape[0] == B * 2
    for b in range(B):
        for k in range(2):
            toks = ref[b]["hypotheses"][k][1]
            assert t2[b * 2 + k, : len(toks)].tolist() == toks
    gen_config.num_return_sequences, gen_config.return_dict_in_generate, gen_config.output_scores = 1, False, False
    t1 = model.generate(**sample, max_length=ml, num_beams=W, output_hidden_states=True)
    for b in range(B):
        toks = ref[b]["tokens"]
        assert t1[b, : len(toks)].tolist() == toks
    # max_new_tokens counts generated tokens: 13 new tokens = max_length 14
    t3 = model.generate(**{k: v for k, v in sample.items()}, max_new_tokens=ml - 1, max_length=None)
    assert torch.equal(t3, t1)
    # greedy: transformers returns GenerateEncoderDecoderOutput without sequence scores
    gg = copy.deepcopy(gen_config)
    gg.num_beams, gg.return_dict_in_generate, gg.output_scores = 1, True, True
    model.generation_config = gg
    og = model.generate(generation_config=gg, **sample)
    assert type(og).__name__ == "GenerateEncoderDecoderOutput" and not hasattr(og, "sequences_scores")
    wantg = g[GM.setting_key(1, 1.0, False, 14) + "/sequences"]
    assert og.sequences.cpu().numpy().tolist() == wantg.tolist()


def test_eos_space_trick_reaches_generate():
    """`apply_eos_space_trick` / `eos_space_trick_weight` (train_enc_dec_asr.py:74-75 -> ctc_scorer.py:333-349) change the processed scores inside the processor; the
    device loop mixes the scores in its own kernel, so a request with the trick decodes through the host loop, which calls the processor.  Pinned against the oracle with
    the same trick on the same model; the fixture `ctc_prefix.npz` case "b" pins the processor's arithmetic against the reference's."""
    from huggingface_asr_amd.decoder import generate, generate_stepwise
    from helpers import oracle_generate
    g, sd, x, am, dec_cfg = gen_case_inputs("gen_tiny")
    eng = _engine(sd, dec_cfg)
    fl = am.sum(-1).to(DEV, torch.int32)
    plain = generate(eng, x.to(DEV), fl, num_beams=3, max_length=10, ctc_weight=0.3, eos_token_id=GM.EOS)
    # a space token the CTC scorer favours at some step where the decoder favours EOS does not exist by construction: take the trick's weight to an extreme and any token
    space = plain[0]["tokens"][2]
    a = generate(eng, x.to(DEV), fl, num_beams=3, max_length=10, ctc_weight=0.3, eos_token_id=GM.EOS, space_token_id=space, apply_eos_space_trick=True, eos_space_trick_weight=0.5)
    b = generate_stepwise(eng, x.to(DEV), fl, num_beams=3, max_length=10, ctc_weight=0.3, eos_token_id=GM.EOS, space_token_id=space, apply_eos_space_trick=True,
                          eos_space_trick_weight=0.5)
    assert [h["hypotheses"] for h in a] == [h["hypotheses"] for h in b]                 # generate() routed the request to the host loop
    from oracle import aed_ref as A
    fn, B = G.joint_score_fn(sd, ENC, dec_cfg, AED_JCFG, x, am, 3, 0.3, q=A.E.bf16_round, eos_space=(GM.EOS, space, 0.5))
    seq, sc = G.beam_search(fn, B, 3, GM.V, max_length=10, eos=GM.EOS, pad=GM.PAD, start=GM.START)
    for u in range(B):
        assert abs(a[u]["score"] - float(sc[u * 3])) < 0.05
