"""GPU parity, BASELINE config 5 at its REAL size: DeCRED_base-shaped joint model — E-Branchformer-base encoder + 8 x 512 GPT-2 decoder (8 heads of
64, fixed positions, auxiliary head at layer 5, V = 5001; hub `Lakoc/gpt2_512h_8l_add_head6_04`, hf_shared_models/DeCRED_base.py:20-22) — bs = 1, one 10 s clip.
With at most 8 rows in flight (1 x beams) the token step runs the fused skinny linears (`v_dot2c` on a bf16 LDS image, K = 512 / 2048): this is the only place they
meet the oracle at that size.  Reference: src/decoding/ctc_scorer.py:58-207, src/models/decoders/multi_head_gpt2.py:80-170."""
import numpy as np
import pytest
import torch

import config5_model as M
from helpers import oracle_generate
from oracle import aed_ref as A
from oracle import fbank_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs():
    w = M.clip()[0]
    f = fbank_ref.extract(w)
    x = torch.from_numpy(np.pad(f, ((0, 1000 - f.shape[0]), (0, 0))))[None].float()
    am = torch.zeros(1, 1000, dtype=torch.long); am[:, : f.shape[0]] = 1
    return x, am


def _engine(sd):
    from huggingface_asr_amd.decoder import JointAEDEngine
    eng = JointAEDEngine(M.ENC_CFG, M.DEC_CFG, M.JCFG, DEV)
    eng.load_state_dict(sd)
    return eng


@pytest.mark.parametrize("W", [1, 5])
def test_token_step_logits_at_decred_base_size(W):
    """The first 7 token steps, W rows with W different prefixes (what beam search feeds the step): the engine's KV-cache step logits against the oracle's teacher-forced
    decoder on the same rows (bf16 storage model), seeded random weights."""
    torch.set_num_threads(8)
    sd = M.state_dict(0, structured=False)
    x, am = _inputs()
    eng = _engine(sd)
    enc_out, enc_bf, T2, key_len = eng.encode(x.to(DEV), am.sum(-1).to(DEV, torch.int32))
    assert (T2, enc_bf.shape[1]) == (250, 512)
    ids = torch.from_numpy(np.stack([np.concatenate([[2], 7 + (np.arange(7) * (37 + 11 * r) + 101 * r) % 4990]) for r in range(W)])).long()     # (W, 8): start + 7 tokens
    q = A.E.bf16_round
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    with torch.no_grad():
        hidden = A.E.encoder_forward(esd, M.ENC_CFG, x, am, q)
        outer = A.E.conv_out_lengths_outer(am.sum(-1), M.ENC_CFG).long()
        mask = (torch.arange(250)[None] < outer[:, None]).repeat_interleave(W, 0)
        _, want = A.decoder_forward(sd, "decoder.", M.DEC_CFG, ids[:, :7], hidden.repeat_interleave(W, 0), mask, None, q)      # (W, 7, V)
    d = enc_bf.shape[1]
    kvs = eng.dec.cross_kv(enc_bf.view(1, T2, d).repeat_interleave(W, 0).reshape(W * T2, d))
    cache = eng.dec.init_cache(W, 16)
    key_rep = key_len.repeat_interleave(W) if key_len is not None else None
    got = torch.stack([eng.dec.step(ids[:, u:u + 1].to(DEV), cache, kvs, T2, key_rep) for u in range(7)], 1).float().cpu()          # (W, 7, V): <= 8 rows -> skinny path
    std = float(want.std())
    err = (got - want).abs()
    assert float(err.max()) < 0.06 * max(std, 1.0) + 0.03 and float(err.mean()) < 0.01 * max(std, 1.0), (float(err.max()), float(err.mean()), std)
    # the argmax agrees wherever the oracle's own top-2 margin is above the noise
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.1
    assert bool((got.argmax(-1)[clear] == want.argmax(-1)[clear]).all()) and int(clear.sum()) >= 3 * W


def test_greedy_and_beam_hypotheses_token_for_token():
    """Joint CTC / attention decoding (ctc_weight 0.3) on a decoder whose logits are well separated by construction (config5_model.state_dict(structured=True): every token
    has six designated successors at distinct logit levels; the transformer blocks stay random, so the KV cache and its beam re-ordering still shape the scores).  The
    oracle's own decision margins — smallest gap among the top W + 1 candidates at every step — are checked to sit far above bf16 noise; then greedy AND every kept beam
    hypothesis must equal the oracle's token for token, with the same length-normalised scores."""
    from huggingface_asr_amd.decoder import generate
    torch.set_num_threads(8)
    sd = M.state_dict(1, structured=True)
    x, am = _inputs()
    eng = _engine(sd)
    for W, maxlen in ((1, 8), (3, 7)):
        mg = []
        want = oracle_generate(sd, M.ENC_CFG, M.DEC_CFG, M.JCFG, x, am, W, maxlen, 0.3, margins=mg)[0]
        assert min(m[0] for m in mg) > 0.15, mg                          # the fixture IS well separated (bf16 noise on these scores: ~0.03)
        got = generate(eng, x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=W, max_length=maxlen, ctc_weight=0.3, eos_token_id=1)[0]
        assert got["tokens"] == want[1], (W, got["tokens"], want[1])
        assert abs(got["score"] - want[0]) < 0.03, (got["score"], want[0])
        assert len(got["hypotheses"]) == len(want[2]) == W
        for (gs, gt), (ws, wt) in zip(got["hypotheses"], want[2]):
            assert gt == wt and abs(gs - ws) < 0.03, (W, gt, wt, gs, ws)


def test_beam5_hypotheses_token_for_token_at_the_reference_default_width():
    """The reference decodes DeCRED_base with `num_beams=5` (hf_shared_models/DeCRED_base.py:20-22); W = 5 keeps the top 10 of 5 x 5001 candidates per step and the beam
    re-ordering touches every KV cache row.  Structured weights of seed 8: with five beams the six successor levels of the fixture combine into near-ties for most seeds
    (seed 1, the W <= 3 fixture above: 0.011 at step 3); seed 8 keeps every decision among the top W + 1 candidates >= 0.13 apart, 4x the bf16 noise on these scores —
    the oracle's own margins are asserted first, then all five kept hypotheses and their scores must equal the oracle's."""
    from huggingface_asr_amd.decoder import generate, generate_stepwise
    torch.set_num_threads(8)
    sd = M.state_dict(8, structured=True)
    x, am = _inputs()
    eng = _engine(sd)
    W, maxlen = 5, 6
    mg = []
    want = oracle_generate(sd, M.ENC_CFG, M.DEC_CFG, M.JCFG, x, am, W, maxlen, 0.3, margins=mg)[0]
    assert min(m[0] for m in mg) > 0.1, mg
    for fn in (generate, generate_stepwise):                  # the device-resident loop and the host loop
        got = fn(eng, x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=W, max_length=maxlen, ctc_weight=0.3, eos_token_id=1)[0]
        assert got["tokens"] == want[1], (got["tokens"], want[1])
        assert abs(got["score"] - want[0]) < 0.03, (got["score"], want[0])
        assert len(got["hypotheses"]) == len(want[2]) == W
        for (gs, gt), (ws, wt) in zip(got["hypotheses"], want[2]):
            assert gt == wt and abs(gs - ws) < 0.03, (gt, wt, gs, ws)


def test_beam5_over_forty_tokens_at_decred_base_size():
    """the reference's defaults are `max_length=200`, `num_beams=5` (hf_shared_models/DeCRED_base.py:20-22) and the bench decodes 40 tokens; the cases
    above stop at 6-8.  Here: W = 5, max_length 44 (43 token steps: the KV cache is re-ordered 43 times, the CTC prefix scorer walks 43 prefixes of every beam), with an
    end-of-sequence id chosen among the structured decoder's second-best successors so that hypotheses close along the way and the rest closes at max_length.  Held by
    the certification of tests/test_gpu_generate.py against the oracle with the kernels' bf16 storage model: the kernel's bookkeeping exact over all 43 steps, every
    candidate value within tolerance of the oracle's for the same prefixes, token-for-token equality of all five kept hypotheses up to a certified near tie."""
    from test_gpu_generate import certified_decode
    from huggingface_asr_amd.decoder import generate
    torch.set_num_threads(8)
    sd = M.state_dict(8, structured=True)
    x, am = _inputs()
    eng = _engine(sd)
    fl = am.sum(-1).to(DEV, torch.int32)
    greedy = generate(eng, x.to(DEV), fl, num_beams=1, max_length=44, ctc_weight=0.3, eos_token_id=10 ** 6)[0]["tokens"]
    assert len(greedy) == 44
    eos = M.successors(greedy[18])[1]                      # the second-best continuation after the 18th token ends a hypothesis
    got, (ref_seq, ref_sc), diverged, worst = certified_decode(eng, sd, M.ENC_CFG, M.DEC_CFG, M.JCFG, x, am, 5, 1.0, False, 44, eos, ref_q=A.E.bf16_round, tol=0.06)
    lens = [len(t) for _, t in got[0]["hypotheses"]]
    print("config 5, W = 5, 43 steps: kept lengths", lens, "worst candidate-value gap", worst, "diverged at a near tie:", diverged)
    assert len(lens) == 5 and max(lens) >= 40


@pytest.mark.parametrize("W", [1, 5, 8])
def test_fused_token_step_against_the_launch_per_op_step(W):
    """Round 5: the token step with one new token per row runs three launches per layer (csrc/decoder_fused.hip: the output projections leave per-head / per-slice partial
    rows that the next launch's prologue sums) instead of eight.  Same rounding points, another order of the fp32 sums: over 20 steps at DeCRED_base size with W different
    prefixes (and a beam re-ordering of the caches half way) its logits must stay within fp32-sum noise of the launch-per-op step's (mi_gpt2_config.step_form = 1) — far
    inside the bf16 noise the oracle comparison above allows — and the K / V caches it appends must be the same bf16 values up to last-bit flips."""
    sd = M.state_dict(0, structured=False)
    x, am = _inputs()
    eng = _engine(sd)
    enc_out, enc_bf, T2, key_len = eng.encode(x.to(DEV), am.sum(-1).to(DEV, torch.int32))
    d = enc_bf.shape[1]
    kvs = eng.dec.cross_kv(enc_bf.view(1, T2, d).repeat_interleave(W, 0).reshape(W * T2, d))
    key_rep = key_len.repeat_interleave(W) if key_len is not None else None
    ids = torch.from_numpy(np.stack([np.concatenate([[2], 7 + (np.arange(20) * (37 + 11 * r) + 101 * r) % 4990]) for r in range(W)])).long().to(DEV)
    ca, cb = eng.dec.init_cache(W, 32), eng.dec.init_cache(W, 32)
    perm = torch.tensor([(r * 3 + 1) % W for r in range(W)], device=DEV)
    worst = 0.0
    for u in range(20):
        eng.dec._gcfg.step_form = 0
        a = eng.dec.step(ids[:, u:u + 1], ca, kvs, T2, key_rep)
        eng.dec._gcfg.step_form = 1
        b = eng.dec.step(ids[:, u:u + 1], cb, kvs, T2, key_rep)
        eng.dec._gcfg.step_form = 0
        assert torch.isfinite(a).all()
        worst = max(worst, float((a - b).abs().max()))
        assert float((a - b).abs().max()) < 2e-2 and float((a - b).abs().mean()) < 2e-3, (u, float((a - b).abs().max()), float((a - b).abs().mean()), float(b.std()))
        if u == 9:
            eng.dec.reorder_cache(ca, perm); eng.dec.reorder_cache(cb, perm)
    for l in range(len(ca["k"])):
        for t in ("k", "v"):
            fa, fb = ca[t][l][:, :20].float(), cb[t][l][:, :20].float()
            assert float((fa - fb).abs().max()) <= 0.02 * float(fb.abs().max()) and float((fa - fb).abs().mean()) < 1e-3 * float(fb.abs().max())     # last-bit flips of bf16 values (deeper layers: more of them)
    print(f"fused vs launch-per-op token step, W = {W}: max |dlogit| over 20 steps {worst:.2e}")
