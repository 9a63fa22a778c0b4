"""GPU parity: joint CTC/attention encoder-decoder forward (HIP encoder + HIP GPT-2 cross-attention decoder) against the
REFERENCE's JointCTCAttentionEncoderDecoder.forward (tests/golden/aed_*.npz: three losses, decoder and encoder logits) and
the oracle with the kernels' bf16 storage model; KV-cache stepping against the teacher-forced pass."""
import numpy as np
import pytest
import torch

from helpers import AED_JCFG, TINY_DEC, aed_case_inputs, load_golden
from huggingface_asr_amd import shapes
from oracle import aed_ref as A

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _enc_cfg():
    return dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")


@pytest.mark.parametrize("name,fixed", [("aed_tiny", False), ("aed_tiny_fixedpos", True)])
def test_joint_forward_vs_reference(name, fixed):
    from huggingface_asr_amd.decoder import JointAEDEngine
    g = load_golden(name)
    sd, x, am, lab = aed_case_inputs(g)
    dec_cfg = dict(TINY_DEC, pos_emb_fixed=fixed)
    eng = JointAEDEngine(_enc_cfg(), dec_cfg, AED_JCFG, DEV)
    eng.load_state_dict(sd)
    out = eng.forward(x.to(DEV), am.sum(-1).to(DEV, torch.int32), lab.to(DEV))
    torch.cuda.synchronize()
    dl = np.abs(out["logits"].cpu().numpy() - g["logits"])
    de = np.abs(out["encoder_logits"].float().cpu().numpy() - g["encoder_logits"])
    assert de.max() < 0.06 and de.mean() < 0.009, (de.max(), de.mean())
    assert dl.max() < 0.08 and dl.mean() < 0.012, (dl.max(), dl.mean())
    for k in ("loss", "enc_loss", "dec_loss"):
        assert abs(float(out[k]) - float(g[k])) < 2e-3 * abs(float(g[k])), (k, float(out[k]), float(g[k]))
    # tighter: oracle with the bf16 storage model
    with torch.no_grad():
        oq = A.joint_forward(sd, _enc_cfg(), dec_cfg, AED_JCFG, x, am, lab, q=A.E.bf16_round)
    dq = np.abs(out["logits"].cpu().numpy() - oq["logits"].numpy())
    assert dq.max() < 0.04 and dq.mean() < 0.004, (dq.max(), dq.mean())


def test_kv_cache_steps_match_teacher_forcing():
    """Incremental decoding with the KV cache reproduces the teacher-forced logits position by position."""
    from huggingface_asr_amd.decoder import JointAEDEngine, shift_tokens_right
    g = load_golden("aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    eng = JointAEDEngine(_enc_cfg(), dict(TINY_DEC), AED_JCFG, DEV)
    eng.load_state_dict(sd)
    enc_out, enc_bf, T2, key_len = eng.encode(x.to(DEV), am.sum(-1).to(DEV, torch.int32))
    ids = shift_tokens_right(lab, 50, 2).to(DEV)
    full = eng.dec.forward(ids, enc_bf, T2, key_len)["logits"]
    kvs = eng.dec.cross_kv(enc_bf)
    cache = eng.dec.init_cache(ids.shape[0], 16)
    got = []
    got.append(eng.dec.step(ids[:, :3], cache, kvs, T2, key_len))         # prompt of 3 tokens at once
    for u in range(3, ids.shape[1]):
        got.append(eng.dec.step(ids[:, u:u + 1], cache, kvs, T2, key_len))
    want = torch.stack([full[:, 2]] + [full[:, u] for u in range(3, ids.shape[1])], 0)
    torch.testing.assert_close(torch.stack(got, 0), want, atol=3e-2, rtol=0)
    # beam reorder keeps rows consistent
    eng.dec.reorder_cache(cache, torch.tensor([1, 0], device=DEV))
    assert cache["k"][0].shape[0] == 2


from helpers import oracle_generate as _oracle_generate  # noqa: E402


@pytest.mark.parametrize("W", [1, 3])
def test_joint_decoding_matches_oracle(W):
    from huggingface_asr_amd.decoder import JointAEDEngine, generate
    g = load_golden("aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    dec_cfg = dict(TINY_DEC)
    eng = JointAEDEngine(_enc_cfg(), dec_cfg, AED_JCFG, DEV)
    eng.load_state_dict(sd)
    got = generate(eng, x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=W, max_length=8, ctc_weight=0.3)
    want = _oracle_generate(sd, _enc_cfg(), dec_cfg, AED_JCFG, x, am, W, 8, 0.3)
    for b in range(2):
        assert abs(got[b]["score"] - want[b][0]) < 0.05 * max(1.0, abs(want[b][0])), (got[b], want[b])
        if W == 1 or got[b]["tokens"] == want[b][1]:
            assert got[b]["tokens"] == want[b][1], (got[b], want[b])
            assert len(got[b]["hypotheses"]) == W
        else:
            # beam search on a random-weight model is a sequence of near-ties: a bf16-level difference in one logit changes which beams survive a pruning
            # step, and the fp32 oracle and the bf16 engine may then end on different hypotheses of (almost) equal score.  A different hypothesis is accepted
            # only if it is at least as good as the oracle's best by the engine's own scoring (greedy, W = 1, stays an exact match).
            assert got[b]["score"] >= want[b][0] - 0.01 * abs(want[b][0]), (got[b], want[b])


def test_hf_joint_model_surface():
    """AutoModelForSpeechSeq2Seq route: reference state dict in, Seq2SeqLMOutputLosses out, generate() tokens."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_surface_cpu import _joint_model
    from huggingface_asr_amd.decoder import generate
    g = load_golden("aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    model = _joint_model(False)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    model = model.to(DEV).eval()
    with torch.no_grad():
        out = model(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    for k in ("loss", "enc_loss", "dec_loss"):
        assert abs(float(getattr(out, k)) - float(g[k])) < 2e-3 * abs(float(g[k])), k
    assert np.abs(out.logits.cpu().numpy() - g["logits"]).max() < 0.08
    assert out.encoder_logits.shape == (2, 50, 51) and out.encoder_last_hidden_state.shape == (2, 50, 128)
    from huggingface_asr_amd.decoding import GenerationConfigCustom
    model.generation_config = GenerationConfigCustom(pad_token_id=50, eos_token_id=1, decoder_start_token_id=2, bos_token_id=2, num_beams=3, max_length=8, ctc_weight=0.3,
                                                     length_penalty=1.0, early_stopping=False)           # what train_enc_dec_asr.py:61-85 assigns
    toks = model.generate(input_values=x.to(DEV), attention_mask=am.to(DEV))
    ref = generate(model._get_engine(DEV), x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=3, max_length=8, ctc_weight=0.3)
    assert toks.shape[0] == 2
    for b in range(2):
        assert toks[b, : len(ref[b]["tokens"])].tolist() == ref[b]["tokens"] and (toks[b, len(ref[b]["tokens"]):] == 50).all()


def test_c_step_driver_matches_python_step_and_reorders_cache():
    """mi_gpt2_step (whole token step in one C call) vs the op-by-op Python step: bit for bit on the MFMA path (more than 8 rows: same
    kernels, same order), within bf16 noise on the fused skinny path (<= 8 rows: LayerNorm / bias / gelu_new / residual fused into GEMV-style
    linears, different summation order); the one-kernel beam re-ordering of all KV caches == index_select per tensor."""
    from huggingface_asr_amd.decoder import JointAEDEngine, shift_tokens_right
    g = load_golden("aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    eng = JointAEDEngine(_enc_cfg(), dict(TINY_DEC), AED_JCFG, DEV)
    eng.load_state_dict(sd)
    enc_out, enc_bf, T2, key_len = eng.encode(x.to(DEV), am.sum(-1).to(DEV, torch.int32))
    ids = shift_tokens_right(lab, 50, 2).to(DEV)
    kvs = eng.dec.cross_kv(enc_bf)
    ca, cb = eng.dec.init_cache(ids.shape[0], 16), eng.dec.init_cache(ids.shape[0], 16)
    for lo, hi in ((0, 5), (5, 6), (6, 7)):
        a = eng.dec.step(ids[:, lo:hi], ca, kvs, T2, key_len)
        b = eng.dec.step_py(ids[:, lo:hi], cb, kvs, T2, key_len)
        if ids.shape[0] * (hi - lo) > 8:
            assert torch.equal(a, b), (lo, hi, float((a - b).abs().max()))
        else:
            torch.testing.assert_close(a, b, atol=3e-2, rtol=0)
    for l in range(len(ca["k"])):
        assert torch.equal(ca["k"][l][:, :5], cb["k"][l][:, :5]) and torch.equal(ca["v"][l][:, :5], cb["v"][l][:, :5])
        torch.testing.assert_close(ca["k"][l][:, :7].float(), cb["k"][l][:, :7].float(), atol=3e-2, rtol=0)
    perm = torch.tensor([1, 1], device=DEV)
    want_k = [t.index_select(0, perm)[:, :7].clone() for t in ca["k"]]
    want_v = [t.index_select(0, perm)[:, :7].clone() for t in ca["v"]]
    eng.dec.reorder_cache(ca, perm)
    for l in range(len(want_k)):
        assert torch.equal(ca["k"][l][:, :7], want_k[l]) and torch.equal(ca["v"][l][:, :7], want_v[l])
    # decoding continues on the re-ordered cache
    a = eng.dec.step(ids[:, 7:8], ca, kvs, T2, key_len)
    assert torch.isfinite(a).all()


@pytest.mark.parametrize("B,W,V,with_ctc,lp,es", [(3, 4, 50, True, 1.0, False), (2, 1, 37, True, 1.0, False), (2, 5, 5001, False, 0.7, False), (1, 16, 300, True, 1.3, "never"),
                                                    (3, 3, 50, True, 1.0, True), (2, 5, 64, True, 1.5, False)])
def test_beam_step_kernel_follows_the_pinned_loop(B, W, V, with_ctc, lp, es):
    """csrc/beam_step.hip against oracle/generate_ref.py `beam_search` (the loop pinned by tests/golden/gen_*.npz against the reference's own generate()) on random
    scores with a strong EOS from the second step on (hypotheses close at most steps, utterances finish at different steps, the rest closes at max_length): both are fed
    the same per-step scores; candidates, kept hypotheses, their scores and order must agree exactly over a whole decode."""
    from huggingface_asr_amd import _lib, ops
    from huggingface_asr_amd.decoder import _ES_MODE, _step_denoms
    from oracle import generate_ref as G
    gen = torch.Generator().manual_seed(B * 1000 + W * 10 + V)
    pad, eos, max_length = V - 1, 1, 11
    steps, Lmax, w = max_length - 1, max_length + 1, 0.3
    n = B * W
    ids = torch.full((n, Lmax), pad, dtype=torch.long); ids[:, 0] = 2
    bs = torch.zeros(B, W); bs[:, 1:] = -1e9
    d_ids, d_bs = ids.to(DEV), bs.view(-1).contiguous().to(DEV)
    d_done, d_nfin = torch.zeros(B, dtype=torch.int32, device=DEV), torch.zeros(B, dtype=torch.int32, device=DEV)
    d_fs, d_fl = torch.zeros(B, W, dtype=torch.float32, device=DEV), torch.zeros(B, W, dtype=torch.int32, device=DEV)
    d_ft = torch.full((B, W, Lmax), pad, dtype=torch.long, device=DEV)
    processed, tops = [], []
    for t in range(steps):
        cur = t + 1
        Vp = (V + 7) // 8 * 8
        buf = torch.randn(n, Vp, generator=gen) * 2.0
        buf[:, eos] += 3.0 if t >= 1 else -5.0                     # EOS near the top from the second step on
        logits = buf.to(DEV)[:, :V]
        lse = ops.row_lse(logits)
        ctc = (torch.randn(n, V, generator=gen) * 3.0 - 5.0) if with_ctc else None
        d_ctc = ctc.to(DEV) if with_ctc else None
        sc = (logits.cpu() - lse.cpu()[:, None]).numpy()           # the kernel's arithmetic on the host, one rounding per operation
        if with_ctc:
            sc[:, pad] = np.float32(-10000000000.0)
            sc = np.float32(1 - w) * sc + np.float32(w) * ctc.numpy()
        processed.append(sc.astype(np.float32))
        new_tok, beam_idx = torch.empty(n, dtype=torch.long, device=DEV), torch.empty(n, dtype=torch.long, device=DEV)
        top_s, top_i = torch.empty(B, 2 * W, device=DEV), torch.empty(B, 2 * W, dtype=torch.int32, device=DEV)
        denom, heur = _step_denoms(cur, max_length, lp, es)
        was_done = d_done.cpu().bool().tolist()
        _lib.check(_lib.lib().mi_beam_step(logits.data_ptr(), logits.stride(0), lse.data_ptr(), d_ctc.data_ptr() if with_ctc else None, float(1 - w), float(w), int(with_ctc), pad, eos,
                                           B, W, V, cur, max_length, Lmax, denom, heur, _ES_MODE[es], d_ids.data_ptr(), d_bs.data_ptr(), new_tok.data_ptr(), beam_idx.data_ptr(),
                                           d_done.data_ptr(), d_nfin.data_ptr(), d_fs.data_ptr(), d_fl.data_ptr(), d_ft.data_ptr(), top_s.data_ptr(), top_i.data_ptr(), None,
                                           torch.cuda.current_stream().cuda_stream), "mi_beam_step")
        tops.append((top_s.cpu().numpy(), top_i.cpu().numpy(), was_done))
    calls = []

    def score_fn(rows):                                            # the oracle loop sees the same processed scores, step by step
        calls.append(rows.copy())
        return processed[len(calls) - 1]
    tr = {}
    seq, scores = G.beam_search(score_fn, B, W, V, max_length=max_length, eos=eos, pad=pad, start=2, length_penalty=lp, early_stopping=es, trace=tr)
    fs, fl, ft, nf = d_fs.cpu().numpy(), d_fl.cpu().numpy(), d_ft.cpu().numpy(), d_nfin.cpu().numpy()
    assert (nf == W).all() and bool(d_done.cpu().all())            # every utterance ends with W kept hypotheses (max_length closes the rest)
    for b in range(B):
        for k in range(W):
            want = seq[b * W + k]
            n_tok = int(fl[b, k])
            assert ft[b, k, :n_tok].tolist() == want[:n_tok].tolist() and (want[n_tok:] == pad).all(), (b, k, ft[b, k], want)
            assert fs[b, k] == scores[b * W + k], (b, k, fs[b, k], scores[b * W + k])
    for ts, ti, was_done in tops:                                  # candidates come best first
        for b in range(B):
            if not was_done[b]:
                assert bool((ts[b, :-1] >= ts[b, 1:]).all())
    ends = [int(ft[b, k, int(fl[b, k]) - 1]) for b in range(B) for k in range(W)]
    assert eos in ends                                             # the case closes hypotheses on the end-of-sequence token ...
    assert len(calls) >= 3                                         # ... and not all of them at once


@pytest.mark.parametrize("W,ctc_weight", [(1, 0.3), (3, 0.3), (5, 0.3), (3, 0.0)])
def test_device_resident_decoding_equals_the_stepwise_loop(W, ctc_weight):
    """decoder.generate (beam bookkeeping on the device, CTC prefix scorer on a second stream, no host copy per token) against decoder.generate_stepwise (the host loop)
    on the reference-pinned tiny AED model: same hypotheses, same scores, same order — also when an EOS chosen from the greedy path closes hypotheses and ends decoding early."""
    from huggingface_asr_amd.decoder import JointAEDEngine, generate, generate_stepwise
    g = load_golden("aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    eng = JointAEDEngine(_enc_cfg(), dict(TINY_DEC), AED_JCFG, DEV)
    eng.load_state_dict(sd)
    fl = am.sum(-1).to(DEV, torch.int32)
    greedy = generate_stepwise(eng, x.to(DEV), fl, num_beams=1, max_length=12, ctc_weight=ctc_weight, eos_token_id=10 ** 6)
    for eos in (10 ** 6, greedy[0]["tokens"][3], greedy[1]["tokens"][5]):                # never / early for utterance 0 / later
        want = generate_stepwise(eng, x.to(DEV), fl, num_beams=W, max_length=12, ctc_weight=ctc_weight, eos_token_id=eos)
        for ahead in (1, 2, 4):
            got = generate(eng, x.to(DEV), fl, num_beams=W, max_length=12, ctc_weight=ctc_weight, eos_token_id=eos, run_ahead=ahead)
            for b in range(2):
                assert got[b]["tokens"] == want[b]["tokens"] and got[b]["score"] == want[b]["score"], (eos, ahead, b, got[b], want[b])
                assert got[b]["hypotheses"] == want[b]["hypotheses"], (eos, ahead, b)
