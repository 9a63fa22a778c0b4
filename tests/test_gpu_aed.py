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
    toks = model.generate(input_values=x.to(DEV), attention_mask=am.to(DEV), num_beams=3, max_length=8, ctc_weight=0.3)
    ref = generate(model._get_engine(DEV), x.to(DEV), am.sum(-1).to(DEV, torch.int32), num_beams=3, max_length=8, ctc_weight=0.3)
    for b in range(2):
        assert toks[b, : len(ref[b]["tokens"])].tolist() == ref[b]["tokens"]


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
