"""GPU parity: HIP CTC prefix scorer / logits processors against the REFERENCE's own outputs (tests/golden/ctc_prefix.npz,
4 decoding steps incl. the beam-0 state-selection quirk and the eos/space trick) and against the oracle at a larger size."""
import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(params=["all_chains_kept", "selected_chains_rerun"])
def state_form(request, monkeypatch):
    """both ways the processor carries its state between tokens: every (hypothesis, token) chain kept (one scan per token, mi_ctc_prefix_score_full) and — what it falls back to
    when that tensor would pass FULL_STATE_BYTES — only the current prefixes' variables, the selected chains re-run (mi_ctc_prefix_advance)"""
    from huggingface_asr_amd.decoding import CTCRescorerLogitsProcessor
    if request.param == "selected_chains_rerun":
        monkeypatch.setattr(CTCRescorerLogitsProcessor, "FULL_STATE_BYTES", 0)
    return request.param


@pytest.mark.parametrize("case", ["a", "b", "c", "m"])
def test_ctc_rescorer_matches_reference(case, state_form):
    from huggingface_asr_amd.decoding import CTCRescorerLogitsProcessor, LogSoftmaxProcessor
    g = load_golden("ctc_prefix")
    B, W, T, O, blank, eos, space, trick, margin = [int(v) for v in g[f"{case}/meta"]]          # case "m": ctc_margin 6 (a no-op in the reference, and here)
    proc = CTCRescorerLogitsProcessor(torch.from_numpy(g[f"{case}/enc_logits"]).to(DEV), torch.from_numpy(g[f"{case}/lens"]).to(DEV),
                                      blank, eos, margin, 0.3, W, space, bool(trick), 0.8)
    for step in range(4):
        ids = torch.from_numpy(g[f"{case}/step{step}/input_ids"]).to(DEV)
        att = torch.from_numpy(g[f"{case}/step{step}/att"]).to(DEV)
        out = proc(ids, att.clone())
        want = g[f"{case}/step{step}/out"]
        got = out.cpu().numpy()
        live = want > -1e9
        np.testing.assert_array_equal(got > -1e9, live)
        np.testing.assert_allclose(got[live], want[live], atol=3e-4, rtol=1e-5)
    lsm = LogSoftmaxProcessor()(ids, att.clone())
    np.testing.assert_allclose(lsm.cpu().numpy(), g[f"{case}/logsoftmax"], atol=1e-5, rtol=0)


def test_ctc_prefix_full_size_vs_oracle(state_form):
    """BASELINE config 5 shape: T'=250, V+1=5001, B=1, W=5 — three steps against the oracle."""
    from huggingface_asr_amd.decoding import CTCRescorerLogitsProcessor
    from oracle import ctc_prefix_ref as P
    B, W, T, O = 1, 5, 250, 5001
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(B, T, O, generator=g) * 3.0
    lens = torch.tensor([248])
    blank = O - 1
    proc = CTCRescorerLogitsProcessor(enc.to(DEV), lens.to(DEV), blank, 1, 0, 0.3, W, 5, False, 1.0)
    ref = P.PrefixScorer(torch.log_softmax(enc, -1).numpy(), lens.numpy(), blank, W)
    ids = torch.full((B * W, 1), 2, dtype=torch.long)
    for step in range(3):
        want = ref.step(ids.numpy())
        got = proc.ctc_scores(ids.to(DEV)).cpu().numpy()
        live = want > -1e9
        np.testing.assert_array_equal(got > -1e9, live)
        np.testing.assert_allclose(got[live], want[live], atol=2e-3, rtol=2e-5)
        nxt = torch.from_numpy(np.where(live, want, -np.inf)).topk(W, dim=1).indices
        ids = torch.cat([ids, torch.stack([nxt[i, i % W] for i in range(B * W)])[:, None]], 1)


def test_processor_rejects_cpu_tensors():
    """no CPU fallback; a ctc_margin > 0 is accepted (a no-op, as in the reference: fixture case "m" above)"""
    from huggingface_asr_amd.decoding import CTCRescorerLogitsProcessor
    enc = torch.randn(1, 10, 7)
    with pytest.raises(RuntimeError):
        CTCRescorerLogitsProcessor(enc, torch.tensor([10]), 6, 1, 0, 0.3, 2, 5, False, 1.0)
    assert CTCRescorerLogitsProcessor(enc.to(DEV), torch.tensor([10]), 6, 1, 3, 0.3, 2, 5, False, 1.0).ctc_margin == 3


def test_both_state_forms_give_the_same_bits(monkeypatch):
    """keeping every chain or re-running the selected ones is the same arithmetic: identical scores over a 6-token decode with beams that swap"""
    from huggingface_asr_amd.decoding import CTCRescorerLogitsProcessor
    B, W, T, O = 2, 3, 60, 301
    g = torch.Generator().manual_seed(9)
    enc = (torch.randn(B, T, O, generator=g) * 3.0).to(DEV)
    lens = torch.tensor([60, 47]).to(DEV)
    outs = []
    for limit in (1 << 30, 0):
        monkeypatch.setattr(CTCRescorerLogitsProcessor, "FULL_STATE_BYTES", limit)
        proc = CTCRescorerLogitsProcessor(enc, lens, O - 1, 1, 0, 0.3, W, 5, False, 1.0)
        gg = torch.Generator().manual_seed(10)
        ids = torch.full((B * W, 1), 2, dtype=torch.long)
        got = []
        for step in range(6):
            got.append(proc.ctc_scores(ids.to(DEV)).clone())
            perm = torch.cat([torch.randperm(W, generator=gg) + b * W for b in range(B)])          # beams re-ordered, then extended
            ids = torch.cat([ids[perm], torch.randint(3, O - 1, (B * W, 1), generator=gg)], 1)
        outs.append(torch.stack(got))
    assert torch.equal(outs[0], outs[1])
