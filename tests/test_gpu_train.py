"""GPU parity of the training step: loss + EVERY parameter gradient of the HIP trainer against
  (a) the golden gradients of the imported reference model (tests/golden/grads_*.npz, made by make_golden.py from
      Wav2Vec2EBranchformerForCTC in train() mode with dropouts 0, fp32), and
  (b) torch autograd of the CPU oracle on other configurations (head size 64 -> fused forward attention + recomputed probabilities).

Tolerance: the HIP path computes activations AND activation gradients in bf16 (the reference's autocast recipe), the fixtures
are fp32: per tensor,  |g - g_ref|_2 <= 3 % of |g_ref|_2  and cosine >= 0.999.  The yard-stick is stored in the fixtures: the reference's own
bf16-autocast backward differs from its fp32 backward by 1.7-2.0 % (mean over tensors, `bf16_grad_relerr_mean`); measured here: <= 1.4 % on
every tensor.  Loss within 1e-3 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import case_inputs, load_golden
from huggingface_asr_amd import shapes
from oracle import ebranchformer_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NO_DROPOUT = dict(hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, final_dropout=0.0, feat_proj_dropout=0.0,
                  csgu_conv_dropout=0.0, apply_spec_augment=False, layerdrop=0.0)


def _trainer(cfg, sd, **kw):
    from huggingface_asr_amd.train import EncoderCTCTrainer
    tr = EncoderCTCTrainer(dict(cfg, **NO_DROPOUT), DEV, **kw)
    tr.load_state_dict(sd)
    return tr


def _compare(grads, ref, rel=0.03, cos_min=0.999):
    worst = []
    gmax = max(float(torch.as_tensor(v).float().norm()) for v in ref.values())
    for k, want in ref.items():
        got = grads[k].float().cpu().reshape(-1)
        want = torch.as_tensor(want).float().reshape(-1)
        assert got.shape == want.shape, k
        nw = float(want.norm())
        err = float((got - want).norm())
        cos = float(F.cosine_similarity(got, want, dim=0)) if nw > 0 else 1.0
        if nw < 1e-5:          # mathematically zero gradients (key bias under softmax): only bf16 noise on our side, fp32 noise in the fixture
            assert err < 1e-3 * gmax, (k, err, gmax)
            continue
        worst.append((err / nw, cos, k))
    worst.sort(reverse=True)
    bad = [(e, c, k) for e, c, k in worst if e > rel or c < cos_min]
    assert not bad, f"{len(bad)} gradient tensors off; worst: {bad[:6]}"
    return worst[0]


@pytest.mark.parametrize("name,extra", [("grads_tiny_rel", {}), ("grads_tiny_rotary", {"position_embeddings_type": "rotary"}),
                                        ("grads_tiny_causal", {"is_causal": True}),
                                        ("grads_tiny_causal_long", {"is_causal": True}),
                                        ("grads_tiny_csgu_linear", {"csgu_activation": "gelu", "csgu_use_linear_after_conv": True}),
                                        ("grads_tiny_csgu_silu", {"csgu_activation": "silu"}),
                                        ("grads_tiny_gated", {"context_awareness_type": "gated"}),
                                        ("grads_tiny_gated_shared", {"context_awareness_type": "gated_shared"})])
def test_gradients_match_reference_golden(name, extra):
    """every parameter gradient of the HIP training step vs the imported reference in train() mode.  `grads_tiny_causal` = the streaming encoder
    (left-padded front end, triu attention mask, CSGU conv dilated by 15: conv_bwd.hip's dilated kernel; `_long`: 475 encoder frames, past the conv's 450-frame reach, so all 31 taps meet data); `grads_tiny_csgu_*` = the CSGU's optional Linear after the
    conv and non-identity activations (the split conv -> [Linear] -> act * gate path); `grads_tiny_gated*` = the context-aware Conv2d front ends (extractors.py:23-65;
    recipes_v0.0.1/librispeech_aed/train_gated_baseline.sh:94): conv * sigmoid(gate), the shared form with one gate row per four time steps."""
    g = load_golden(name)
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean", **extra)
    sd, x, am, lab = case_inputs(g, cfg)
    tr = _trainer(cfg, sd)
    tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"])) + 1e-3
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    grads = tr.grad_dict()
    assert set(ref) <= set(grads), sorted(set(ref) - set(grads))[:5]
    _compare(grads, ref)



def test_base_size_gradients_match_reference_strided_golden():
    """The BASE model's training step (d 512, 16 layers, head size 128 = the LDS-staged attention forward and its backward, B = 2 x 10 s, one padded row) against the
    imported reference in train() mode (tests/golden/grads_base_rel.npz, make_golden.py `run_grad_case_strided`): loss, and for each of the 672 parameter gradients
    its L2 norm and a strided 256-element sample.  Yard-stick in the fixture: the reference's own bf16-autocast backward vs its fp32 backward (mean gap 1.4-1.5 %)."""
    g = load_golden("grads_base_rel")
    cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    tr = _trainer(cfg, sd)
    tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"])), (float(out["loss"]), float(g["loss"]))
    grads = tr.grad_dict()
    names = [k[5:] for k in g.files if k.startswith("norm:")]
    assert len(names) == 672 and set(names) <= set(grads)
    gmax = max(float(g["norm:" + k]) for k in names)
    rel_n, rel_s, bad = [], [], []
    for k in names:
        got = grads[k].float().cpu().reshape(-1)
        nw = float(g["norm:" + k])
        if nw < 1e-4 * gmax:                     # mathematically (near-)zero gradients, e.g. the key bias under softmax: noise on both sides
            assert float(got.norm()) < 2e-3 * gmax, (k, float(got.norm()), gmax)
            continue
        en = abs(float(got.double().norm()) - nw) / nw
        samp = got[:: max(1, got.numel() // 256) | 1][:256].numpy()
        want = g["samp:" + k]
        es = float(np.linalg.norm(samp - want) / max(np.linalg.norm(want), 1e-12))
        cos = float(np.dot(samp, want) / max(np.linalg.norm(samp) * np.linalg.norm(want), 1e-20))
        rel_n.append(en); rel_s.append(es)
        if en > 0.04 or es > 0.08 or cos < 0.997:
            bad.append((k, round(en, 4), round(es, 4), round(cos, 5)))
    worst = sorted(zip(rel_s, rel_n, [k for k in names if float(g["norm:" + k]) >= 1e-4 * gmax]), reverse=True)[:12]
    print("worst sample errors:", [(k, round(a, 4), round(b, 4)) for a, b, k in worst])
    assert not bad, f"{len(bad)} of {len(names)} gradient tensors off: {bad[:8]}"
    # on average no worse than the reference's own bf16 backward
    assert float(np.mean(rel_s)) < 1.5 * float(g["bf16_samp_relerr_mean"]), (float(np.mean(rel_s)), float(g["bf16_samp_relerr_mean"]))
    print(f"base-size gradients: norm rel err mean {np.mean(rel_n):.4f} max {np.max(rel_n):.4f}; sample rel err mean {np.mean(rel_s):.4f} max {np.max(rel_s):.4f} "
          f"(reference bf16-vs-fp32: mean {float(g['bf16_samp_relerr_mean']):.4f})")


def test_config3_shape_joint_training_step_properties():
    """BASELINE config 3 at its real size (small encoder 12 x 256 + 6 x 256 GPT-2 decoder, per-GPU batch 96, clips of 1-20 s padded to 2000 frames; what
    `bench.py --train` times).  No reference output exists at this size (CPU), so the step is held to size-independent properties of the reference's losses:
    additivity over a split of the batch (CTC: mean of per-utterance normalised nll; CE: mean over valid tokens), independence of an utterance from its batch
    mates, invariance to extra zero padding, finite gradients in every tensor."""
    from huggingface_asr_amd import synth
    from huggingface_asr_amd.train_aed import JointAEDTrainer
    cfg = dict(shapes.SMALL, position_embeddings_type="relative", ctc_zero_infinity=True, ctc_loss_reduction="mean", **NO_DROPOUT)
    dcfg = dict(vocab_size=5000, n_embd=256, n_layer=6, n_head=4, n_positions=1024, head_locations=[], head_weights=[1.0], lsm_factor=0.1,
                layer_norm_epsilon=1e-5, pos_emb_fixed=True, tie_word_embeddings=False)
    jcfg = dict(ctc_weight=0.3, pad_token_id=3, decoder_start_token_id=1)
    tr = JointAEDTrainer(cfg, dcfg, jcfg, DEV)
    tr.enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()})
    gen = torch.Generator().manual_seed(1)
    for s_ in tr.store.specs.values():
        zero = s_.name.endswith(("_b", "bqkv", "bq", "bkv", "bo", "bco", "bfc", "bpr"))
        tr.store.p(s_.name).copy_((torch.ones(s_.shape) if s_.name.endswith("_g") else torch.randn(s_.shape, generator=gen) * (0.0 if zero else 0.02)).to(DEV))
    tr.store.refresh_mirrors(cast=True)
    B, T, U = 96, 2000, 60
    rng = np.random.default_rng(0)
    fl = np.sort(rng.integers(100, 2001, size=B))[::-1].copy()
    feats = torch.from_numpy(synth.normal(100, "feats", (B, T, 80), 1.0))
    for b in range(B):
        feats[b, fl[b]:] = 0.0
    labels = torch.from_numpy(synth.labels(0, B, U, 5000, lo=5))
    ntok = []
    for b in range(B):
        n = max(2, int(fl[b] / 100 * 3))
        labels[b, n:] = -100
        ntok.append(min(n, U))
    lens = torch.from_numpy(fl.astype(np.int32))

    def run(idx, t_pad=T):
        x = feats[idx]
        if t_pad > T:
            x = torch.cat([x, torch.zeros(len(idx), t_pad - T, 80)], 1)
        tr.enc.store.zero_grad(); tr.store.zero_grad()
        o = tr.forward_backward(x.to(DEV), lens[idx].to(DEV), labels[idx].to(DEV))
        torch.cuda.synchronize()
        return {k: float(o[k]) for k in ("loss", "enc_loss", "dec_loss")}, o["encoder_logits"].float().cpu()
    allb = np.arange(B)
    full, lg_full = run(allb)
    assert all(np.isfinite(v) for v in full.values()), full
    for st in (tr.enc.store, tr.store):
        assert bool(torch.isfinite(st.flat_g).all()) and float(st.flat_g.abs().max()) > 0
    ha, lg_a = run(allb[0::2])
    hb, _ = run(allb[1::2])
    # additivity: CTC mean over utterances; CE mean over the (shifted) valid target tokens
    assert abs(full["enc_loss"] - 0.5 * (ha["enc_loss"] + hb["enc_loss"])) < 2e-4 * full["enc_loss"], (full, ha, hb)
    na, nb = sum(ntok[i] - 1 for i in allb[0::2]), sum(ntok[i] - 1 for i in allb[1::2])        # the decoder loss is over labels[:, 1:] (double shift)
    assert abs(full["dec_loss"] - (na * ha["dec_loss"] + nb * hb["dec_loss"]) / (na + nb)) < 2e-4 * full["dec_loss"], (full, ha, hb, na, nb)
    assert abs(full["loss"] - (0.3 * full["enc_loss"] + 0.7 * full["dec_loss"])) < 1e-5 * full["loss"]
    # batch independence: an utterance's encoder logits do not depend on its batch mates (other GEMM tiles / kernels at the other M: bf16-level differences only)
    d = (lg_full[0::2] - lg_a).abs()
    assert float(d.max()) < 0.05 and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))
    # padding invariance: 100 more zero frames behind every clip
    pad, _ = run(allb, T + 100)
    for k in full:
        assert abs(pad[k] - full[k]) < 2e-4 * abs(full[k]), (k, pad[k], full[k])


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_backward_is_bit_reproducible(dropout):
    """VERDICT r3 item 4 / ADVICE r3 (medium): no parameter-gradient reduction of the training step ends in float atomics any more (bias, LayerNorm-affine, position-bias,
    depthwise-conv, conv1, embedding and masked-embedding gradients, the CTC rows, the CE loss: per-block partial rows + a fixed-order sum), so the same state, batch and
    dropout seed give the same gradient BITS twice — for the encoder + CTC step (small-encoder shapes: the grouped / 256-wide weight-gradient tiles, the K = 31 depthwise
    kernels) and for the joint model (decoder store too: embedding gather, label-smoothed CE), and three AdamW steps later the weights are still equal."""
    from huggingface_asr_amd import synth
    from huggingface_asr_amd.train import EncoderCTCTrainer
    from huggingface_asr_amd.train_aed import JointAEDTrainer
    drops = dict(hidden_dropout=dropout, activation_dropout=dropout, attention_dropout=dropout, final_dropout=dropout, feat_proj_dropout=0.0,
                 csgu_conv_dropout=dropout, layerdrop=0.0, apply_spec_augment=True, mask_time_prob=0.05, mask_time_length=4, mask_feature_prob=0.0)
    cfg = dict(shapes.SMALL, position_embeddings_type="relative", ctc_zero_infinity=True, ctc_loss_reduction="mean", **drops)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    B, T, U = 6, 420, 14
    feats = torch.from_numpy(synth.normal(5, "feats", (B, T, 80), 1.0)).to(DEV)
    lens = torch.tensor([420, 400, 333, 250, 411, 97], dtype=torch.int32, device=DEV)
    labels = torch.from_numpy(synth.labels(5, B, U, cfg["vocab_size"])).to(DEV)
    labels[:, 3] = labels[:, 1]                               # repeated labels: the CTC rows' chains
    labels[2, 9:] = -100

    def enc_run():
        np.random.seed(11)                                    # the in-model SpecAugment masks are drawn on the host (numpy's global RNG, as the reference)
        tr = EncoderCTCTrainer(cfg, DEV, lr=1e-3, seed=3)
        tr.load_state_dict(sd)
        tr.store.zero_grad()
        o = tr.forward_backward(feats, lens, labels)
        g = tr.store.flat_g.clone()
        tr.optimizer_step()
        for _ in range(2):
            tr.store.zero_grad(); tr.forward_backward(feats, lens, labels); tr.optimizer_step()
        torch.cuda.synchronize()
        return float(o["loss"]), g, tr.store.flat_p.clone()
    l0, g0, p0 = enc_run()
    l1, g1, p1 = enc_run()
    assert l0 == l1 and float(g0.abs().max()) > 0
    assert torch.equal(g0, g1), f"{int((g0 != g1).sum())} of {g0.numel()} gradient elements differ between two runs (max {float((g0 - g1).abs().max()):.3e})"
    assert torch.equal(p0, p1)

    dcfg = dict(vocab_size=500, n_embd=256, n_layer=2, n_head=4, n_positions=64, head_locations=[], head_weights=[1.0], lsm_factor=0.1, layer_norm_epsilon=1e-5,
                pos_emb_fixed=False, tie_word_embeddings=False, embd_pdrop=dropout, attn_pdrop=dropout, resid_pdrop=dropout)
    jcfg = dict(ctc_weight=0.3, pad_token_id=3, decoder_start_token_id=1)
    cfg2 = dict(cfg, vocab_size=500)
    sd2 = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg2), 0).items()}
    lab2 = torch.from_numpy(synth.labels(6, B, U, 500, lo=5)).to(DEV)
    lab2[:, 5] = lab2[:, 2]
    lab2[4, 7:] = -100

    def aed_run():
        np.random.seed(12)
        tr = JointAEDTrainer(cfg2, dcfg, jcfg, DEV, lr=1e-3)
        tr.enc.load_state_dict(sd2)
        gen = torch.Generator().manual_seed(1)
        for s_ in tr.store.specs.values():
            tr.store.p(s_.name).copy_((torch.ones(s_.shape) if s_.name.endswith("_g") else torch.randn(s_.shape, generator=gen) * 0.02).to(DEV))
        tr.store.refresh_mirrors(cast=True)
        tr.enc.store.zero_grad(); tr.store.zero_grad()
        o = tr.forward_backward(feats, lens, lab2)
        torch.cuda.synchronize()
        return float(o["loss"]), tr.enc.store.flat_g.clone(), tr.store.flat_g.clone()
    a0, a1 = aed_run(), aed_run()
    assert a0[0] == a1[0]
    for x, y, what in ((a0[1], a1[1], "encoder"), (a0[2], a1[2], "decoder")):
        assert float(x.abs().max()) > 0
        assert torch.equal(x, y), f"{what} store: {int((x != y).sum())} of {x.numel()} gradient elements differ (max {float((x - y).abs().max()):.3e})"


FINETUNE_CASES = ["finetune_tiny_mix_extra", "finetune_tiny_mix", "finetune_tiny_extra"]


def _finetune_cfg(g):
    extra, mix = (bool(v) for v in g["flags"])
    return dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean", finetune_with_additional_layer=extra, finetune_with_layer_mixing=mix)


@pytest.mark.parametrize("name", FINETUNE_CASES)
def test_finetune_head_matches_reference_golden(name):
    """BestRQEBranchformerForCTC with the recipes' fine-tuning options (bestrq.py:192-322): softmax-weighted mix of all hidden states and /
    or one more E-Branchformer layer before the CTC head — eval logits + loss and every training gradient against the reference's own."""
    g = load_golden(name)
    cfg = _finetune_cfg(g)
    sd, x, am, lab = case_inputs(g, cfg)
    tr = _trainer(cfg, sd)
    ev = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), backward=False)
    assert abs(float(ev["loss"]) - float(g["eval_loss"])) <= 1e-3 * abs(float(g["eval_loss"])) + 1e-3
    valid = int(ev["outer_len"].min())
    dl = (ev["logits"].float().cpu() - torch.from_numpy(g["eval_logits"]))[:, :valid].abs()
    assert float(dl.max()) < 0.06 and float(dl.mean()) < 0.009, (float(dl.max()), float(dl.mean()))
    # the single-call forward engine (mi_ebf_forward with extra_layers / layer_mixing): same bound against the reference, and it agrees
    # with the per-op path to bf16 rounding; with and without the caller asking for the encoder's last hidden state
    from huggingface_asr_amd import ops
    from huggingface_asr_amd.engine import EBranchformerEngine
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict(sd)
    eo = eng.forward(x.to(DEV), am.sum(-1).to(torch.int32).to(DEV), want_hidden=True)
    eo2 = eng.forward(x.to(DEV), am.sum(-1).to(torch.int32).to(DEV), want_hidden=False)
    assert torch.equal(eo["logits"], eo2["logits"])
    de = (eo["logits"].float().cpu() - torch.from_numpy(g["eval_logits"]))[:, :valid].abs()
    assert float(de.max()) < 0.06 and float(de.mean()) < 0.009, (float(de.max()), float(de.mean()))
    dt = (eo["logits"].float() - ev["logits"].float())[:, :valid].abs()
    assert float(dt.max()) < 0.06 and float(dt.mean()) < 0.006, (float(dt.max()), float(dt.mean()))
    el, _, _ = ops.ctc_loss(eo["logits"], lab.to(DEV), eo["outer_len"], reduction="mean", zero_infinity=True)
    assert abs(float(el) - float(g["eval_loss"])) <= 1e-3 * abs(float(g["eval_loss"])) + 1e-3
    assert float((eo["last_hidden"].reshape(-1, 64) - ev["last_hidden"].reshape(-1, 64)).abs().max()) < 0.1 if ev["last_hidden"] is not None else True
    tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"])) + 1e-3
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    grads = tr.grad_dict()
    assert set(ref) <= set(grads), sorted(set(ref) - set(grads))[:5]
    if cfg["finetune_with_layer_mixing"]:
        assert "per_layer_weights" in ref and float(np.abs(ref["per_layer_weights"]).max()) > 0
    if cfg["finetune_with_additional_layer"]:
        assert any(k.startswith("additional_layer.") for k in ref)
    _compare(grads, ref)
    # round trip of the extra parameters through the reference's names
    back = tr.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k


def test_finetune_head_trains_and_layerdrop_keeps_the_mix_consistent():
    """a few optimizer steps reduce the loss; with a dropped layer the mix still sees that layer's (unchanged) input, like the oracle"""
    g = load_golden("finetune_tiny_mix_extra")
    cfg = _finetune_cfg(g)
    sd, x, am, lab = case_inputs(g, cfg)
    tr = _trainer(cfg, sd, lr=1e-3)
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), backward=False, skip_layers=[1])
    want, _ = R.finetune_ctc_forward(sd, dict(cfg), x, am, lab, skip_layers=(1,))
    assert abs(float(out["loss"]) - float(want)) <= 2e-3 * abs(float(want))
    losses = [float(tr.train_step(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))["loss"]) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < 0.8 * losses[0], losses
    # the recipes' frozen fine-tuning on the native route: encoder parameters stay bit-identical (no decay, no update), everything else moves,
    # and the clip norm counts trainable gradients only
    tr = _trainer(cfg, sd, lr=1e-3, weight_decay=0.1)
    enc_keys = {k for k in sd if k.startswith("wav2vec2.encoder.")}
    tr.set_frozen(enc_keys)
    for _ in range(3):
        out = tr.train_step(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    live = torch.sqrt(sum((v.double() ** 2).sum() for k, v in tr.grad_dict().items() if k not in enc_keys))
    assert abs(float(out["grad_norm"]) - float(live)) <= 1e-4 * float(live)
    after = tr.state_dict()
    for k, v in sd.items():
        if k in enc_keys:
            assert torch.equal(after[k].cpu(), v), k
    moved = [k for k, v in sd.items() if k not in enc_keys and k != "wav2vec2.masked_spec_embed" and not torch.equal(after[k].cpu(), v)]
    assert {"per_layer_weights", "additional_layer.merge_proj.weight", "lm_head.weight", "wav2vec2.feature_projection.projection.weight"} <= set(moved)
    tr.set_frozen(())                                   # thawing restores decay and updates
    tr.train_step(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    assert not torch.equal(tr.state_dict()["wav2vec2.encoder.layers.0.merge_proj.weight"].cpu(), sd["wav2vec2.encoder.layers.0.merge_proj.weight"])


def _oracle_grads(cfg, sd, x, am, lab, skip_layers=()):
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hidden = R.encoder_forward(sdr, cfg, x, am, skip_layers=skip_layers)
    logits = R.ctc_head(sdr, hidden)
    in_len = R.conv_out_lengths_outer(am.sum(-1), cfg).long()
    tl = (lab >= 0).sum(-1)
    loss = F.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), lab[lab >= 0], in_len, tl, blank=logits.shape[-1] - 1,
                      reduction=cfg.get("ctc_loss_reduction", "mean"), zero_infinity=True)
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in sdr.items() if v.grad is not None}


def test_gradients_match_oracle_autograd_head64():
    """d=256 / 4 heads (head 64: fused LDS attention forward, probabilities recomputed in backward), no attention mask."""
    cfg = dict(shapes.TINY, hidden_size=256, intermediate_size=512, num_hidden_layers=2, vocab_size=50, ctc_zero_infinity=True, ctc_loss_reduction="sum")
    from helpers import seeded_state_dict, synth_feats, synth_labels
    sd = seeded_state_dict(cfg, 31)
    x, am = synth_feats(31, 3, 160, [160, 160, 160])
    lab = synth_labels(31, 3, 6, 50, [6, 4, 5])
    loss_ref, ref = _oracle_grads(cfg, sd, x, am, lab)
    tr = _trainer(cfg, sd)
    tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), None, lab.to(DEV))
    assert abs(float(out["loss"]) - loss_ref) <= 2e-3 * abs(loss_ref)
    _compare(tr.grad_dict(), ref, rel=0.04)


def test_layerdrop_skipped_layer_is_identity_and_gets_no_gradient():
    """LayerDrop (tf:686-690): with layer 1 of 3 dropped the loss and every gradient equal the oracle's with that layer removed; drawn
    decisions follow config.layerdrop, are reproducible per (seed, step) and never apply in eval."""
    cfg = dict(shapes.TINY, num_hidden_layers=3, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    from helpers import seeded_state_dict, synth_feats, synth_labels
    sd = seeded_state_dict(cfg, 47)
    x, am = synth_feats(47, 3, 200, [200, 170, 140])
    lab = synth_labels(47, 3, 6, cfg["vocab_size"], [6, 4, 5])
    loss_ref, ref = _oracle_grads(cfg, sd, x, am, lab, skip_layers=(1,))
    tr = _trainer(cfg, sd)
    tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), skip_layers=[1])
    assert tr.last_skipped == [1]
    assert abs(float(out["loss"]) - loss_ref) <= 2e-3 * abs(loss_ref)
    grads = tr.grad_dict()
    for k, v in grads.items():
        if ".layers.1." in k:
            assert float(v.abs().max()) == 0.0, k
    _compare(grads, {k: v for k, v in ref.items()}, rel=0.04)
    # drawn decisions: p = 1 drops everything (the encoder is then the front end + final LayerNorm), p = 0.5 is reproducible per step
    from huggingface_asr_amd.train import EncoderCTCTrainer
    tr1 = EncoderCTCTrainer(dict(cfg, **dict(NO_DROPOUT, layerdrop=1.0)), DEV, seed=5)
    tr1.load_state_dict(sd)
    tr1.store.zero_grad()
    tr1.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    assert tr1.last_skipped == [0, 1, 2]
    tr1.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), backward=False)           # eval: nothing is dropped
    assert tr1.last_skipped == []
    picks = []
    for seed in (5, 5, 6):
        trh = EncoderCTCTrainer(dict(cfg, num_hidden_layers=3, **dict(NO_DROPOUT, layerdrop=0.5)), DEV, seed=seed)
        trh.load_state_dict(sd)
        seq = []
        for step in range(8):
            trh.store.zero_grad()
            trh.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
            seq.append(tuple(trh.last_skipped))
        picks.append(seq)
    assert picks[0] == picks[1] and picks[0] != picks[2]
    n = sum(len(t) for t in picks[0])
    assert 4 <= n <= 20                                             # 24 draws at p = 0.5


def test_train_steps_reduce_loss_and_roundtrip_state_dict():
    g = load_golden("grads_tiny_rel")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    tr = _trainer(cfg, sd, lr=1e-3, weight_decay=1e-6)
    back = tr.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k
    losses = []
    for _ in range(8):
        out = tr.train_step(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
        losses.append(float(out["loss"]))
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[0], losses
    assert float(out["grad_norm"]) > 0
    # the inference engine loads the trained weights and reproduces the trainer's forward
    from huggingface_asr_amd.engine import EBranchformerEngine
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict(tr.state_dict())
    o_inf = eng.forward(x.to(DEV), am.sum(-1).to(DEV))
    o_tr = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), backward=False)
    d = (o_inf["logits"].float() - o_tr["logits"].float()).abs()
    assert float(d.max()) < 0.06 and float(d.mean()) < 0.009


def test_checkpoint_average_loads_into_the_flat_store():
    """the average of trainer checkpoints (reference names; model_utils.py:54-65) loaded back into the device store."""
    from huggingface_asr_amd import checkpoint as C
    g = load_golden("grads_tiny_rel")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    tr = _trainer(cfg, sd, lr=1e-3)
    snaps = []
    for _ in range(3):
        tr.train_step(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
        snaps.append({k: v.cpu() for k, v in tr.state_dict().items()})
    want = {k: ((snaps[0][k] + snaps[1][k]) + snaps[2][k]).div(3) for k in snaps[0]}
    first = {k: v.clone() for k, v in snaps[0].items()}
    avg = C.average_into_trainer(tr, *snaps)
    back = tr.state_dict()
    for k in want:
        assert torch.equal(avg[k], want[k]), k
        assert torch.equal(back[k].cpu(), want[k]), k
    for k in first:
        assert torch.equal(snaps[0][k], first[k]), k                                     # inputs are not modified (cloned before the in-place sum)
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), backward=False)
    assert np.isfinite(float(out["loss"]))


@pytest.mark.parametrize("name,fixed", [("grads_aed_tiny", False), ("grads_aed_tiny_fixedpos", True)])
def test_joint_aed_gradients_match_reference_golden(name, fixed):
    """JointCTCAttentionEncoderDecoder (E-Branchformer + multi-head GPT-2, auxiliary head, label smoothing, ctc_weight 0.3):
    the three losses and every parameter gradient of the HIP trainer vs the imported reference in train() mode."""
    from helpers import AED_JCFG, TINY_DEC, aed_case_inputs
    from huggingface_asr_amd.train_aed import JointAEDTrainer
    g = load_golden(name)
    sd, x, am, lab = aed_case_inputs(g)
    enc_cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean", **NO_DROPOUT)
    dec_cfg = dict(TINY_DEC, pos_emb_fixed=fixed, tie_word_embeddings=False)
    tr = JointAEDTrainer(enc_cfg, dec_cfg, AED_JCFG, DEV)
    tr.load_state_dict(sd)
    back = tr.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k
    tr.enc.store.zero_grad(); tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    torch.cuda.synchronize()
    for key in ("loss", "enc_loss", "dec_loss"):
        assert abs(float(out[key]) - float(g[key])) <= 2e-3 * abs(float(g[key])) + 1e-3, (key, float(out[key]), float(g[key]))
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    grads = tr.grad_dict()
    assert set(ref) <= set(grads), sorted(set(ref) - set(grads))[:5]
    _compare(grads, ref)
    # two optimizer steps run and reduce the loss
    l0 = float(out["loss"])
    for _ in range(4):
        o = tr.train_step(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    assert np.isfinite(float(o["loss"])) and float(o["loss"]) < l0


# ---------------------------------------------------------------------------------------------------------------- HF surface
HF_NO_DROPOUT = dict(hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, final_dropout=0.0, feat_proj_dropout=0.0,
                     ebranchformer_conv_dropout=0.0, apply_spec_augment=False, layerdrop=0.0)


def test_hf_ctc_model_trains_through_autograd_bridge():
    """AutoModelForCTC route in train() mode: `model(**batch).loss.backward()` fills the nn.Parameters' .grad (== the reference's
    gradients), a torch optimizer steps them, and default configs with dropout > 0 are refused instead of silently running without it."""
    from transformers import AutoModelForCTC
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    bind_all()
    g = load_golden("grads_tiny_rel")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    model = AutoModelForCTC.from_config(Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean", **HF_NO_DROPOUT))
    assert not any(model.load_state_dict(sd, strict=False))
    model = model.to(DEV).train()
    out = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    assert out.loss.requires_grad and abs(float(out.loss.detach()) - float(g["loss"])) <= 1e-3 * float(g["loss"]) + 1e-3
    (2.0 * out.loss).backward()                                   # an upstream factor (loss scaling / accumulation) must reach the gradients
    ref = {k[5:]: 2.0 * g[k] for k in g.files if k.startswith("grad:")}
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(ref) <= set(grads)
    _compare(grads, ref)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    l0 = float(out.loss)
    for _ in range(5):
        opt.zero_grad()
        o = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
        o.loss.backward()
        opt.step()
    assert float(o.loss) < l0
    # frozen sub-modules (freeze_encoder, train_ctc_asr.py:51-52): their parameters receive no gradient, the head still does
    model.zero_grad(set_to_none=True)
    model.freeze_encoder()
    model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV)).loss.backward()
    assert all(p.grad is None for p in model.wav2vec2.encoder.parameters()) and model.lm_head.weight.grad is not None
    tr = getattr(model, "_trainer", None)
    if tr is not None:                                        # the frozen linears' dW GEMMs were skipped: their gradient ranges stayed zero
        assert "l0.ff1_w1" in tr.frozen and "head_w" not in tr.frozen and "conv1_w" not in tr.frozen
        assert float(tr.store.g("l0.ff1_w1").abs().max()) == 0.0 and float(tr.store.g("head_w").abs().max()) > 0.0
        assert float(tr.store.g("feout_w").abs().max()) > 0.0                  # still reached through the frozen layers
    # the reference's default config (dropouts 0.1, layerdrop 0.1, in-model SpecAugment on) trains as it stands
    dflt = AutoModelForCTC.from_config(Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True)).to(DEV).train()
    assert not any(dflt.load_state_dict(sd, strict=False))
    o = dflt(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    o.loss.backward()
    assert torch.isfinite(o.loss) and all(torch.isfinite(p.grad).all() for p in dflt.parameters() if p.grad is not None)


def test_training_label_range_check_is_asynchronous_and_still_raises():
    """The reference's `labels.max() >= vocab_size` check is a host sync per forward; the training route reports it through pinned memory one forward later
    (autograd_bridge.LabelRangeCheck) — no GPU fault from the out-of-range label meanwhile (clamped), ValueError at the next training forward or at the next eval one."""
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    from huggingface_asr_amd.modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
    g = load_golden("grads_tiny_rel")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    model = Wav2Vec2EBranchformerForCTC(Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean", **HF_NO_DROPOUT))
    assert not any(model.load_state_dict(sd, strict=False))
    model.to(DEV).train()
    ok = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    bad = lab.clone(); bad[0, 0] = cfg["vocab_size"] + 3
    out = model(x.to(DEV), attention_mask=am.to(DEV), labels=bad.to(DEV))          # enqueued, not yet reported
    assert torch.isfinite(out.loss)
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    again = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))        # the report was consumed: training goes on
    assert abs(float(again.loss) - float(ok.loss)) < 1e-4 * abs(float(ok.loss))
    model(x.to(DEV), attention_mask=am.to(DEV), labels=bad.to(DEV))
    model.eval()
    with pytest.raises(ValueError), torch.no_grad():                                # an eval forward flushes what is pending
        model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))


def test_store_adamw_matches_torch_adamw_on_the_hf_route():
    """optim.StoreAdamW (the native fused clip + AdamW step behind a torch.optim.Optimizer facade, for HF Trainer's `optimizers=`) against torch's
    `clip_grad_norm_` + AdamW with the reference trainers' parameter grouping on a twin model: same losses and parameters after four steps, the copied piece
    (front-end `out` Linear) follows, optimizer state round-trips, and a model that is not on the zero-copy route is refused."""
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    from huggingface_asr_amd.modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
    from huggingface_asr_amd.optim import StoreAdamW
    g = load_golden("grads_tiny_rel")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    mk = lambda: Wav2Vec2EBranchformerForCTC(Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean", **HF_NO_DROPOUT))
    a, b = mk(), mk()
    for m in (a, b):
        assert not any(m.load_state_dict(sd, strict=False))
        m.to(DEV).train()
    with pytest.raises(RuntimeError):
        StoreAdamW(mk().to(DEV), lr=1e-3).step()                         # never ran a training forward: parameters not adopted, no fallback
    # torch side: HF Trainer's grouping (weight decay on everything but biases and LayerNorm weights: tf:trainer.py get_decay_parameter_names)
    nodecay = lambda n, p: p.ndim == 1 or n.endswith(".bias") or "pos_bias" in n or "layer_norm" in n.lower()
    decay = [p for n, p in b.named_parameters() if not nodecay(n, p)]
    rest = [p for n, p in b.named_parameters() if nodecay(n, p)]
    # eps 1e-5: elements whose gradient is pure summation noise (the key bias under softmax, far-off relative positions: |g| ~ 1e-9, float-atomic order) would
    # otherwise take full +-lr steps in run-dependent directions and dominate the comparison
    wd, lr, eps = 0.1, 2e-3, 1e-5
    opt_b = torch.optim.AdamW([dict(params=decay, weight_decay=wd), dict(params=rest, weight_decay=0.0)], lr=lr, eps=eps)
    opt_a = StoreAdamW(a, lr=lr, weight_decay=wd, eps=eps, max_grad_norm=1.0)
    la, lb = [], []
    for step in range(4):
        oa = a(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV)); oa.loss.backward(); opt_a.step(); opt_a.zero_grad()
        ob = b(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV)); ob.loss.backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0); opt_b.step(); opt_b.zero_grad()
        la.append(float(oa.loss)); lb.append(float(ob.loss))
    assert la[-1] < la[0]
    np.testing.assert_allclose(la, lb, rtol=2e-3)
    # the UPDATES agree (relative L2 distance of the whole update vector)
    pb = dict(b.named_parameters())
    num = den = 0.0
    for n, p in a.named_parameters():
        p0 = sd[n].to(DEV).float()
        ua, ub = p.detach().float() - p0, pb[n].detach().float() - p0
        num += float((ua - ub).pow(2).sum()); den += float(ub.pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < 0.01, (num / den) ** 0.5
    assert float(opt_a.last_grad_norm) > 0
    # checkpointing: state out, perturbed, back in
    st = opt_a.state_dict()
    tr = a._hip_bridge.trainer
    m0 = tr.store.flat_m.clone(); tr.store.flat_m.zero_(); step0 = tr.store.step_count; tr.store.step_count = 0
    opt_a.load_state_dict(st)
    assert torch.equal(tr.store.flat_m, m0) and tr.store.step_count == step0


def test_hf_route_parameters_and_gradients_alias_the_flat_store():
    """Round 2: after the first training forward the model's nn.Parameters ARE views of the trainer's flat fp32 master store and the
    `.grad`s autograd installs ARE views of its flat gradient store — no per-step state-dict import / gradient export; the one piece whose reference layout is
    not a view of the packed one (the front end's `out` Linear) is copied.  torch's optimizer therefore updates the masters in place; gradient accumulation
    (no zero_grad between two backwards) sums; re-allocating the parameters (`.to()`) is detected and re-adopted."""
    from transformers import AutoModelForCTC
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    bind_all()
    g = load_golden("grads_tiny_rel")
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    model = AutoModelForCTC.from_config(Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean", **HF_NO_DROPOUT))
    model.load_state_dict(sd, strict=False)
    model = model.to(DEV).train()
    batch = dict(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    model(**batch).loss.backward()
    tr = model._trainer
    pbase, gbase = tr.store.flat_p.untyped_storage().data_ptr(), tr.store.flat_g.untyped_storage().data_ptr()
    named = dict(model.named_parameters())
    copied = [n for n, p in named.items() if p.untyped_storage().data_ptr() != pbase]
    assert copied == ["wav2vec2.feature_extractor.out.weight"], copied
    conv2 = named["wav2vec2.feature_extractor.conv.1.0.conv.weight"]
    assert not conv2.is_contiguous() and conv2.untyped_storage().data_ptr() == pbase            # the channels-last packed weight seen through a permuted view
    for n, p in named.items():
        assert p.grad is not None and tuple(p.grad.shape) == tuple(p.shape), n
        if n not in copied:
            assert p.grad.untyped_storage().data_ptr() == gbase, n
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    _compare({n: p.grad for n, p in named.items()}, ref)
    for n, v in sd.items():                                       # values survived the adoption bit for bit
        if n in named:
            assert torch.equal(named[n].detach().cpu(), v), n
    # accumulation: a second backward without zero_grad doubles the gradients and keeps them where they are
    g1 = {n: p.grad.clone() for n, p in named.items()}
    model(**batch).loss.backward()
    for n, p in named.items():
        assert p.grad.untyped_storage().data_ptr() == (gbase if n not in copied else p.grad.untyped_storage().data_ptr())
        torch.testing.assert_close(p.grad, 2.0 * g1[n], rtol=2e-3, atol=1e-6 + 2e-3 * float(g1[n].abs().max()))
    # an optimizer step lands in the flat store; the next forward sees it (loss moves), and state_dict() returns contiguous copies in the reference layout
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    opt.zero_grad()
    l0 = float(model(**batch).loss.detach())
    model.zero_grad(); model(**batch).loss.backward(); opt.step(); opt.zero_grad()
    assert torch.equal(tr.store.p("head_w")[: cfg["vocab_size"]], named["lm_head.weight"].detach())
    l1 = float(model(**batch).loss.detach())
    assert l1 < l0
    back = tr.state_dict()
    assert all(v.is_contiguous() for v in back.values()) and torch.equal(back["lm_head.weight"], named["lm_head.weight"].detach())
    # re-allocated parameters are adopted again
    model.float(); model.to("cpu"); model.to(DEV)
    model.zero_grad(); model(**batch).loss.backward()
    assert dict(model.named_parameters())["lm_head.weight"].untyped_storage().data_ptr() == model._trainer.store.flat_p.untyped_storage().data_ptr()


def test_hf_joint_model_trains_through_autograd_bridge():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import aed_case_inputs
    from test_surface_cpu import _joint_model
    g = load_golden("grads_aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    model = _joint_model(False)
    for k, v in HF_NO_DROPOUT.items():
        setattr(model.config.encoder, "csgu_conv_dropout" if k == "ebranchformer_conv_dropout" else k, v)
    for k in ("resid_pdrop", "embd_pdrop", "attn_pdrop"):
        setattr(model.config.decoder, k, 0.0)
    assert not any(model.load_state_dict(sd, strict=False))
    model = model.to(DEV).train()
    out = model(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    for key in ("loss", "enc_loss", "dec_loss"):
        assert abs(float(getattr(out, key)) - float(g[key])) <= 2e-3 * abs(float(g[key])) + 1e-3, key
    out.loss.backward()
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(ref) <= set(grads)
    _compare(grads, ref)
    # optim.StoreAdamW drives BOTH flat stores of the joint trainer (encoder + decoder): the loss falls, the moments of both stores move
    from huggingface_asr_amd.optim import StoreAdamW
    opt = StoreAdamW(model, lr=1e-3, max_grad_norm=1.0)
    model.zero_grad(set_to_none=True)
    l0 = float(out.loss.detach())
    for _ in range(4):
        o = model(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
        o.loss.backward(); opt.step(); opt.zero_grad()
    assert float(o.loss.detach()) < l0
    tr = model._hip_bridge.trainer
    assert all(float(st.flat_m.abs().max()) > 0 and st.step_count == 4 for st in tr.stores())


# ---------------------------------------------------------------------------------------------------------------- dropout
def _dm(seed, step, pmap):
    """the kernels' counter-based dropout masks (csrc/dropout.hip), regenerated on the host for the oracle"""
    from huggingface_asr_amd import synth

    def dm(x, layer, site):
        p = pmap(layer, site)
        if p == 0.0:
            return x
        sid = ((step * 64 + layer) * 16 + site) & 0xFFFFFFFF
        keep = synth.dropout_keep(seed, sid, x.numel(), p).reshape(tuple(x.shape))
        return x * (torch.from_numpy(keep).float() / (1.0 - p))
    return dm


DROP = dict(activation_dropout=0.1, hidden_dropout=0.15, attention_dropout=0.1, csgu_conv_dropout=0.2, final_dropout=0.1, feat_proj_dropout=0.05)


def _enc_pmap(L):
    def pmap(layer, site):
        if layer == L:
            return {0: DROP["feat_proj_dropout"], 1: DROP["hidden_dropout"], 2: DROP["final_dropout"]}[site]
        if layer < L:
            return {0: DROP["activation_dropout"], 6: DROP["activation_dropout"], 1: DROP["hidden_dropout"], 7: DROP["hidden_dropout"],
                    2: DROP["attention_dropout"], 3: DROP["attention_dropout"], 5: DROP["attention_dropout"], 4: DROP["csgu_conv_dropout"]}[site]
        if layer == 63:
            return 0.1                                    # embd_pdrop
        return {0: 0.1, 2: 0.1, 1: 0.2, 3: 0.2, 4: 0.2}[site]   # decoder: attn_pdrop 0.1, resid_pdrop 0.2
    return pmap


@pytest.mark.parametrize("pos", ["relative", "rotary"])
def test_dropout_training_step_matches_oracle_with_identical_masks(pos):
    """All eleven dropout sites of the encoder + CTC head on (p = 0.05 .. 0.2): loss and every gradient vs torch autograd of the oracle
    run with the SAME masks; a second step draws different masks; eval-mode forward is unaffected."""
    from helpers import seeded_state_dict, synth_feats, synth_labels
    cfg = dict(shapes.TINY, position_embeddings_type=pos, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd = seeded_state_dict(cfg, 41)
    x, am = synth_feats(41, 2, 200, [200, 163])
    lab = synth_labels(41, 2, 6, 50, [6, 4])
    from huggingface_asr_amd.train import EncoderCTCTrainer
    tr = EncoderCTCTrainer(dict(cfg, layerdrop=0.0, apply_spec_augment=False, **DROP), DEV, seed=1234)
    tr.load_state_dict(sd)
    L = cfg["num_hidden_layers"]

    def oracle(step):
        dm = _dm(1234, step, _enc_pmap(L))
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        hidden = R.encoder_forward(sdr, cfg, x, am, dm=dm)
        logits = R.ctc_head(sdr, hidden, None, dm, L)
        in_len = R.conv_out_lengths_outer(am.sum(-1), cfg).long()
        loss = F.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), lab[lab >= 0], in_len, (lab >= 0).sum(-1), blank=logits.shape[-1] - 1,
                          reduction="mean", zero_infinity=True)
        loss.backward()
        return float(loss.detach()), {k: v.grad for k, v in sdr.items() if v.grad is not None}

    losses = []
    for step in range(2):
        want_loss, ref = oracle(step)
        tr.store.zero_grad()
        out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
        assert abs(float(out["loss"]) - want_loss) <= 2e-3 * want_loss, (step, float(out["loss"]), want_loss)
        _compare(tr.grad_dict(), ref, rel=0.04)
        losses.append(want_loss)
    assert abs(losses[0] - losses[1]) > 1e-3               # the step index feeds the mask streams
    ev = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), backward=False)
    with torch.no_grad():
        want_eval, _ = R.ctc_forward(sd, cfg, x, am, lab)
    assert abs(float(ev["loss"]) - float(want_eval)) <= 2e-3 * float(want_eval)


def test_dropout_finetune_head_matches_oracle_with_identical_masks():
    """the fine-tuning head in training mode: every dropout site of the additional layer (its own stream ids, layer L + 1), the head dropout on
    the mixed states, and LayerDrop of an encoder layer whose input still enters the mix — loss and all gradients vs autograd of the oracle."""
    from helpers import seeded_state_dict, synth_feats, synth_labels
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean", finetune_with_additional_layer=True, finetune_with_layer_mixing=True)
    sd = seeded_state_dict(cfg, 43)
    x, am = synth_feats(43, 2, 200, [200, 157])
    lab = synth_labels(43, 2, 6, 50, [6, 5])
    from huggingface_asr_amd.train import EncoderCTCTrainer
    tr = EncoderCTCTrainer(dict(cfg, layerdrop=0.0, apply_spec_augment=False, **DROP), DEV, seed=77)
    tr.load_state_dict(sd)
    L = cfg["num_hidden_layers"]
    base = _enc_pmap(L)
    pmap = lambda layer, site: base(0, site) if layer == L + 1 else base(layer, site)
    for step, skip in ((0, ()), (1, (1,))):
        dm = _dm(77, step, pmap)
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        hidden = R.finetune_hidden(sdr, cfg, x, am, dm=dm, skip_layers=skip)
        logits = R.ctc_head(sdr, hidden, None, dm, L)
        in_len = R.conv_out_lengths_outer(am.sum(-1), cfg).long()
        loss = F.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), lab[lab >= 0], in_len, (lab >= 0).sum(-1), blank=logits.shape[-1] - 1,
                          reduction="mean", zero_infinity=True)
        loss.backward()
        ref = {k: v.grad for k, v in sdr.items() if v.grad is not None}
        tr.store.zero_grad()
        out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV), skip_layers=list(skip))
        assert abs(float(out["loss"]) - float(loss)) <= 2e-3 * float(loss), (step, float(out["loss"]), float(loss))
        grads = tr.grad_dict()
        if skip:                       # the dropped layer has no gradient in the oracle; ours stays zero
            assert all(float(grads[k].abs().max()) == 0.0 for k in grads if k.startswith("wav2vec2.encoder.layers.1."))
        _compare(grads, ref, rel=0.04)


def test_dropout_joint_training_step_matches_oracle_with_identical_masks():
    from helpers import AED_JCFG, TINY_DEC, aed_case_inputs
    from huggingface_asr_amd.train_aed import JointAEDTrainer
    from oracle import aed_ref as A
    g = load_golden("grads_aed_tiny")
    sd, x, am, lab = aed_case_inputs(g)
    enc_cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    dec_cfg = dict(TINY_DEC, pos_emb_fixed=False, tie_word_embeddings=False)
    tr = JointAEDTrainer(dict(enc_cfg, layerdrop=0.0, apply_spec_augment=False, **DROP), dict(dec_cfg, embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.2),
                         AED_JCFG, DEV, seed=77)
    tr.load_state_dict(sd)
    dm = _dm(77, 0, _enc_pmap(enc_cfg["num_hidden_layers"]))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = A.joint_forward(sdr, enc_cfg, dec_cfg, AED_JCFG, x, am, lab, dm=dm)
    o["loss"].backward()
    ref = {k: v.grad for k, v in sdr.items() if v.grad is not None}
    tr.enc.store.zero_grad(); tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    for key in ("loss", "enc_loss", "dec_loss"):
        assert abs(float(out[key]) - float(o[key].detach())) <= 3e-3 * abs(float(o[key].detach())), (key, float(out[key]), float(o[key].detach()))
    _compare(tr.grad_dict(), ref, rel=0.04)


def test_in_model_specaugment_matches_reference_golden():
    """apply_spec_augment with time AND feature masks: the trainer draws its masks with transformers' `_compute_mask_indices` from numpy's global
    RNG like the reference does, so seeding numpy identically reproduces the reference's masks -> loss and all gradients (incl. masked_spec_embed)."""
    g = load_golden("grads_tiny_specaug")
    sa = dict(apply_spec_augment=True, mask_time_prob=0.3, mask_time_length=4, mask_time_min_masks=2, mask_feature_prob=0.2, mask_feature_length=3,
              mask_feature_min_masks=1)
    cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd, x, am, lab = case_inputs(g, cfg)
    from huggingface_asr_amd.train import EncoderCTCTrainer
    tr = EncoderCTCTrainer(dict(cfg, **{**NO_DROPOUT, **sa}), DEV)
    tr.load_state_dict(sd)
    tr.store.zero_grad()
    np.random.seed(int(g["np_seed"]))
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), lab.to(DEV))
    assert abs(float(out["loss"]) - float(g["loss"])) <= 1e-3 * float(g["loss"]) + 1e-3, (float(out["loss"]), float(g["loss"]))
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    assert "wav2vec2.masked_spec_embed" in ref
    _compare(tr.grad_dict(), ref)
    # without the masks the loss differs: the fixture really exercises SpecAugment
    base = load_golden("grads_tiny_rel")
    assert abs(float(g["loss"]) - float(base["loss"])) > 1.0


def test_bestrq_pretraining_step_matches_reference_golden():
    """BEST-RQ: random-projection-quantizer targets, masked-frame noise, classifier CE (sum / books) and every gradient of the
    HIP trainer vs the imported reference (`BestRQEBranchformerForPreTraining`, train mode, the kernels' counter-based noise injected)."""
    from helpers import BESTRQ_CFG, bestrq_case_inputs
    from huggingface_asr_amd.train_bestrq import BestRQTrainer
    g = load_golden("bestrq_tiny")
    sd, x, am, mask = bestrq_case_inputs(g)
    cfg = dict(shapes.TINY, **NO_DROPOUT, **BESTRQ_CFG)
    tr = BestRQTrainer(cfg, DEV, seed=int(g["seed"]))
    tr.load_state_dict(sd)
    tr.enc.store.zero_grad(); tr.store.zero_grad()
    out = tr.forward_backward(x.to(DEV), am.sum(-1).to(DEV), mask.to(DEV))
    tg = out["targets"].cpu().numpy()
    want = np.where(g["mask"][:, None, :], g["targets"], -100)
    assert (tg == want).mean() == 1.0, float((tg == want).mean())
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-3 * float(g["loss"]), (float(out["loss"]), float(g["loss"]))
    d = np.abs(out["last_hidden"].float().cpu().numpy() - g["last_hidden"])
    assert d.max() < 0.06 and d.mean() < 0.009
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    _compare(tr.grad_dict(), ref)
    l0 = float(out["loss"])
    for _ in range(5):
        o = tr.train_step(x.to(DEV), am.sum(-1).to(DEV), mask.to(DEV))
    assert float(o["loss"]) < l0


def test_hf_bestrq_model_forward_and_training_bridge():
    """AutoModelForPreTraining route: eval forward and train()-mode loss.backward() of the drop-in BEST-RQ class vs the reference fixture."""
    from transformers import AutoModelForPreTraining
    from helpers import BESTRQ_CFG, bestrq_case_inputs
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.modeling_bestrq import BestRQEBranchformerForPreTrainingConfig, _bestrq_cfg
    from huggingface_asr_amd.train_bestrq import BestRQTrainer
    bind_all()
    g = load_golden("bestrq_tiny")
    sd, x, am, mask = bestrq_case_inputs(g)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    model = AutoModelForPreTraining.from_config(BestRQEBranchformerForPreTrainingConfig(**base, **BESTRQ_CFG, **HF_NO_DROPOUT))
    assert not any(model.load_state_dict(sd, strict=False))
    model = model.to(DEV)
    model._trainer = BestRQTrainer(_bestrq_cfg(model.config), DEV, dp_sync=False, seed=int(g["seed"]))     # the fixture's noise seed
    model._trainer_key = None
    model.eval()
    with torch.no_grad():
        out = model(x.to(DEV), attention_mask=am.to(DEV), mask_time_indices=mask.to(DEV))
    assert abs(float(out.loss) - float(g["loss"])) <= 2e-3 * float(g["loss"])
    assert out.projected_states.shape == (2, 50, 64)
    model.train()
    model._trainer.enc.train_steps_seen = 0                                   # same noise stream (step 0) as the fixture
    out = model(x.to(DEV), attention_mask=am.to(DEV), mask_time_indices=mask.to(DEV))
    out.loss.backward()
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(ref) <= set(grads)
    _compare(grads, ref)


def test_hf_bestrq_ctc_finetune_options_forward_and_training_bridge():
    """AutoModelForCTC route of BestRQEBranchformerForCTC with the recipes' `finetune_with_additional_layer=True,finetune_with_layer_mixing=True`
    (recipes/librispeech/ssl/*/lumi/finetune_frozen*.sh): eval forward vs the reference's logits, train()-mode loss.backward() vs its gradients,
    and `freeze_encoder()` (train_ctc_asr.py:51-52) leaving exactly the encoder's parameters without a gradient."""
    from transformers import AutoModelForCTC
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.modeling_bestrq import BestRQEBranchformerForCTC, BestRQEBranchformerForPreTrainingConfig
    bind_all()
    g = load_golden("finetune_tiny_mix_extra")
    cfg = _finetune_cfg(g)
    sd, x, am, lab = case_inputs(g, cfg)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    model = AutoModelForCTC.from_config(BestRQEBranchformerForPreTrainingConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean", **HF_NO_DROPOUT,
                                                                                finetune_with_additional_layer=True, finetune_with_layer_mixing=True))
    assert isinstance(model, BestRQEBranchformerForCTC)
    assert float(model.per_layer_weights[-1]) == 1.0 and float(model.per_layer_weights[:-1].abs().sum()) == 0.0          # bestrq.py:203-205
    assert not any(model.load_state_dict(sd, strict=False))
    model = model.to(DEV).eval()
    with torch.no_grad():
        out = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
        nolab = model(x.to(DEV), attention_mask=am.to(DEV))
    assert nolab.loss is None and torch.equal(nolab.logits, out.logits)
    assert abs(float(out.loss) - float(g["eval_loss"])) <= 1e-3 * float(g["eval_loss"]) + 1e-3
    valid = int(model._get_feat_extract_output_lengths(am.sum(-1)).min())
    dl = (out.logits.float().cpu() - torch.from_numpy(g["eval_logits"]))[:, :valid].abs()
    assert float(dl.max()) < 0.06 and float(dl.mean()) < 0.009
    model.train()
    out = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    out.loss.backward()
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(ref) <= set(grads)
    _compare(grads, ref)
    # frozen encoder: the recipes' setting — front end, mixing weights, additional layer and head still train
    model.zero_grad()
    model.freeze_encoder()
    out = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    out.loss.backward()
    got = {n for n, p in model.named_parameters() if p.grad is not None}
    assert not any(n.startswith("wav2vec2.encoder.") for n in got)
    assert {"per_layer_weights", "lm_head.weight", "additional_layer.merge_proj.weight", "wav2vec2.feature_projection.projection.weight"} <= got
    live = {k: v for k, v in ref.items() if not k.startswith("wav2vec2.encoder.")}
    _compare({n: p.grad for n, p in model.named_parameters() if p.grad is not None}, live)
    # after an optimizer step the eval route sees the new weights
    with torch.no_grad():
        for p in model.parameters():
            if p.grad is not None:
                p.add_(p.grad, alpha=-1e-3)
    model.eval()
    with torch.no_grad():
        after = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    assert float(after.loss) != float(out.loss) and np.isfinite(float(after.loss))


def _tiny_ctc_model(seed=11):
    from transformers import AutoModelForCTC
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    bind_all()
    base = dict(shapes.TINY); base.pop("num_fbanks")
    model = AutoModelForCTC.from_config(Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean", **HF_NO_DROPOUT))
    from helpers import seeded_state_dict
    model.load_state_dict(seeded_state_dict(dict(shapes.TINY), seed), strict=False)
    return model.to(DEV).train()


def test_save_pretrained_after_the_bridge_adopted_the_parameters(tmp_path):
    """after the first training forward every nn.Parameter is a view of ONE flat storage, some non-contiguous (conv2's permuted weight); transformers'
    save path (`_find_disjoint` -> `_end_ptr` -> `.view(-1)`) raised on them.  The models' state-dict hook hands out private contiguous copies: training forward ->
    optimizer step -> save_pretrained -> from_pretrained returns the trained weights bit for bit, for the CTC model and for the joint model (GPT-2 Conv1D `.t()` views)."""
    from transformers import AutoModelForCTC
    from helpers import synth_feats, synth_labels
    model = _tiny_ctc_model()
    x, am = synth_feats(11, 2, 200, [198, 150])
    lab = synth_labels(11, 2, 7, 50, [7, 5])
    batch = dict(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    model(**batch).loss.backward(); opt.step(); opt.zero_grad()
    conv2 = dict(model.named_parameters())["wav2vec2.feature_extractor.conv.1.0.conv.weight"]
    assert not conv2.is_contiguous()                                     # the adoption is in place
    sd = model.state_dict()
    assert all(v.is_contiguous() and v.untyped_storage().nbytes() == v.numel() * v.element_size() for v in sd.values())
    model.save_pretrained(tmp_path / "ctc")
    back = AutoModelForCTC.from_pretrained(tmp_path / "ctc")
    for k, v in back.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].cpu()), k
    # eval after training sees the trained weights (the engine cache is keyed on the bridge's generation as well as on tensor versions)
    model.eval()
    with torch.no_grad():
        l_eval = float(model(**batch).loss)
    fresh = back.to(DEV).eval()
    with torch.no_grad():
        assert abs(float(fresh(**batch).loss) - l_eval) < 1e-6
    # the joint model
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import aed_case_inputs
    from test_surface_cpu import _joint_model
    g = load_golden("grads_aed_tiny")
    jsd, jx, jam, jlab = aed_case_inputs(g)
    jm = _joint_model(False)
    jm.load_state_dict(jsd, strict=False)
    jm = jm.to(DEV).train()
    jopt = torch.optim.SGD(jm.parameters(), lr=1e-2)
    jm(input_values=jx.to(DEV), attention_mask=jam.to(DEV), labels=jlab.to(DEV)).loss.backward(); jopt.step(); jopt.zero_grad()
    jsd2 = jm.state_dict()
    jm.save_pretrained(tmp_path / "joint")
    from safetensors.torch import load_file
    files = [f for f in os.listdir(tmp_path / "joint") if f.endswith(".safetensors")]
    assert files
    saved = {}
    for f in files:
        saved.update(load_file(str(tmp_path / "joint" / f)))
    assert len(saved) >= len([k for k in jsd2 if "attn.bias" not in k and "masked_bias" not in k]) - 2
    for k, v in saved.items():
        assert torch.equal(v, jsd2[k].cpu()), k


def test_backward_of_a_stale_forward_is_refused():
    """The gradients of a step live in the trainer's flat store: forward A, forward B, A.backward() would hand A the gradients of B.  Refused, as is a second backward."""
    from helpers import synth_feats, synth_labels
    model = _tiny_ctc_model()
    x, am = synth_feats(11, 2, 200, [198, 150])
    lab = synth_labels(11, 2, 7, 50, [7, 5])
    batch = dict(input_values=x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV))
    la = model(**batch).loss
    lb = model(**batch).loss
    with pytest.raises(RuntimeError, match="no longer the latest"):
        la.backward()
    lb.backward()
    lc = model(**batch).loss
    lc.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="twice"):
        lc.backward()


def test_hf_trainer_with_store_adamw_saves_and_resumes(tmp_path):
    """the documented route `Trainer(optimizers=(StoreAdamW(model), None))`.  accelerate's `prepare(optimizer)` round-trips the optimizer state BEFORE any
    forward and a resume loads it before the first step; both used to raise (the parameters are adopted by the first training forward).  Two steps of a real HF Trainer with
    save_steps=1, then a fresh model + optimizer resumed from checkpoint-1 reproduces step 2: same weights as the uninterrupted run."""
    from transformers import Trainer, TrainingArguments
    from huggingface_asr_amd.optim import StoreAdamW
    from helpers import synth_feats, synth_labels
    x, am = synth_feats(11, 4, 200, [198, 150, 200, 120])
    lab = synth_labels(11, 4, 7, 50, [7, 5, 6, 4])
    data = [dict(input_values=x[i], attention_mask=am[i], labels=lab[i]) for i in range(4)]

    def run(out, max_steps, resume=None):
        model = _tiny_ctc_model()
        args = TrainingArguments(output_dir=str(out), per_device_train_batch_size=2, max_steps=max_steps, save_strategy="steps", save_steps=1, learning_rate=1e-3,
                                 lr_scheduler_type="constant", max_grad_norm=0.0, report_to=[], remove_unused_columns=False, dataloader_num_workers=0, seed=3,
                                 logging_steps=1, dataloader_drop_last=True)
        opt = StoreAdamW(model, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
        sd0 = opt.state_dict()                                                # before adoption: an empty state, not a raise
        assert sd0["state"]["stores"] == []
        opt.load_state_dict(sd0)
        tr = Trainer(model=model, args=args, train_dataset=data, optimizers=(opt, torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)))
        tr.train(resume_from_checkpoint=resume)
        return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, opt
    full, opt_full = run(tmp_path / "a", 2)
    assert (tmp_path / "a" / "checkpoint-1").is_dir() and (tmp_path / "a" / "checkpoint-2").is_dir()
    assert opt_full.state_dict()["state"]["stores"][0]["step"] == 2
    resumed, opt_res = run(tmp_path / "a", 2, resume=str(tmp_path / "a" / "checkpoint-1"))
    assert opt_res.state_dict()["state"]["stores"][0]["step"] == 2
    for k, v in full.items():
        torch.testing.assert_close(resumed[k], v, rtol=1e-5, atol=1e-6, msg=k)
