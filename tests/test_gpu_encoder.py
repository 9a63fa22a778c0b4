"""GPU parity, end to end: the HIP encoder (mi_ebf_forward through the C ABI) against
 (1) the golden vectors produced by the imported REFERENCE (fp32) and
 (2) the oracle run with the bf16 storage model of the kernels (q = bf16_round).

north_star tolerance: logits / CTC loss within 1e-3 in bf16.  Measured on the reference itself
(tests/golden/base_rel.npz: its own bf16-autocast vs fp32 forward) the logit gap is max 0.059 / mean 0.0088 and
the relative loss gap 8e-5, so the bar is stated as: relative CTC-loss error <= 1e-3 against the fp32 reference,
logit error no worse than the reference's own bf16 gap (max <= 0.06, mean <= 0.009 at logit std 0.81), and a
tighter bound against the bf16-modelled oracle (same rounding points, only accumulation order differs)."""
import numpy as np
import pytest
import torch

from helpers import case_inputs, load_golden
from huggingface_asr_amd import shapes
from oracle import ebranchformer_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(base, **kw):
    c = dict(base)
    c.update(ctc_zero_infinity=True, ctc_loss_reduction="mean")
    c.update(kw)
    return c


def _run_engine(cfg, sd, x, am, lab):
    from huggingface_asr_amd import ops
    from huggingface_asr_amd.engine import EBranchformerEngine
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict(sd)
    out = eng.forward(x.to(DEV), am.sum(-1).to(DEV, torch.int32))
    loss, nll, tl = ops.ctc_loss(out["logits"], lab.to(DEV), out["outer_len"], reduction="mean", zero_infinity=True)
    torch.cuda.synchronize()
    return out, float(loss)


TINY = [
    ("tiny_rel", _cfg(shapes.TINY)),
    ("tiny_rotary", _cfg(shapes.TINY, position_embeddings_type="rotary")),
    ("tiny_causal", _cfg(shapes.TINY, is_causal=True)),
    ("tiny_nomacaron", _cfg(shapes.TINY, csgu_activation="gelu", csgu_use_linear_after_conv=True)),      # CSGU: conv -> Linear -> GELU -> gate (e_branchformer.py:196-201)
    # context-aware Conv2d front ends (extractors.py:23-65; train_gated_baseline.sh:94): conv * sigmoid(gate); the shared gate = one row per four time steps;
    # the recipes' misspelt `shared_gated` resolves to the plain conv in the reference's dict lookup and must do so here
    ("tiny_gated", _cfg(shapes.TINY, context_awareness_type="gated")),
    ("tiny_gated_shared", _cfg(shapes.TINY, context_awareness_type="gated_shared")),
    ("tiny_shared_gated_fallthrough", _cfg(shapes.TINY, context_awareness_type="shared_gated")),
]


@pytest.mark.parametrize("name,cfg", TINY, ids=[c[0] for c in TINY])
def test_tiny_vs_reference_and_oracle(name, cfg):
    g = load_golden(name)
    sd, x, am, lab = case_inputs(g, cfg)
    out, loss = _run_engine(cfg, sd, x, am, lab)
    logits = out["logits"].float().cpu().numpy()
    hidden = out["last_hidden"].cpu().numpy()
    np.testing.assert_array_equal(out["outer_len"].cpu().numpy(), g["outer_lens"])
    np.testing.assert_array_equal(out["inner_len"].cpu().numpy(), np.minimum(g["inner_lens"], hidden.shape[1]))
    # (1) reference fp32 golden
    d = np.abs(logits - g["logits"])
    assert d.max() < 0.06 and d.mean() < 0.009, (d.max(), d.mean())
    assert abs(loss - float(g["loss"])) < 1e-3 * abs(float(g["loss"])), (loss, float(g["loss"]))
    assert np.abs(hidden - g["last_hidden"]).mean() < 0.01
    # (2) oracle with the kernels' bf16 storage model
    with torch.no_grad():
        hq = R.encoder_forward(sd, cfg, x, am, q=R.bf16_round)
        lq = R.ctc_head(sd, hq, q=R.bf16_round).numpy()
    dq = np.abs(logits - lq)
    assert dq.max() < 0.03 and dq.mean() < 0.003, (dq.max(), dq.mean())


BIG = [
    ("small_rel", _cfg(shapes.SMALL)),
    ("small_causal", _cfg(shapes.SMALL, is_causal=True)),           # the streaming model at a real size (config 5's encoder side)
    ("base_rel", _cfg(shapes.BASE)),
    ("base_rotary", _cfg(shapes.BASE, position_embeddings_type="rotary")),
    ("small_gated", _cfg(shapes.SMALL, context_awareness_type="gated")),      # 256-channel gated front end: two filter banks in conv1, conv + gate rows in ONE implicit GEMM (fused epilogue)
]


@pytest.mark.parametrize("name,cfg", BIG, ids=[c[0] for c in BIG])
def test_small_base_vs_reference(name, cfg):
    g = load_golden(name)
    sd, x, am, lab = case_inputs(g, cfg)
    out, loss = _run_engine(cfg, sd, x, am, lab)
    logits = out["logits"].float().cpu().numpy()
    assert logits.shape == (2, 250, 5001)
    np.testing.assert_array_equal(out["outer_len"].cpu().numpy(), g["outer_lens"])
    d1 = np.abs(logits[:, ::25, :64] - g["logits_slice"])
    d2 = np.abs(logits[:, :, -1] - g["logits_blank"])
    assert max(d1.max(), d2.max()) < 0.06 and max(d1.mean(), d2.mean()) < 0.009, (d1.max(), d2.max(), d1.mean(), d2.mean())
    assert abs(float(logits.std()) - float(g["logits_std"])) < 2e-3
    assert abs(loss - float(g["loss"])) < 1e-3 * abs(float(g["loss"])), (loss, float(g["loss"]))


def test_gated_shared_refuses_lengths_the_reference_cannot_view():
    """GatedConv2dShared views the conv output as (B, C, -1, 4, F) (extractors.py:52): a time axis not divisible by 4 raises in the reference; here too, never silence."""
    from huggingface_asr_amd import synth
    from huggingface_asr_amd.engine import EBranchformerEngine
    cfg = _cfg(shapes.TINY, context_awareness_type="gated_shared")
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()})
    with pytest.raises(RuntimeError, match="gated_shared"):
        eng.forward(torch.zeros(1, 200, 80, device=DEV), None)                    # 200 -> 100 -> 50: 50 % 4 != 0
    assert eng.forward(torch.zeros(1, 208, 80, device=DEV), None)["logits"].shape[1] == 52


def test_gated_layer2_fused_epilogue_equals_unfused():
    """GatedConv2d layer 2: the fused form (conv + gate rows interleaved by 32 in ONE implicit GEMM, product + GELU in the 256-wide kernel's epilogue) against the
    un-fused form (the same GEMM raw + mi_gated_act_bf16) and against torch on the same bf16 operands."""
    import torch.nn.functional as F
    from huggingface_asr_amd import ops, synth
    B, T1, F1, C1, C2 = 2, 120, 40, 64, 256
    x = torch.from_numpy(synth.normal(3, "g2x", (B, T1, F1, C1), 1.0)).to(DEV).to(torch.bfloat16)
    wc = torch.from_numpy(synth.normal(3, "g2wc", (C2, C1, 3, 3), 0.05)).to(DEV).to(torch.bfloat16)
    wg = torch.from_numpy(synth.normal(3, "g2wg", (C2, C1, 3, 3), 0.05)).to(DEV).to(torch.bfloat16)
    bc = torch.from_numpy(synth.normal(3, "g2bc", (C2,), 0.3)).to(DEV)
    bg = torch.from_numpy(synth.normal(3, "g2bg", (C2,), 0.3)).to(DEV)
    cl = lambda w: w.permute(0, 2, 3, 1).reshape(C2, -1)
    nb = C2 // 32
    wp = torch.stack([cl(wc).view(nb, 32, -1), cl(wg).view(nb, 32, -1)], 1).reshape(2 * C2, -1).contiguous()
    bp = torch.stack([bc.view(nb, 32), bg.view(nb, 32)], 1).reshape(2 * C2).contiguous()
    fused = ops.conv2d_cl_geo(x, wp, bp, act="gelu", gated=True)
    raw = ops.conv2d_cl_geo(x, wp, bp, act="none", gated=False)
    T2, F2 = fused.shape[1], fused.shape[2]
    unf = ops.gated_act(raw, raw, B, T2, F2, C2, 1, 32).view(B, T2, F2, C2)
    xt = x.float().permute(0, 3, 1, 2)
    z = F.conv2d(xt, wc.float(), bc, stride=2, padding=1)
    g = F.conv2d(xt, wg.float(), bg, stride=2, padding=1)
    want = F.gelu(z * torch.sigmoid(g)).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert (fused.float() - want).abs().max() < 0.02 * max(1.0, float(want.abs().max()))          # one bf16 rounding of the result
    assert (unf.float() - want).abs().max() < 0.04 * max(1.0, float(want.abs().max()))            # + bf16 rounding of the raw conv / gate outputs
    assert (fused.float() - want).abs().mean() < 2e-3


@pytest.mark.parametrize("name,cfg", [BIG[0], BIG[2]], ids=["small_rel", "base_rel"])
def test_layernorm_fold_path_vs_reference(name, cfg):
    """The engine's `ln_fold` path (LayerNorms folded into the FFN-in / QKV / cgMLP-in GEMMs, statistics and the bf16 residual copy out of the producing GEMMs' epilogues; it
    selects itself at bench-size batches) forced on the two-utterance reference fixtures: the same bounds against the reference as the un-folded kernels, and both paths
    agree with each other far inside them."""
    from huggingface_asr_amd import ops
    from huggingface_asr_amd.engine import EBranchformerEngine
    g = load_golden(name)
    sd, x, am, lab = case_inputs(g, cfg)
    outs = {}
    for mode in ("fold", "plain", "wide"):
        fold = mode != "plain"
        eng = EBranchformerEngine(cfg, DEV)
        eng.ln_fold = fold
        eng.wide_tiles = mode == "wide"                                                # throughput mode: the N = d GEMMs on the 256 x 256 tile (mi_ebf_config.wide_tiles)
        eng.load_state_dict(sd)
        cs = eng._config_struct(2, 1000, 80)
        assert cs.ln_fold == int(fold) and cs.wide_tiles == int(mode == "wide")
        out = eng.forward(x.to(DEV), am.sum(-1).to(DEV, torch.int32))
        loss, _, _ = ops.ctc_loss(out["logits"], lab.to(DEV), out["outer_len"], reduction="mean", zero_infinity=True)
        outs[mode] = (out["logits"].float().cpu().numpy(), float(loss), out["last_hidden"].cpu().numpy())
    for mode in ("fold", "wide"):
        logits, loss, hid = outs[mode]
        d1 = np.abs(logits[:, ::25, :64] - g["logits_slice"])
        d2 = np.abs(logits[:, :, -1] - g["logits_blank"])
        assert max(d1.max(), d2.max()) < 0.06 and max(d1.mean(), d2.mean()) < 0.009, (mode, d1.max(), d2.max(), d1.mean(), d2.mean())
        assert abs(loss - float(g["loss"])) < 1e-3 * abs(float(g["loss"])), (mode, loss, float(g["loss"]))
        dd = np.abs(logits - outs["plain"][0])
        assert dd.max() < 0.05 and dd.mean() < 0.006, (mode, dd.max(), dd.mean())
        assert np.abs(hid[1, 175:]).max() < 10 and np.isfinite(hid).all()                 # padded frames stay finite (their statistics are those of a zero row)
    dw = np.abs(outs["wide"][0] - outs["fold"][0])                                       # the two tiles differ by summation order only
    assert dw.max() < 0.02 and dw.mean() < 0.002, (dw.max(), dw.mean())


def test_no_attention_mask_and_batch_invariance():
    """No mask -> all frames valid; an utterance's logits do not depend on its batch neighbours."""
    from huggingface_asr_amd.engine import EBranchformerEngine
    cfg = _cfg(shapes.TINY)
    g = load_golden("tiny_rel")
    sd, x, am, lab = case_inputs(g, cfg)
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict(sd)
    full = torch.ones_like(am)
    a = eng.forward(x.to(DEV), None)["logits"].clone()
    b = eng.forward(x.to(DEV), full.sum(-1).to(DEV, torch.int32))["logits"].clone()
    assert torch.equal(a, b)
    one = eng.forward(x[:1].to(DEV), None)["logits"]
    torch.testing.assert_close(one[0], a[0], atol=0, rtol=0)


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_forward_returns_the_logits_row_log_sum_exp(size):
    """engine.forward(...)["lse"] (the head GEMM's epilogue, mi_ebf_forward_lse; the tiny config's head is outside that kernel and takes the GEMM + mi_row_lse route
    behind the same entry): equal to the log-sum-exp of the returned logits, and the CTC loss computed with it equals the loss computed from the logits alone"""
    from huggingface_asr_amd import ops, synth
    from huggingface_asr_amd.engine import EBranchformerEngine
    if size == "tiny":
        cfg = _cfg(shapes.TINY)
        g = load_golden("tiny_rel")
        sd, x, am, lab = case_inputs(g, cfg)
        lens = am.sum(-1).to(DEV, torch.int32)
        x, lab = x.to(DEV), lab.to(DEV)
    else:
        cfg = _cfg(shapes.BASE)
        sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 3).items()}
        B, T = 8, 1000
        x = torch.from_numpy(synth.normal(5, "feats", (B, T, 80), 1.0)).to(DEV)
        lens = torch.tensor([1000, 998, 900, 777, 640, 512, 300, 120], dtype=torch.int32, device=DEV)
        lab = torch.from_numpy(synth.labels(1, B, 12, cfg["vocab_size"], lo=5)).to(DEV)
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict(sd)
    assert eng.forward(x, lens, want_hidden=False)["lse"] is None            # off by default (measured: no gain)
    eng.head_lse = True
    out = eng.forward(x, lens, want_hidden=False)
    lg = out["logits"]
    Bz, T2, V1 = lg.shape
    want = torch.logsumexp(lg.double(), dim=-1).reshape(-1)
    torch.testing.assert_close(out["lse"].double(), want, rtol=2e-6, atol=2e-6)
    l1 = ops.ctc_loss(lg, lab, out["outer_len"], reduction="mean", zero_infinity=True, lse=out["lse"])[0]
    l0 = ops.ctc_loss(lg, lab, out["outer_len"], reduction="mean", zero_infinity=True)[0]
    torch.testing.assert_close(l1, l0, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("overlap", [False, True])
def test_forward_is_bit_reproducible_at_the_bench_size(overlap):
    """The whole forward gives identical bits run after run, also when issued from a non-default stream: every reduction in it has a
    fixed order.  overlap=True is the opt-in two-stream form (attention and cgMLP branches side by side, HFASR_BRANCH_OVERLAP=1): it
    must reproduce the single-stream bits exactly ('Concurrent kernels': the pairing that used to corrupt the CSGU tile)."""
    from huggingface_asr_amd import synth
    from huggingface_asr_amd.engine import EBranchformerEngine
    cfg = _cfg(shapes.BASE)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    eng = EBranchformerEngine(cfg, DEV)
    if overlap:
        eng.ln_fold = False          # the two-stream form exists for the un-folded layer only (engine._use_fold): compare like with like
    eng.load_state_dict(sd)
    feats = torch.from_numpy(synth.normal(1, "feats", (16, 1000, 80), 1.0)).to(DEV)
    lens = torch.full((16,), 998, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    o = eng.forward(feats, lens)
    ref = (o["logits"].clone(), o["last_hidden"].clone())            # single-stream reference
    eng.branch_overlap = overlap
    for i in range(12 if overlap else 8):
        if i % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                o = eng.forward(feats, lens)
            torch.cuda.current_stream().wait_stream(side)
        else:
            o = eng.forward(feats, lens)
        got = (o["logits"].clone(), o["last_hidden"].clone())
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), i


@pytest.mark.parametrize("lanes,wide", [(2, False), (4, True)], ids=["two_lanes", "four_lanes_wide_tiles"])
def test_pipelined_steps_on_two_streams_reproduce_the_single_stream_bits(lanes, wide):
    """bench.py's default: consecutive steps (each its own batch of 32 clips, its own engine / workspace) in flight together on HIP streams of their own.  Whatever shares
    the chip with a step's kernels, its logits are the bits the same engine gives alone — 24 overlapped steps at the bench size against the per-lane references.
    Second case: the throughput mode (256 x 256 tiles for the N = d GEMMs, four steps in flight)."""
    from huggingface_asr_amd import synth
    from huggingface_asr_amd.engine import EBranchformerEngine
    cfg = _cfg(shapes.BASE)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    from huggingface_asr_amd.pipeline import ForwardPipeline
    lens = torch.full((32,), 998, dtype=torch.int32, device=DEV)
    pipe = ForwardPipeline(cfg, DEV, sd, lanes=lanes, wide_tiles=wide)
    assert all(e._config_struct(32, 1000, 80).wide_tiles == int(wide) for e in pipe.engines)
    feats = [torch.from_numpy(synth.normal(11 + i, "feats", (32, 1000, 80), 1.0)).to(DEV) for i in range(lanes)]
    refs = [pipe.engines[i].forward(feats[i], lens)["logits"].clone() for i in range(lanes)]          # each engine alone on the default stream
    torch.cuda.synchronize()
    if wide:                                                                                          # and the wide tiles compute the product tile's rows
        plain = EBranchformerEngine(cfg, DEV)
        plain.load_state_dict(sd)
        dd = (plain.forward(feats[0], lens)["logits"].float() - refs[0].float()).abs()
        assert float(dd.max()) < 0.05 and float(dd.mean()) < 0.003, (float(dd.max()), float(dd.mean()))
    outs = [pipe.submit(lambda e, lane: e.forward(feats[lane], lens)["logits"].clone()) for _ in range(24)]       # the clone runs on the lane's stream, behind its forward
    torch.cuda.synchronize()
    assert [lane for lane, _ in outs] == [j % lanes for j in range(24)]
    for k, (lane, got) in enumerate(outs):
        assert torch.equal(got, refs[lane]), k


@pytest.mark.parametrize("fold", [True, False])
def test_full_size_batch_independence_and_loss_additivity(fold):
    """BASELINE configs[1] size (base, 32 x 10 s): size-independent properties instead of an oracle run — an utterance's logits do not depend on
    its batch neighbours (bit for bit: every kernel reduces over an utterance's own rows only), extra zero padding behind an utterance does not
    move its valid frames, and the batch-mean CTC loss is the mean of the per-utterance losses.  Both layer forms: LayerNorms folded into the GEMMs (what a batch of this
    size selects by itself) and LayerNorm kernels + plain GEMMs (what a batch of one selects: the form is pinned here so that like is compared with like)."""
    from huggingface_asr_amd import ops, synth
    from huggingface_asr_amd.engine import EBranchformerEngine
    cfg = _cfg(shapes.BASE)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    eng = EBranchformerEngine(cfg, DEV)
    assert eng._use_fold(32, 250) and not eng._use_fold(1, 250)
    eng.ln_fold = fold
    eng.load_state_dict(sd)
    B, T = 32, 1000
    feats = torch.from_numpy(synth.normal(7, "feats", (B, T, 80), 1.0)).to(DEV)
    lens = torch.tensor([998 - 37 * (i % 9) for i in range(B)], dtype=torch.int32, device=DEV)
    feats = feats * (torch.arange(T, device=DEV)[None, :, None] < lens[:, None, None])          # the collator's zero padding
    labels = torch.from_numpy(synth.labels(7, B, 40, cfg["vocab_size"])).to(DEV)
    out = eng.forward(feats, lens)
    logits, outer = out["logits"].clone(), out["outer_len"].clone()
    assert torch.isfinite(logits).all()
    for i in (0, 5, 31):
        one = eng.forward(feats[i:i + 1].contiguous(), lens[i:i + 1].contiguous())["logits"]
        n = int(outer[i])
        assert torch.equal(one[0, :n], logits[i, :n]), i
    # 200 more frames of padding: T' grows from 250 to 300, the valid frames keep their values up to bf16 re-rounding of the position table
    padded = torch.zeros(1, T + 200, 80, device=DEV); padded[0, :T] = feats[5]
    more = eng.forward(padded, lens[5:6].contiguous())["logits"]
    n = int(outer[5])
    dpad = (more[0, :n] - logits[5, :n]).abs()                      # bf16 noise of 16 layers (the reference's own bf16-vs-fp32 gap is 0.044 max)
    assert float(dpad.max()) < 0.12 and float(dpad.mean()) < 0.01, (float(dpad.max()), float(dpad.mean()))
    loss, nll, tl = ops.ctc_loss(logits, labels, outer, reduction="mean", zero_infinity=True)
    want = (nll / tl.clamp(min=1)).mean()
    assert abs(float(loss) - float(want)) < 1e-4 * abs(float(want))
    each = [float(ops.ctc_loss(logits[i:i + 1].contiguous(), labels[i:i + 1].contiguous(), outer[i:i + 1].contiguous(), reduction="sum", zero_infinity=True)[0]) for i in (0, 9)]
    assert abs(each[0] - float(nll[0])) < 1e-3 * abs(each[0]) and abs(each[1] - float(nll[9])) < 1e-3 * abs(each[1])


def test_engine_requires_device_tensors():
    from huggingface_asr_amd.engine import EBranchformerEngine
    cfg = _cfg(shapes.TINY)
    g = load_golden("tiny_rel")
    sd, x, am, lab = case_inputs(g, cfg)
    eng = EBranchformerEngine(cfg, DEV)
    eng.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        eng.forward(x, None)        # CPU tensor: no fallback


def test_hf_model_surface_matches_reference_golden():
    """The drop-in class (AutoModelForCTC route): state dict in, CausalLMOutput out, same loss/logits as the reference."""
    from transformers import AutoModelForCTC
    from huggingface_asr_amd.bind import bind_all
    from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
    bind_all()
    cfg = _cfg(shapes.TINY)
    g = load_golden("tiny_rel")
    sd, x, am, lab = case_inputs(g, cfg)
    base = dict(shapes.TINY); base.pop("num_fbanks")
    hf_cfg = Wav2Vec2EBranchformerConfig(**base, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    model = AutoModelForCTC.from_config(hf_cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not missing
    model = model.to(DEV).eval()
    with torch.no_grad():
        out = model(x.to(DEV), attention_mask=am.to(DEV), labels=lab.to(DEV), output_hidden_states=True)
    d = np.abs(out.logits.float().cpu().numpy() - g["logits"])
    assert d.max() < 0.06 and d.mean() < 0.009
    assert abs(float(out.loss) - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    assert out.hidden_states[-1].shape == (2, 50, 64)
    # output_hidden_states: L + 1 tensors as HuggingFace returns them (the input of every layer, then the last hidden state), against the reference's own tuple
    assert len(out.hidden_states) == cfg["num_hidden_layers"] + 1
    for i in range(cfg["num_hidden_layers"]):
        dh = np.abs(out.hidden_states[i].cpu().numpy() - g[f"layer_in_{i}"])
        assert dh.mean() < 0.01 and dh.max() < 0.12, (i, dh.mean(), dh.max())
    assert np.abs(out.hidden_states[-1].cpu().numpy() - g["last_hidden"]).mean() < 0.01
    assert np.array_equal(out.hidden_states[0].cpu().numpy()[1, 38:], g["layer_in_0"][1, 38:])      # padded frames of layer 0's input are zero (tf:662-665)
    # weights edited in place -> engine repacks (version bump)
    with torch.no_grad():
        model.lm_head.bias.add_(1.0)
        out2 = model(x.to(DEV), attention_mask=am.to(DEV))
    torch.testing.assert_close(out2.logits[..., :-1], out.logits[..., :-1] + 1.0, atol=1e-5, rtol=0)
    bad = lab.clone(); bad[0, 0] = 50
    with pytest.raises(ValueError):
        model(x.to(DEV), attention_mask=am.to(DEV), labels=bad.to(DEV))
