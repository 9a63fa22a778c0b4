"""State-dict contract of the E-Branchformer CTC encoder (names + shapes).

The key names are an external contract (checkpoint averaging / resume in the reference,
SURVEY.md §8b); they mirror what `Wav2Vec2EBranchformerForCTC.state_dict()` yields in the
reference (`src/models/encoders/e_branchformer.py:225-261,408-416`, `src/models/extractors.py:68-131`).
"""
from __future__ import annotations


def conv_freq_out(num_fbanks: int, kernels, strides, paddings, causal: bool = False) -> int:
    """reference src/models/utils.py:4-38 for the frequency axis."""
    f = num_fbanks
    for k, s, p in zip(kernels, strides, paddings):
        f = (f + ((2 * p) if causal else 2 * p) - (k - 1) - 1) // s + 1
    return f


GATE_SHARE = 4          # GatedConv2dShared's shared_scale_factor (extractors.py:36): one gate value per four output time steps


def context_mode(cfg: dict) -> int:
    """0 plain / 1 "gated" / 2 "gated_shared" Conv2d sub-sampling (extractors.py:57-65).  The reference looks the string up in a two-entry dict and
    falls back to nn.Conv2d for ANYTHING else — None, and also the `shared_gated` that recipes_v0.0.1/librispeech_aed/train_shared_gated_baseline.sh:94
    passes — so unknown strings are the plain conv here too; a causal encoder ignores the field altogether (CausalConv2d, extractors.py:74-81)."""
    if cfg.get("is_causal", False):
        return 0
    return {"gated": 1, "gated_shared": 2}.get(cfg.get("context_awareness_type"), 0)


def gate_geometry(k: int, s: int, p: int, mode: int):
    """(KH, KW, stride_t, stride_f, pad_t, pad_f) of the gate conv of a layer whose conv is (k, k) / s / p (extractors.py:27, 41-47)."""
    return (k, k, s, s, p, p) if mode == 1 else (k * GATE_SHARE, k, s * GATE_SHARE, s, p * GATE_SHARE, p)


def param_shapes(cfg: dict) -> dict:
    d, I, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    H = cfg["num_attention_heads"]
    V = cfg["vocab_size"]
    conv_dim, ks = list(cfg["conv_dim"]), list(cfg["conv_kernel"])
    pads = list(cfg.get("conv_padding", [1] * len(ks)))
    ptype = cfg.get("position_embeddings_type", "relative")
    kc, km = cfg.get("csgu_kernel_size", 31), cfg.get("merge_conv_kernel", 31)
    out = {}
    if cfg.get("mask_time_prob", 0.05) > 0.0 or cfg.get("mask_feature_prob", 0.0) > 0.0:
        out["wav2vec2.masked_spec_embed"] = (d,)
    cin = 1
    # CausalConv2d IS the nn.Conv2d (streaming_modules.py:31) while the non-causal conv is wrapped in
    # ContextAwareConv2d (extractors.py:57-65), hence the extra ".conv" in the non-causal key.
    cw = "" if cfg.get("is_causal", False) else ".conv"
    mode = context_mode(cfg)
    for i, (c, k) in enumerate(zip(conv_dim, ks)):
        if mode:        # ContextAwareConv2d.conv is a Gated* module holding .conv and .gate (extractors.py:23-54)
            out[f"wav2vec2.feature_extractor.conv.{i}.0.conv.conv.weight"] = (c, cin, k, k)
            out[f"wav2vec2.feature_extractor.conv.{i}.0.conv.conv.bias"] = (c,)
            out[f"wav2vec2.feature_extractor.conv.{i}.0.conv.gate.weight"] = (c, cin, k * (GATE_SHARE if mode == 2 else 1), k)
            out[f"wav2vec2.feature_extractor.conv.{i}.0.conv.gate.bias"] = (c,)
        else:
            out[f"wav2vec2.feature_extractor.conv.{i}.0{cw}.weight"] = (c, cin, k, k)
            out[f"wav2vec2.feature_extractor.conv.{i}.0{cw}.bias"] = (c,)
        cin = c
    fo = conv_freq_out(cfg.get("num_fbanks", 80), ks, cfg["conv_stride"], pads, cfg.get("is_causal", False))
    out["wav2vec2.feature_extractor.out.weight"] = (d, conv_dim[-1] * fo)
    out["wav2vec2.feature_extractor.out.bias"] = (d,)
    out["wav2vec2.feature_projection.layer_norm.weight"] = (d,)
    out["wav2vec2.feature_projection.layer_norm.bias"] = (d,)
    out["wav2vec2.feature_projection.projection.weight"] = (d, d)
    out["wav2vec2.feature_projection.projection.bias"] = (d,)
    out["wav2vec2.encoder.layer_norm.weight"] = (d,)
    out["wav2vec2.encoder.layer_norm.bias"] = (d,)
    prefixes = [f"wav2vec2.encoder.layers.{i}." for i in range(L)]
    if cfg.get("finetune_with_layer_mixing", False):          # BestRQEBranchformerForCTC (bestrq.py:202-205)
        out["per_layer_weights"] = (L + 1,)
    if cfg.get("finetune_with_additional_layer", False):      # bestrq.py:199-200
        prefixes.append("additional_layer.")
    for p in prefixes:
        ffs = ("ff1", "ff2") if cfg.get("use_macaron_ff", True) else ()
        if "ff1" in ffs:
            out[p + "ff1.0.weight"] = (d,); out[p + "ff1.0.bias"] = (d,)
            out[p + "ff1.1.intermediate_dense.weight"] = (I, d); out[p + "ff1.1.intermediate_dense.bias"] = (I,)
            out[p + "ff1.1.output_dense.weight"] = (d, I); out[p + "ff1.1.output_dense.bias"] = (d,)
        out[p + "self_attn_layer_norm.weight"] = (d,); out[p + "self_attn_layer_norm.bias"] = (d,)
        if ptype == "relative":
            out[p + "self_attn.pos_bias_u"] = (H, d // H); out[p + "self_attn.pos_bias_v"] = (H, d // H)
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            out[p + f"self_attn.{n}.weight"] = (d, d); out[p + f"self_attn.{n}.bias"] = (d,)
        if ptype == "relative":
            out[p + "self_attn.linear_pos.weight"] = (d, d)
        out[p + "cgMLP.channel_proj1.0.weight"] = (I, d); out[p + "cgMLP.channel_proj1.0.bias"] = (I,)
        out[p + "cgMLP.csgu.norm.weight"] = (I // 2,); out[p + "cgMLP.csgu.norm.bias"] = (I // 2,)
        out[p + "cgMLP.csgu.conv.weight"] = (I // 2, 1, kc); out[p + "cgMLP.csgu.conv.bias"] = (I // 2,)
        if cfg.get("csgu_use_linear_after_conv", False):
            out[p + "cgMLP.csgu.linear.weight"] = (I // 2, I // 2); out[p + "cgMLP.csgu.linear.bias"] = (I // 2,)
        out[p + "cgMLP.channel_proj2.weight"] = (d, I // 2); out[p + "cgMLP.channel_proj2.bias"] = (d,)
        out[p + "cgMLP_layer_norm.weight"] = (d,); out[p + "cgMLP_layer_norm.bias"] = (d,)
        out[p + "merge_proj.weight"] = (d, 2 * d); out[p + "merge_proj.bias"] = (d,)
        out[p + "depthwise_conv_fusion.weight"] = (2 * d, 1, km); out[p + "depthwise_conv_fusion.bias"] = (2 * d,)
        out[p + "final_layer_norm.weight"] = (d,); out[p + "final_layer_norm.bias"] = (d,)
        if "ff2" in ffs:
            out[p + "ff2.0.weight"] = (d,); out[p + "ff2.0.bias"] = (d,)
            out[p + "ff2.1.intermediate_dense.weight"] = (I, d); out[p + "ff2.1.intermediate_dense.bias"] = (I,)
            out[p + "ff2.1.output_dense.weight"] = (d, I); out[p + "ff2.1.output_dense.bias"] = (d,)
    out["lm_head.weight"] = (V, d); out["lm_head.bias"] = (V,)
    out["blank_projection.weight"] = (1, d); out["blank_projection.bias"] = (1,)
    return out


TINY = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
            conv_dim=[32, 32], conv_kernel=[3, 3], conv_stride=[2, 2], conv_padding=[1, 1], vocab_size=50, num_fbanks=80)
SMALL = dict(hidden_size=256, num_hidden_layers=12, num_attention_heads=4, intermediate_size=1024,
             conv_dim=[256, 256], conv_kernel=[3, 3], conv_stride=[2, 2], conv_padding=[1, 1], vocab_size=5000, num_fbanks=80)
BASE = dict(hidden_size=512, num_hidden_layers=16, num_attention_heads=4, intermediate_size=2048,
            conv_dim=[256, 256], conv_kernel=[3, 3], conv_stride=[2, 2], conv_padding=[1, 1], vocab_size=5000, num_fbanks=80)
