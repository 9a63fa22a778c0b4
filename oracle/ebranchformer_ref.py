"""ORACLE (test infrastructure, never shipped): CPU restatement of the E-Branchformer encoder + CTC path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

A functional, torch-CPU fp32 restatement (no nn.Module, no transformers import) of
  * reference `src/models/encoders/e_branchformer.py` (layer :263-313, attention :74-141,
    cgMLP/CSGU :144-222, CTC head :422-496),
  * reference `src/models/extractors.py` (Conv2d sub-sampling :68-113, lengths :133-162),
  * the bodies it inherits from transformers' wav2vec2_conformer (pinned 4.39.3; line numbers
    below are those of the installed 5.15.0, `tf:` = models/wav2vec2_conformer/modeling_wav2vec2_conformer.py):
    rel-pos table tf:159-205, rotary tf:125-156, rel-shift tf:528-565, rotary apply tf:509-526,
    FFN tf:350-357, feature projection tf:328-333, encoder loop tf:651-717, mask tf:917-935.
Pinned against the imported reference by tests/golden/*.npz (see tests/golden/make_golden.py).

`q` is an optional rounding hook used to *model* the GPU path's storage precision (bf16 at the
points where the HIP kernels store bf16); with q=None this is the plain fp32 algorithm.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn.functional as F


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def _id(x):
    return x


# ----------------------------------------------------------------------------- lengths
def conv_out_lengths_inner(lengths: torch.Tensor, cfg: dict) -> torch.Tensor:
    """extractors.py:133-162 (CustomFE, used for the encoder attention mask): padded formula."""
    out = lengths.clone()
    for k, s, p in zip(cfg["conv_kernel"], cfg["conv_stride"], cfg["conv_padding"]):
        out = torch.div(out + ((k - 1) if cfg.get("is_causal", False) else 2 * p) - k, s, rounding_mode="floor") + 1
    return out


def conv_out_lengths_outer(lengths: torch.Tensor, cfg: dict) -> torch.Tensor:
    """Wav2Vec2ForCTC._get_feat_extract_output_lengths (un-padded formula) — the quirk of
    SURVEY.md §8a row 8': used for CTC input_lengths at e_branchformer.py:468."""
    out = lengths.clone()
    for k, s in zip(cfg["conv_kernel"], cfg["conv_stride"]):
        out = torch.div(out - k, s, rounding_mode="floor") + 1
    return out


def feature_vector_attention_mask(t_out: int, attention_mask: torch.Tensor, cfg: dict) -> torch.Tensor:
    """tf:917-935 — True for the first L_out frames."""
    lens = conv_out_lengths_inner(attention_mask.sum(-1), cfg).long()
    return torch.arange(t_out)[None, :] < lens[:, None]


# ----------------------------------------------------------------------------- positions
def rel_pos_table(t: int, d: int) -> torch.Tensor:
    """tf:159-205 restricted to the (2t-1) rows used for a length-t input: rows are relative
    positions t-1, ..., 0, ..., -(t-1); even cols sin, odd cols cos."""
    pos = torch.arange(t - 1, -t, -1, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.int64).float() * -(math.log(10000.0) / d))
    pe = torch.zeros(2 * t - 1, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def rotary_table(t: int, head: int, base: float = 10000.0):
    """tf:125-156 -> cos, sin of shape (t, head)."""
    inv = 1.0 / (base ** (torch.arange(0, head, 2, dtype=torch.int64).float() / head))
    fr = torch.einsum("i,j->ij", torch.arange(t).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def apply_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int) -> torch.Tensor:
    """tf:509-526 — rotate the *input* of the Q/K projections, per head."""
    b, t, d = x.shape
    hd = d // heads
    xh = x.view(b, t, heads, hd)
    rot = torch.cat((-xh[..., hd // 2:], xh[..., : hd // 2]), dim=-1)
    return (xh * cos[None, :, None, :] + rot * sin[None, :, None, :]).reshape(b, t, d)


# ----------------------------------------------------------------------------- blocks
def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def conv_subsample(sd: dict, cfg: dict, feats: torch.Tensor, q: Callable = _id) -> torch.Tensor:
    """extractors.py:110-113 (+ causal left padding streaming_modules.py:31-55; context-aware gates extractors.py:23-65) -> (B, T', d)."""
    p = "wav2vec2.feature_extractor."
    h = feats[:, None]
    # extractors.py:59-61: a dict lookup with nn.Conv2d as the default — only the exact strings "gated" / "gated_shared" select a gate; the causal stack never looks
    ctx = None if cfg.get("is_causal", False) else {"gated": 1, "gated_shared": 4}.get(cfg.get("context_awareness_type"))
    for i, (k, s, pad) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"], cfg["conv_padding"])):
        cw = "" if cfg.get("is_causal", False) else ".conv"   # CausalConv2d is the Conv2d itself
        qw = q if i > 0 else _id
        if ctx is not None:
            w, b = sd[f"{p}conv.{i}.0.conv.conv.weight"], sd[f"{p}conv.{i}.0.conv.conv.bias"]
            gw, gb = sd[f"{p}conv.{i}.0.conv.gate.weight"], sd[f"{p}conv.{i}.0.conv.gate.bias"]
            c = F.conv2d(h, qw(w), b, stride=s, padding=pad)
            if ctx == 1:                                      # GatedConv2d.forward (extractors.py:31-32)
                h = c * torch.sigmoid(F.conv2d(h, qw(gw), gb, stride=s, padding=pad))
            else:                                             # GatedConv2dShared.forward (extractors.py:49-54): raises where the reference's view / broadcast does
                g = torch.sigmoid(F.conv2d(h, qw(gw), gb, stride=(s * ctx, s), padding=(pad * ctx, pad)))
                h = (c.view(*c.size()[:2], -1, ctx, c.size(3)) * g.unsqueeze(3)).view(c.size())
            h = q(F.gelu(h))
            continue
        w, b = sd[f"{p}conv.{i}.0{cw}.weight"], sd[f"{p}conv.{i}.0{cw}.bias"]
        w = qw(w)
        if cfg.get("is_causal", False):
            lp = pad * 2
            h = F.conv2d(F.pad(h, (lp, 0, lp, 0)), w, b, stride=s)
        else:
            h = F.conv2d(h, w, b, stride=s, padding=pad)
        h = q(F.gelu(h))
    h = h.transpose(1, 2).flatten(2, 3)                       # (B, T', C*F')
    return F.linear(h, q(sd[p + "out.weight"]), sd[p + "out.bias"])


def rel_attention_scores(qh, kh, pos_proj, u, v, q: Callable = _id):
    """tf:528-565 restated as the direct index map bd[i,j] = (q_i+v)·p[T-1-i+j].
    qh,kh: (B,H,T,hd); pos_proj: (H, 2T-1, hd)."""
    t = qh.shape[2]
    ac = torch.matmul(q(qh + u[None, :, None, :]), kh.transpose(-2, -1))
    bd_full = torch.matmul(q(qh + v[None, :, None, :]), pos_proj.transpose(-2, -1)[None])   # (B,H,T,2T-1)
    idx = (t - 1) - torch.arange(t)[:, None] + torch.arange(t)[None, :]
    bd = torch.gather(bd_full, 3, idx[None, None].expand(qh.shape[0], qh.shape[1], t, t))
    return (ac + bd) / math.sqrt(qh.shape[-1])


def self_attention(sd, pre, cfg, x, add_mask, pos, q: Callable = _id, dm=None, layer=None):
    """e_branchformer.py:74-141.  x: LN output (B,T,d).  dm(x, layer, site): training-mode dropout hook (site 2 = probabilities, :132)."""
    b, t, d = x.shape
    H = cfg["num_attention_heads"]
    hd = d // H
    ptype = cfg.get("position_embeddings_type", "relative")
    xq = x
    if ptype == "rotary":
        xq = q(apply_rotary(x, pos[0], pos[1], H))
    lin = lambda n, inp: F.linear(inp, q(sd[f"{pre}{n}.weight"]), sd[f"{pre}{n}.bias"])
    qh = q(lin("linear_q", xq)).view(b, t, H, hd).transpose(1, 2)
    kh = q(lin("linear_k", xq)).view(b, t, H, hd).transpose(1, 2)
    vh = q(lin("linear_v", x)).view(b, t, H, hd).transpose(1, 2)
    if ptype == "relative":
        pp = q(F.linear(q(pos), q(sd[pre + "linear_pos.weight"]))).view(-1, H, hd).transpose(0, 1)
        scores = rel_attention_scores(qh, kh, pp, sd[pre + "pos_bias_u"], sd[pre + "pos_bias_v"], q)
    else:
        scores = torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(hd)
    if cfg.get("is_causal", False):
        causal = torch.ones(t, t, dtype=torch.bool).triu(1)
        fmin = torch.finfo(torch.float32).min
        add_mask = (causal * fmin)[None, None] if add_mask is None else add_mask.masked_fill(causal, fmin)
    if add_mask is not None:
        scores = scores + add_mask
    if q is _id:
        probs = torch.softmax(scores, dim=-1)
        if dm is not None:                      # mask indexed over the head-major (H,B,T,T) layout of the kernels
            probs = dm(probs.transpose(0, 1), layer, 2).transpose(0, 1)
        ctx = torch.matmul(probs, vh)
    else:  # precision model of the fused kernel: un-normalised p rounded for the PV product
        m = scores.max(dim=-1, keepdim=True).values
        p = torch.exp(scores - m)
        ctx = torch.matmul(q(p), vh) / p.sum(dim=-1, keepdim=True)
    ctx = q(ctx.transpose(1, 2).reshape(b, t, d))
    return lin("linear_out", ctx)


def dwconv1d(x, w, b, causal=False, dilation=1):
    """depthwise Conv1d over time on (B,T,C); w (C,1,K).  The causal form is CausalConv1d
    (streaming_modules.py:12-28): left pad (K-1)*dilation, no right pad."""
    k = w.shape[-1]
    xt = x.transpose(1, 2)
    if causal:
        y = F.conv1d(F.pad(xt, ((k - 1) * dilation, 0)), w, b, groups=w.shape[0], dilation=dilation)
    else:
        y = F.conv1d(xt, w, b, padding=(k - 1) // 2, groups=w.shape[0])
    return y.transpose(1, 2)


def cgmlp(sd, pre, cfg, x, q: Callable = _id, dm=None, layer=None):
    """e_branchformer.py:184-222.  x: LN output.  dropout site 4 = after the gating (:203)."""
    h = q(F.gelu(F.linear(x, q(sd[pre + "channel_proj1.0.weight"]), sd[pre + "channel_proj1.0.bias"])))
    r, g = h.chunk(2, dim=-1)
    g = layer_norm(g, sd[pre + "csgu.norm.weight"], sd[pre + "csgu.norm.bias"])
    # Quirk: e_branchformer.py:153-160 passes (K-1)//2 in CausalConv1d's *dilation* slot, so the
    # causal CSGU conv is dilated by 15 (left pad 450); reproduced, not "fixed".
    kcs = sd[pre + "csgu.conv.weight"].shape[-1]
    g = dwconv1d(g, sd[pre + "csgu.conv.weight"], sd[pre + "csgu.conv.bias"], cfg.get("is_causal", False), (kcs - 1) // 2)
    if cfg.get("csgu_use_linear_after_conv", False):
        g = F.linear(g, sd[pre + "csgu.linear.weight"], sd[pre + "csgu.linear.bias"])
    act = cfg.get("csgu_activation", "identity")
    if act != "identity":
        g = {"gelu": F.gelu, "relu": F.relu, "silu": F.silu, "swish": F.silu}[act](g)
    s = q(r * g)
    if dm is not None:
        s = dm(s, layer, 4)
    return F.linear(s, q(sd[pre + "channel_proj2.weight"]), sd[pre + "channel_proj2.bias"])


def ffn(sd, pre, x, q: Callable = _id, dm=None, layer=None, sites=(0, 1)):
    """tf:350-357 (hidden_act gelu); dropout after the activation (:353) and after the output projection (:356)."""
    h = q(F.gelu(F.linear(x, q(sd[pre + "intermediate_dense.weight"]), sd[pre + "intermediate_dense.bias"])))
    if dm is not None:
        h = dm(h, layer, sites[0])
    y = F.linear(h, q(sd[pre + "output_dense.weight"]), sd[pre + "output_dense.bias"])
    return dm(y, layer, sites[1]) if dm is not None else y


def encoder_layer(sd, i, cfg, x, add_mask, pos, q: Callable = _id, dm=None, pre=None):
    """e_branchformer.py:263-313.  dm: training-mode dropout hook; sites 0/1 ff1, 2 attention probabilities, 3 attention output (:288),
    4 CSGU, 5 merge output (:301), 6/7 ff2.  `pre`: state-dict prefix when the layer is not one of the encoder's own
    (the fine-tuning head's `additional_layer.`, bestrq.py:199-200)."""
    pre = pre or f"wav2vec2.encoder.layers.{i}."
    eps = 1e-5  # nn.LayerNorm default (layers use nn.LayerNorm(embed_dim), e_branchformer.py:233-261)
    if cfg.get("use_macaron_ff", True):
        x = x + 0.5 * ffn(sd, pre + "ff1.1.", q(layer_norm(x, sd[pre + "ff1.0.weight"], sd[pre + "ff1.0.bias"], eps)), q, dm, i, (0, 1))
    res = x
    g = self_attention(sd, pre + "self_attn.", cfg,
                       q(layer_norm(x, sd[pre + "self_attn_layer_norm.weight"], sd[pre + "self_attn_layer_norm.bias"], eps)),
                       add_mask, pos, q, dm, i)
    if dm is not None:
        g = dm(g, i, 3)
    l = cgmlp(sd, pre + "cgMLP.", cfg,
              q(layer_norm(x, sd[pre + "cgMLP_layer_norm.weight"], sd[pre + "cgMLP_layer_norm.bias"], eps)), q, dm, i)
    m = q(torch.cat([g, l], dim=-1))
    m = q(m + dwconv1d(m, sd[pre + "depthwise_conv_fusion.weight"], sd[pre + "depthwise_conv_fusion.bias"]))
    mo = F.linear(m, q(sd[pre + "merge_proj.weight"]), sd[pre + "merge_proj.bias"])
    x = res + (dm(mo, i, 5) if dm is not None else mo)
    if cfg.get("use_macaron_ff", True):
        x = x + 0.5 * ffn(sd, pre + "ff2.1.", q(layer_norm(x, sd[pre + "ff2.0.weight"], sd[pre + "ff2.0.bias"], eps)), q, dm, i, (6, 7))
    return layer_norm(x, sd[pre + "final_layer_norm.weight"], sd[pre + "final_layer_norm.bias"], eps)


def encoder_forward(sd: dict, cfg: dict, feats: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                    q: Optional[Callable] = None, return_layers: bool = False, dm=None, skip_layers=(), hidden_states: Optional[list] = None):
    """Wav2Vec2EBranchformerModel.forward (tf:1133-1195, tf:651-717) -> last hidden state (B,T',d).  Eval mode, or train mode with the
    dropout hook dm(x, layer, site) (global sites use layer = num_hidden_layers: 0 feature projection, 1 encoder input tf:674).
    skip_layers: LayerDrop decisions (tf:686-690: a layer whose uniform draw falls below config.layerdrop is skipped = identity).
    hidden_states: a list that receives HF's `output_hidden_states` tuple (tf:679-680,714-715): the INPUT of every layer, then the output of
    the encoder's final LayerNorm — num_hidden_layers + 1 tensors."""
    q = q or _id
    eps = cfg.get("layer_norm_eps", 1e-5)
    h = conv_subsample(sd, cfg, feats, q)
    t = h.shape[1]
    mask = feature_vector_attention_mask(t, attention_mask, cfg) if attention_mask is not None else None
    p = "wav2vec2.feature_projection."
    h = q(layer_norm(h, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], eps))
    x = F.linear(h, q(sd[p + "projection.weight"]), sd[p + "projection.bias"])
    nl = cfg["num_hidden_layers"]
    if dm is not None:
        x = dm(x, nl, 0)
    add_mask = None
    if mask is not None:
        x = x * mask[..., None]                                  # tf:662-665 zero padded frames once
        am = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min   # tf:667-672
        add_mask = am.expand(-1, 1, t, -1)
    d, H = cfg["hidden_size"], cfg["num_attention_heads"]
    ptype = cfg.get("position_embeddings_type", "relative")
    pos = rel_pos_table(t, d) if ptype == "relative" else (rotary_table(t, d // H, cfg.get("rotary_embedding_base", 10000)) if ptype == "rotary" else None)
    if dm is not None:
        x = dm(x, nl, 1)
    layers = []
    for i in range(cfg["num_hidden_layers"]):
        if hidden_states is not None:
            hidden_states.append(x)
        if i not in skip_layers:
            x = encoder_layer(sd, i, cfg, x, add_mask, pos, q, dm)
        if return_layers:
            layers.append(x)
    x = layer_norm(x, sd["wav2vec2.encoder.layer_norm.weight"], sd["wav2vec2.encoder.layer_norm.bias"], eps)
    if hidden_states is not None:
        hidden_states.append(x)
    return (x, layers) if return_layers else x


def finetune_hidden(sd: dict, cfg: dict, feats: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, q: Optional[Callable] = None, dm=None,
                    skip_layers=()):
    """What BestRQEBranchformerForCTC puts between the encoder and the CTC head (bestrq.py:229-279): with `finetune_with_layer_mixing` the
    softmax(per_layer_weights)-weighted sum of all num_hidden_layers + 1 hidden states instead of the last one (:239-245); with
    `finetune_with_additional_layer` one more E-Branchformer layer (`additional_layer.*`) on top, padded frames zeroed first, same key mask and
    the encoder's position table (:247-274) — and NO LayerNorm after it."""
    q = q or _id
    hs: list = []
    x = encoder_forward(sd, cfg, feats, attention_mask, q, dm=dm, skip_layers=skip_layers, hidden_states=hs)
    if cfg.get("finetune_with_layer_mixing", False):
        w = torch.softmax(sd["per_layer_weights"].float(), dim=-1)
        x = (torch.stack(hs) * w[:, None, None, None]).sum(dim=0)
    if cfg.get("finetune_with_additional_layer", False):
        t = x.shape[1]
        d, H = cfg["hidden_size"], cfg["num_attention_heads"]
        ptype = cfg.get("position_embeddings_type", "relative")
        pos = rel_pos_table(t, d) if ptype == "relative" else (rotary_table(t, d // H, cfg.get("rotary_embedding_base", 10000)) if ptype == "rotary" else None)
        add_mask = None
        if attention_mask is not None:
            mask = feature_vector_attention_mask(t, attention_mask, cfg)
            x = x * mask[..., None]                                                      # bestrq.py:260
            am = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
            add_mask = am.expand(-1, 1, t, -1)
        # dropout-hook layer id: num_hidden_layers names the global sites, so the additional layer takes num_hidden_layers + 1
        x = encoder_layer(sd, cfg["num_hidden_layers"] + 1, cfg, x, add_mask, pos, q, dm, pre="additional_layer.")
    return x


def finetune_ctc_forward(sd: dict, cfg: dict, feats, attention_mask=None, labels=None, q=None, dm=None, skip_layers=()):
    """BestRQEBranchformerForCTC.forward (bestrq.py:212-322) -> (loss|None, logits); the head and the loss are those of ctc_forward."""
    hidden = finetune_hidden(sd, cfg, feats, attention_mask, q, dm, skip_layers)
    logits = ctc_head(sd, hidden, q, dm, cfg["num_hidden_layers"])
    loss = None
    if labels is not None:
        am = attention_mask if attention_mask is not None else torch.ones(feats.shape[:2], dtype=torch.long)
        in_len = conv_out_lengths_outer(am.sum(-1), cfg).long()
        lmask = labels >= 0
        # the reference flattens `labels.masked_select(labels >= 0)` (e_branchformer.py:472-475): the valid ids of a row count wherever the -100s sit
        # (the collator's `mask_unks` puts them mid-row, collators.py:97-98) -> compact them to the front of the row
        packed = torch.full_like(labels, -100)
        for b in range(labels.shape[0]):
            v = labels[b][lmask[b]]
            packed[b, : v.numel()] = v
        labels = packed
        loss = ctc_loss_ref(torch.log_softmax(logits.float(), -1), labels, in_len, lmask.sum(-1), blank=logits.shape[-1] - 1,
                            reduction=cfg.get("ctc_loss_reduction", "mean"), zero_infinity=cfg.get("ctc_zero_infinity", False))
    return loss, logits


def ctc_head(sd: dict, hidden: torch.Tensor, q: Optional[Callable] = None, dm=None, nl=None) -> torch.Tensor:
    """e_branchformer.py:451-457 — dropout (train mode; global site 2), lm_head ⊕ blank_projection, blank is the LAST class."""
    q = q or _id
    if dm is not None:
        hidden = dm(hidden, nl, 2)
    w = torch.cat([sd["lm_head.weight"], sd["blank_projection.weight"]], 0)
    b = torch.cat([sd["lm_head.bias"], sd["blank_projection.bias"]], 0)
    return F.linear(q(hidden), q(w), b)


def ctc_forward(sd: dict, cfg: dict, feats, attention_mask=None, labels=None, q=None):
    """Wav2Vec2EBranchformerForCTC.forward (e_branchformer.py:422-496), eval mode -> (loss|None, logits)."""
    hidden = encoder_forward(sd, cfg, feats, attention_mask, q)
    logits = ctc_head(sd, hidden, q)
    loss = None
    if labels is not None:
        if labels.max() >= cfg["vocab_size"]:
            raise ValueError(f"Label values must be <= vocab_size: {cfg['vocab_size']}")
        am = attention_mask if attention_mask is not None else torch.ones(feats.shape[:2], dtype=torch.long)
        in_len = conv_out_lengths_outer(am.sum(-1), cfg).long()
        lmask = labels >= 0
        # the reference flattens `labels.masked_select(labels >= 0)` (e_branchformer.py:472-475): the valid ids of a row count wherever the -100s sit
        # (the collator's `mask_unks` puts them mid-row, collators.py:97-98) -> compact them to the front of the row
        packed = torch.full_like(labels, -100)
        for b in range(labels.shape[0]):
            v = labels[b][lmask[b]]
            packed[b, : v.numel()] = v
        labels = packed
        loss = ctc_loss_ref(torch.log_softmax(logits.float(), -1), labels, in_len, lmask.sum(-1),
                            blank=logits.shape[-1] - 1, reduction=cfg.get("ctc_loss_reduction", "mean"),
                            zero_infinity=cfg.get("ctc_zero_infinity", False))
    return loss, logits


# ----------------------------------------------------------------------------- CTC loss
def ctc_nll_ref(log_probs: torch.Tensor, labels: torch.Tensor, in_len: torch.Tensor, tgt_len: torch.Tensor, blank: int):
    """Per-utterance negative log-likelihood by the alpha recursion (Graves 2006), float64.
    log_probs (B,T,V) log-softmaxed; labels (B,U) with padding (<0) AFTER the tgt_len valid ids."""
    B = log_probs.shape[0]
    out = torch.zeros(B, dtype=torch.float64)
    lp = log_probs.double()
    for b in range(B):
        T, U = int(in_len[b]), int(tgt_len[b])
        lab = labels[b][labels[b] >= 0][:U]
        ext = torch.full((2 * U + 1,), blank, dtype=torch.long)
        ext[1::2] = lab
        S = 2 * U + 1
        neg = float("-inf")
        a = torch.full((S,), neg, dtype=torch.float64)
        if T == 0:
            out[b] = 0.0 if U == 0 else float("inf")
            continue
        a[0] = lp[b, 0, blank]
        if S > 1:
            a[1] = lp[b, 0, ext[1]]
        skip_ok = torch.zeros(S, dtype=torch.bool)
        if S > 2:
            skip_ok[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        for t in range(1, T):
            a1 = torch.cat([torch.tensor([neg], dtype=torch.float64), a])[:S]
            a2 = torch.cat([torch.tensor([neg, neg], dtype=torch.float64), a])[:S]
            a2 = torch.where(skip_ok, a2, torch.full_like(a2, neg))
            a = torch.logsumexp(torch.stack([a, a1, a2]), 0) + lp[b, t, ext]
        ll = torch.logsumexp(a[-2:], 0) if S > 1 else a[-1]
        out[b] = -ll
    return out


def ctc_loss_ref(log_probs, labels, in_len, tgt_len, blank, reduction="mean", zero_infinity=False):
    """Semantics of torch.nn.functional.ctc_loss as called at e_branchformer.py:480-488.
    log_probs here is (B,T,V) (the reference transposes to (T,B,V) for aten)."""
    nll = ctc_nll_ref(log_probs, labels, in_len, tgt_len, blank)
    if zero_infinity:
        nll = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)
    if reduction == "mean":
        return (nll / tgt_len.clamp(min=1).double()).mean().float()
    if reduction == "sum":
        return nll.sum().float()
    return nll.float()
