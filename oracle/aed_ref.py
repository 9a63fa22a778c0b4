"""ORACLE (test infrastructure, never shipped): CPU restatement of the joint CTC/attention encoder-decoder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Functional torch-CPU fp32 restatement of
  * reference `src/models/ctc_encoder_plus_autoregressive_decoder.py:237-358` (JointCTCAttentionEncoderDecoder.forward:
    encoder with labels, enc_to_dec_proj, cross mask from the OUTER length formula, shift_tokens_right, loss mix),
  * reference `src/models/decoders/multi_head_gpt2.py:80-170` (GPT2LMMultiHeadModel: auxiliary heads on intermediate
    hidden states, its own shifted label-smoothed CE -> the "double shift" of SURVEY.md §8a row 16),
  * reference `src/models/embeddings.py:33-86` + `auto_wrappers.py:186-209` (fixed sinusoidal positions, scaled embedding),
  * the GPT-2 block it inherits from transformers (pinned 4.39.3; installed 5.15.0 `models/gpt2/modeling_gpt2.py`:
    eager attention :54-72, block :262-309, MLP with gelu_new, Conv1D weights stored (in, out)).
Pinned by tests/golden/aed_*.npz (made from the imported reference by tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from . import ebranchformer_ref as E


def _id(x):
    return x


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def shift_tokens_right(labels: torch.Tensor, pad_id: int, start_id: int) -> torch.Tensor:
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1].clone()
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def conv1d(x, w, b):          # transformers Conv1D: weight (in, out)
    return x @ w + b


def mha(q, k, v, H, add_mask, drop=None):
    """drop(probs in the kernels' head-major (H,B,Tq,Tk) layout) -> dropped probs (GPT-2 attn_dropout, train mode)"""
    B, Tq, d = q.shape
    hd = d // H
    qh = q.view(B, Tq, H, hd).transpose(1, 2)
    kh = k.view(B, -1, H, hd).transpose(1, 2)
    vh = v.view(B, -1, H, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    if add_mask is not None:
        s = s + add_mask
    pr = torch.softmax(s, -1)
    if drop is not None:
        pr = drop(pr.transpose(0, 1)).transpose(0, 1)
    return (pr @ vh).transpose(1, 2).reshape(B, Tq, d)


def embed(sd, pre, cfg, ids, past=0):
    d = cfg["n_embd"]
    pos = torch.arange(past, past + ids.shape[1])
    if cfg.get("pos_emb_fixed", False):
        tok = sd[pre + "transformer.wte.emb_layers.0.weight"][ids] * (d ** 0.5)          # embeddings.py:33-62
        inv = 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d))                              # embeddings.py:66-86
        sin = torch.outer(pos.float(), inv)
        pe = torch.cat([sin.sin(), sin.cos()], dim=-1)
    else:
        tok = sd[pre + "transformer.wte.weight"][ids]
        pe = sd[pre + "transformer.wpe.weight"][pos]
    return tok + pe[None]


def decoder_hidden_states(sd, pre, cfg, ids, enc, enc_mask, q: Callable = _id, dm=None):
    """GPT2Model.forward with cross-attention, no cache -> list of hidden states as HF returns them:
    [embeddings, block_0 out, ..., block_{L-2} out, ln_f(block_{L-1} out)].
    dm(x, layer, site): train-mode dropout hook; decoder layer l uses layer index 32 + l with sites 0 self-attn probs, 1 self-attn output,
    2 cross-attn probs, 3 cross-attn output, 4 MLP output; the embedding dropout is (63, 0)."""
    d, H, L = cfg["n_embd"], cfg["n_head"], cfg["n_layer"]
    eps = cfg.get("layer_norm_epsilon", 1e-5)
    x = embed(sd, pre, cfg, ids)
    if dm is not None:
        x = dm(x, 63, 0)
    dr = (lambda lay, site: (lambda t: dm(t, lay, site))) if dm is not None else (lambda lay, site: None)
    dz = (lambda t, lay, site: dm(t, lay, site)) if dm is not None else (lambda t, lay, site: t)
    B, U, _ = x.shape
    fmin = torch.finfo(torch.float32).min
    causal = torch.ones(U, U, dtype=torch.bool).triu(1)
    self_mask = torch.zeros(U, U).masked_fill(causal, fmin)[None, None]
    cross_mask = None
    if enc_mask is not None:
        cross_mask = torch.zeros(B, 1, 1, enc.shape[1]).masked_fill(~enc_mask[:, None, None, :], fmin)
    hs = [x]
    for l in range(L):
        p = f"{pre}transformer.h.{l}."
        h = q(E.layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps))
        qkv = q(conv1d(h, q(sd[p + "attn.c_attn.weight"]), sd[p + "attn.c_attn.bias"]))
        a = q(mha(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], H, self_mask, dr(32 + l, 0)))
        x = x + dz(conv1d(a, q(sd[p + "attn.c_proj.weight"]), sd[p + "attn.c_proj.bias"]), 32 + l, 1)
        h = q(E.layer_norm(x, sd[p + "ln_cross_attn.weight"], sd[p + "ln_cross_attn.bias"], eps))
        qq = q(conv1d(h, q(sd[p + "crossattention.q_attn.weight"]), sd[p + "crossattention.q_attn.bias"]))
        kv = q(conv1d(q(enc), q(sd[p + "crossattention.c_attn.weight"]), sd[p + "crossattention.c_attn.bias"]))
        a = q(mha(qq, kv[..., :d], kv[..., d:], H, cross_mask, dr(32 + l, 2)))
        x = x + dz(conv1d(a, q(sd[p + "crossattention.c_proj.weight"]), sd[p + "crossattention.c_proj.bias"]), 32 + l, 3)
        h = q(E.layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps))
        m = q(gelu_new(conv1d(h, q(sd[p + "mlp.c_fc.weight"]), sd[p + "mlp.c_fc.bias"])))
        x = x + dz(conv1d(m, q(sd[p + "mlp.c_proj.weight"]), sd[p + "mlp.c_proj.bias"]), 32 + l, 4)
        if l + 1 < L:
            hs.append(x)
    hs.append(E.layer_norm(x, sd[pre + "transformer.ln_f.weight"], sd[pre + "transformer.ln_f.bias"], eps))
    return hs


def smoothed_ce(logits: torch.Tensor, target: torch.Tensor, eps: float) -> torch.Tensor:
    """torch CrossEntropyLoss(label_smoothing=eps), ignore_index=-100, mean over valid targets."""
    lp = torch.log_softmax(logits.float(), -1)
    valid = target >= 0
    t = target.clamp(min=0)
    nll = -lp.gather(-1, t[..., None])[..., 0]
    smooth = -lp.mean(-1)
    per = (1 - eps) * nll + eps * smooth
    return (per * valid).sum() / valid.sum()


def decoder_forward(sd, pre, cfg, ids, enc, enc_mask, labels=None, q: Callable = _id, dm=None):
    """GPT2LMMultiHeadModel.forward (multi_head_gpt2.py:80-170) -> (loss|None, logits of the last head)."""
    hs = decoder_hidden_states(sd, pre, cfg, ids, enc, enc_mask, q, dm)
    logits = F.linear(q(hs[-1]), q(sd[pre + "lm_head.weight"]))
    loss = None
    if labels is not None:
        locs = list(cfg.get("head_locations") or [])
        weights = list(cfg.get("head_weights") or [1.0])
        loss = torch.tensor(0.0)
        for k, (idx, w) in enumerate(zip(locs + [-1], weights)):
            hw = sd[pre + "lm_head.weight"] if idx == -1 else sd[f"{pre}additional_lm_heads.{k}.weight"]
            lg = F.linear(q(hs[idx]), q(hw))
            loss = loss + w * smoothed_ce(lg[:, :-1].reshape(-1, lg.shape[-1]), labels[:, 1:].reshape(-1), cfg.get("lsm_factor", 0.0))
    return loss, logits


def joint_forward(sd, enc_cfg, dec_cfg, jcfg, feats, attention_mask, labels, q: Optional[Callable] = None, dm=None):
    """JointCTCAttentionEncoderDecoder.forward (eval mode, or train mode with the dropout hook dm) -> dict(loss, enc_loss, dec_loss, logits, encoder_logits)."""
    qq = q or _id
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    hidden = E.encoder_forward(esd, enc_cfg, feats, attention_mask, q, dm=dm)
    enc_logits = E.ctc_head(esd, hidden, q, dm, enc_cfg["num_hidden_layers"])
    am = attention_mask if attention_mask is not None else torch.ones(feats.shape[:2], dtype=torch.long)
    outer = E.conv_out_lengths_outer(am.sum(-1), enc_cfg).long()
    enc_loss = None
    if labels is not None:
        lmask = labels >= 0
        enc_loss = E.ctc_loss_ref(torch.log_softmax(enc_logits.float(), -1), labels, outer, lmask.sum(-1), blank=enc_logits.shape[-1] - 1,
                                  reduction=enc_cfg.get("ctc_loss_reduction", "mean"), zero_infinity=enc_cfg.get("ctc_zero_infinity", False))
    enc_h = hidden
    if "enc_to_dec_proj.weight" in sd:                                     # :289-293
        enc_h = F.linear(qq(hidden), qq(sd["enc_to_dec_proj.weight"]), sd["enc_to_dec_proj.bias"])
    enc_mask = (torch.arange(hidden.shape[1])[None, :] < outer[:, None]) if attention_mask is not None else None   # :296-301 (outer lengths)
    dec_ids = shift_tokens_right(labels, jcfg["pad_token_id"], jcfg["decoder_start_token_id"])   # :303-304
    dec_loss, logits = decoder_forward(sd, "decoder.", dec_cfg, dec_ids, enc_h, enc_mask, labels, qq, dm)
    w = jcfg["ctc_weight"]
    loss = w * enc_loss + (1 - w) * dec_loss if labels is not None else None
    return dict(loss=loss, enc_loss=enc_loss, dec_loss=dec_loss, logits=logits, encoder_logits=enc_logits, encoder_hidden=enc_h)
