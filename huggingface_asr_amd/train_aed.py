"""Training step of the joint CTC/attention encoder-decoder on the HIP path (BASELINE config 3).

Differentiates what `huggingface_asr_amd.decoder.JointAEDEngine.forward` computes, i.e. the reference's
`JointCTCAttentionEncoderDecoder.forward` (src/models/ctc_encoder_plus_autoregressive_decoder.py:237-358) with its
`GPT2LMMultiHeadModel` decoder (src/models/decoders/multi_head_gpt2.py:80-170: auxiliary heads, shifted label-smoothed CE) and
the transformers GPT-2 block (ln_1 -> causal self-attention -> ln_cross_attn -> cross-attention over the encoder frames -> ln_2 ->
gelu_new MLP):   loss = w * CTC + (1 - w) * sum_k head_weight_k * CE_k.

The encoder side is `train.EncoderCTCTrainer`; the decoder hooks into its backward at the encoder output (`extra_hidden_grad`),
so, with `GradSync(overlap=True)`, the encoder's per-layer gradient all-reduces still overlap the remaining backward.  Same precision model and the same
restrictions as train.py; GPT-2's embd / attn / resid dropouts use the same counter-based masks (decoder layer l = stream layer 32 + l).
"""
from __future__ import annotations

import math

import torch

from . import ops
from . import ops_train as T
from .decoder import shift_tokens_right
from .train import BF16, F32, EncoderCTCTrainer, GradSync, ParamStore, Spec


def decoder_specs(c: dict, enc_dim: int, with_proj: bool) -> list[Spec]:
    d, L, V = c["n_embd"], c["n_layer"], c["vocab_size"]
    S = []
    mat = lambda n, *sh: S.append(Spec(n, tuple(sh), True, True))
    vec = lambda n, *sh, decay=False: S.append(Spec(n, tuple(sh), False, decay))
    if with_proj:
        mat("proj_w", d, enc_dim); vec("proj_b", d)
    mat("wte", V, d)                                       # fp32 master feeds the embedding gather, bf16 mirror a tied lm_head
    if not c.get("pos_emb_fixed", False):
        vec("wpe", c.get("n_positions", 1024), d, decay=True)
    for l in range(L):
        p = f"h{l}."
        vec(p + "ln1_g", d); vec(p + "ln1_b", d); mat(p + "wqkv", 3 * d, d); vec(p + "bqkv", 3 * d); mat(p + "wo", d, d); vec(p + "bo", d)
        vec(p + "lnc_g", d); vec(p + "lnc_b", d); mat(p + "wq", d, d); vec(p + "bq", d); mat(p + "wkv", 2 * d, d); vec(p + "bkv", 2 * d)
        mat(p + "wco", d, d); vec(p + "bco", d)
        vec(p + "ln2_g", d); vec(p + "ln2_b", d); mat(p + "wfc", 4 * d, d); vec(p + "bfc", 4 * d); mat(p + "wpr", d, 4 * d); vec(p + "bpr", d)
    vec("lnf_g", d); vec("lnf_b", d)
    if not c.get("tie_word_embeddings", False):
        mat("lm_head", V, d)
    for k in range(len(c.get("head_locations") or [])):
        mat(f"head{k}", V, d)
    return S


def _dec_map(c: dict, with_proj: bool, prefix="decoder."):
    L = c["n_layer"]
    m = {}
    one = lambda name, key, fwd=lambda t: t, bwd=lambda t: t: m.__setitem__(name, (lambda sd: fwd(sd[key]), [(key, bwd)]))
    tr_in = lambda t: t.t().contiguous()                   # transformers Conv1D stores (in, out); the store keeps (out, in)
    tr = lambda t: t.t()                                   # export: a transposed VIEW (alias_views hands it to the nn.Parameter; state_dict() clones it contiguous)
    if with_proj:
        one("proj_w", "enc_to_dec_proj.weight"); one("proj_b", "enc_to_dec_proj.bias")
    t = prefix + "transformer."
    if c.get("pos_emb_fixed", False):
        one("wte", t + "wte.emb_layers.0.weight")
    else:
        one("wte", t + "wte.weight"); one("wpe", t + "wpe.weight")
    for l in range(L):
        p, r = f"h{l}.", f"{t}h.{l}."
        one(p + "ln1_g", r + "ln_1.weight"); one(p + "ln1_b", r + "ln_1.bias")
        one(p + "wqkv", r + "attn.c_attn.weight", tr_in, tr); one(p + "bqkv", r + "attn.c_attn.bias")
        one(p + "wo", r + "attn.c_proj.weight", tr_in, tr); one(p + "bo", r + "attn.c_proj.bias")
        one(p + "lnc_g", r + "ln_cross_attn.weight"); one(p + "lnc_b", r + "ln_cross_attn.bias")
        one(p + "wq", r + "crossattention.q_attn.weight", tr_in, tr); one(p + "bq", r + "crossattention.q_attn.bias")
        one(p + "wkv", r + "crossattention.c_attn.weight", tr_in, tr); one(p + "bkv", r + "crossattention.c_attn.bias")
        one(p + "wco", r + "crossattention.c_proj.weight", tr_in, tr); one(p + "bco", r + "crossattention.c_proj.bias")
        one(p + "ln2_g", r + "ln_2.weight"); one(p + "ln2_b", r + "ln_2.bias")
        one(p + "wfc", r + "mlp.c_fc.weight", tr_in, tr); one(p + "bfc", r + "mlp.c_fc.bias")
        one(p + "wpr", r + "mlp.c_proj.weight", tr_in, tr); one(p + "bpr", r + "mlp.c_proj.bias")
    one("lnf_g", t + "ln_f.weight"); one("lnf_b", t + "ln_f.bias")
    if not c.get("tie_word_embeddings", False):
        one("lm_head", prefix + "lm_head.weight")
    for k in range(len(c.get("head_locations") or [])):
        one(f"head{k}", f"{prefix}additional_lm_heads.{k}.weight")
    return m


def _scores(q, k, B, Tq, Tk, H, hd):
    Ts = T.pad8(Tk)
    ac = torch.empty((H, B, Tq, Ts), device=q.device, dtype=F32)
    T.bgemm(q, (hd, Tq * q.stride(0), q.stride(0), 1), k, (hd, Tk * k.stride(0), k.stride(0), 1), ac, (B * Tq * Ts, Tq * Ts, Ts), H, B, Tq, Tk, hd)
    return ac, Ts


def attention_fwd_plain(q, k, v, B, Tq, Tk, H, *, lengths=None, causal=False, drop=None):
    """Un-fused attention forward (used when attention-probability dropout is on): -> (ctx (B*Tq, d) bf16, prob, prob_dropped)."""
    d = q.shape[1]
    hd = d // H
    ac, Ts = _scores(q, k, B, Tq, Tk, H, hd)
    r = T.attn_softmax_fwd(ac, None, lengths, H, B, Tq, Tk, 1.0 / math.sqrt(hd), causal, drop=drop)
    prob, pdrop = r if drop else (r, r)                    # (without dropout the op returns the probabilities alone)
    ctx = torch.empty((B * Tq, d), device=q.device, dtype=BF16)
    T.bgemm(pdrop, (B * Tq * Ts, Tq * Ts, Ts, 1), v, (hd, Tk * v.stride(0), 1, v.stride(0)), ctx, (hd, Tq * d, d), H, B, Tq, hd, Tk)
    return ctx, prob, pdrop


def attention_bwd_plain(q, k, v, dctx, dq, dk, dv, B, Tq, Tk, H, *, lengths=None, causal=False, drop=None, saved=None):
    """Backward of ctx = dropout(softmax(q k^T / sqrt(hd) + mask)) v per (utterance, head); all operands are (rows, >= d) bf16 row
    views with head h at columns [h*hd, (h+1)*hd).  Probabilities are recomputed when the forward was the fused LDS kernel
    (saved = None), or passed in as saved = (prob, prob_dropped) from attention_fwd_plain."""
    d = dctx.shape[1]
    hd = d // H
    scale = 1.0 / math.sqrt(hd)
    sq, sk, sv = q.stride(0), k.stride(0), v.stride(0)
    ac, Ts = _scores(q, k, B, Tq, Tk, H, hd)
    sS = (B * Tq * Ts, Tq * Ts)
    if saved is None:
        prob = pdrop = T.attn_softmax_fwd(ac, None, lengths, H, B, Tq, Tk, scale, causal)
    else:
        prob, pdrop = saved
    dp = ac                                                # reuse the fp32 buffer
    sd_ = dctx.stride(0)
    T.bgemm(dctx, (hd, Tq * sd_, sd_, 1), v, (hd, Tk * sv, sv, 1), dp, (*sS, Ts), H, B, Tq, Tk, hd)
    ds, _ = T.attn_softmax_bwd(prob, dp, H, B, Tq, Tk, scale, drop=drop)
    T.bgemm(pdrop, (*sS, 1, Ts), dctx, (hd, Tq * sd_, 1, sd_), dv, (hd, Tk * dv.stride(0), dv.stride(0)), H, B, Tk, hd, Tq)
    T.bgemm(ds, (*sS, 1, Ts), q, (hd, Tq * sq, 1, sq), dk, (hd, Tk * dk.stride(0), dk.stride(0)), H, B, Tk, hd, Tq)
    T.bgemm(ds, (*sS, Ts, 1), k, (hd, Tk * sk, 1, sk), dq, (hd, Tq * dq.stride(0), dq.stride(0)), H, B, Tq, hd, Tk)


def attention_bwd_fused(q, k, v, ctx, dctx, lse, dq, dk, dv, B, Tq, Tk, H, *, lengths=None, causal=False, drop=None):
    """Backward of `ops_train.attention_x_lse` (head size 64 / 128): ONE walk recomputes the probabilities, leaves P (dropped) and dS in bf16 and accumulates dQ = dS K on the
    way (mi_attention_x_bwd_probs); dV = P^T dctx and dK = dS^T q stay batched GEMMs over them.  Three launches where attention_bwd_plain has seven."""
    d = dctx.shape[1]
    hd = d // H
    prob, ds = T.attn_x_bwd_probs(q, k, v, B, Tq, Tk, H, ctx, dctx, lse, dq, lengths=lengths, causal=causal, drop=drop)
    Ts = prob.shape[-1]
    sS = (B * Tq * Ts, Tq * Ts)
    sd_, sq = dctx.stride(0), q.stride(0)
    T.bgemm(prob, (*sS, 1, Ts), dctx, (hd, Tq * sd_, 1, sd_), dv, (hd, Tk * dv.stride(0), dv.stride(0)), H, B, Tk, hd, Tq, m_valid=lengths)
    T.bgemm(ds, (*sS, 1, Ts), q, (hd, Tq * sq, 1, sq), dk, (hd, Tk * dk.stride(0), dk.stride(0)), H, B, Tk, hd, Tq, m_valid=lengths)


class JointAEDTrainer:
    """forward + backward + AdamW for JointCTCAttentionEncoderDecoder (E-Branchformer encoder + multi-head GPT-2 decoder)."""

    def __init__(self, enc_cfg: dict, dec_cfg: dict, joint_cfg: dict, device="cuda:0", *, lr=2e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, max_grad_norm=1.0, group=None, with_proj=None, dp_sync=True, seed=0):
        c = self.dcfg = dict(dec_cfg)
        self.jcfg = dict(joint_cfg)
        self.device = torch.device(device)
        d, H = c["n_embd"], c["n_head"]
        if d // H not in (64, 128):
            raise NotImplementedError("HIP decoder attention supports head sizes 64 and 128")
        if c.get("activation_function", "gelu_new") != "gelu_new":
            raise NotImplementedError("decoder MLP activation other than gelu_new")
        self.pdrop = {k: float(c.get(k, 0.0) or 0.0) for k in ("resid_pdrop", "embd_pdrop", "attn_pdrop")}
        self.enc = EncoderCTCTrainer(enc_cfg, device, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                                     dp_sync=dp_sync, seed=seed)
        enc_dim = enc_cfg["hidden_size"]
        self.with_proj = (enc_dim != d) if with_proj is None else with_proj
        self.store = ParamStore(decoder_specs(c, enc_dim, self.with_proj), self.device)
        self.map = _dec_map(c, self.with_proj)
        self.sync = GradSync(self.store.flat_g, group, enabled=dp_sync)
        self.hp = self.enc.hp
        if c.get("pos_emb_fixed", False):
            n = c.get("n_positions", 1024)
            inv = 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d))
            s = torch.outer(torch.arange(n).float(), inv)
            self.pos_fixed = torch.cat([s.sin(), s.cos()], -1).to(self.device).contiguous()
            self.emb_scale = float(d) ** 0.5
        else:
            self.pos_fixed, self.emb_scale = None, 1.0

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: dict):
        self.enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
        sdd = {k: v.detach().to(self.device, F32) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and not k.startswith("encoder.")}
        for name in self.store.order:
            self.store.p(name).copy_(self.map[name][0](sdd).reshape(self.store.specs[name].shape))
        self.store.refresh_mirrors(cast=True)

    def set_frozen(self, reference_names):
        """frozen encoder parameters (freeze_encoder): their weight-gradient GEMMs are skipped and, on the native route, AdamW leaves them
        untouched (train.EncoderCTCTrainer.set_frozen); decoder parameters always train"""
        self.enc.set_frozen({k[len("encoder."):] for k in (reference_names or ()) if k.startswith("encoder.")})

    def _export(self, view):
        out = {"encoder." + k: v for k, v in (self.enc.state_dict() if view == "p" else self.enc.grad_dict()).items()}
        for name in self.store.order:
            t = self.store.p(name) if view == "p" else self.store.g(name)
            for key, fn in self.map[name][1]:
                out[key] = fn(t).clone(memory_format=torch.contiguous_format)
        return out

    def state_dict(self): return self._export("p")
    def grad_dict(self): return self._export("g")

    def alias_views(self, which: str = "p") -> dict:
        """train.EncoderCTCTrainer.alias_views for both stores (encoder keys prefixed `encoder.`)"""
        out = self.enc.alias_views(which, prefix="encoder.")
        flat = self.store.flat_p if which == "p" else self.store.flat_g
        base = flat.untyped_storage().data_ptr()
        for name in self.store.order:
            t = self.store._view(flat, name)
            for key, fn in self.map[name][1]:
                v = fn(t)
                out[key] = v if v.untyped_storage().data_ptr() == base else None
        return out

    def import_piece(self, key: str, value):
        if key.startswith("encoder."):
            return self.enc.import_piece(key[len("encoder."):], value)
        for name in self.store.order:
            if len(self.map[name][1]) == 1 and self.map[name][1][0][0] == key:
                self.store.p(name).copy_(self.map[name][0]({key: value.detach().to(self.device, F32)}).reshape(self.store.specs[name].shape))
                return
        raise KeyError(key)

    def export_grad_piece(self, key: str):
        if key.startswith("encoder."):
            return self.enc.export_grad_piece(key[len("encoder."):])
        for name in self.store.order:
            for k, fn in self.map[name][1]:
                if k == key:
                    return fn(self.store.g(name)).clone(memory_format=torch.contiguous_format)
        raise KeyError(key)

    def import_grad_piece(self, key: str, grad):
        if key.startswith("encoder."):
            return self.enc.import_grad_piece(key[len("encoder."):], grad)
        for name in self.store.order:
            if len(self.map[name][1]) == 1 and self.map[name][1][0][0] == key:
                self.store.g(name).copy_(self.map[name][0]({key: grad.detach().to(self.device, F32)}).reshape(self.store.specs[name].shape))
                return
        raise KeyError(key)

    def export_piece(self, key: str):
        if key.startswith("encoder."):
            return self.enc.export_piece(key[len("encoder."):])
        for name in self.store.order:
            for k, fn in self.map[name][1]:
                if k == key:
                    return fn(self.store.p(name)).clone(memory_format=torch.contiguous_format)
        raise KeyError(key)

    def stores(self):
        return [self.enc.store, self.store]

    # ------------------------------------------------------------------ decoder forward + backward (called inside the encoder's backward)
    def _decoder(self, last_hidden, B, T2, key_len, labels, out, gs):
        """last_hidden (B*T2, d_enc) f32.  Fills out[...] and returns d(loss)/d(last_hidden) f32 (gradients scaled by gs)."""
        c, st = self.dcfg, self.store
        P, G, W, WT = st.p, st.g, st.bf, st.bfT
        dev = self.device
        d, H, L, V = c["n_embd"], c["n_head"], c["n_layer"], c["vocab_size"]
        eps = float(c.get("layer_norm_epsilon", 1e-5))
        e32 = lambda *sh: torch.empty(sh, device=dev, dtype=F32)
        e16 = lambda *sh: torch.empty(sh, device=dev, dtype=BF16)
        LN = ops.layernorm_chain
        jc = self.jcfg
        ids = shift_tokens_right(labels, jc["pad_token_id"], jc["decoder_start_token_id"])
        U = ids.shape[1]
        M, Me = B * U, B * T2
        # encoder states at the decoder width (ctc_encoder_plus...:289-293)
        hb = ops.cast_bf16(last_hidden)
        enc_bf = ops.gemm(hb, W("proj_w"), P("proj_b")) if self.with_proj else hb
        pos = self.pos_fixed if self.pos_fixed is not None else P("wpe")
        x = ops.embed_tokens(ids, P("wte"), pos, scale=self.emb_scale)
        pe, pa, pr = self.pdrop["embd_pdrop"], self.pdrop["attn_pdrop"], self.pdrop["resid_pdrop"]
        fused_att = (self.dcfg["n_embd"] // self.dcfg["n_head"]) in (64, 128)          # the LDS-staged attention kernels' head sizes
        seed, sid = self.enc.seed, self.enc._sid
        if pe > 0:
            T.dropout_(x, pe, seed, sid(63, 0))

        def resid_add(res, a16, wname, bname, lay, site):
            """res + dropout(a16 W^T + b)"""
            if pr > 0:
                return T.dropout_add(res, ops.gemm(a16, W(wname), P(bname), out_dtype=F32), 1.0, pr, seed, sid(lay, site))
            return ops.gemm(a16, W(wname), P(bname), out_dtype=F32, resid=res, alpha=1.0)
        locs = list(c.get("head_locations") or [])
        weights = list(c.get("head_weights") or [1.0])
        lsm = float(c.get("lsm_factor", 0.0))
        saved, taps = [], {}
        if 0 in locs:
            taps[0] = x
        for l in range(L):
            p = f"h{l}."
            S = {"x": x, "p1": None, "p2": None}
            a1 = e16(M, d)
            LN(x, lna=(P(p + "ln1_g"), P(p + "ln1_b")), eps2=eps, outa=a1)
            qkv = ops.gemm(a1, W(p + "wqkv"), P(p + "bqkv"))
            if fused_att:                                       # head size 64 / 128: fused forward with the row log-sum-exp (and the dropout mask) for the fused backward
                ctx1, lse1 = T.attention_x_lse(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, U, U, H, causal=True, drop=(pa, seed, sid(32 + l, 0)) if pa > 0 else None)
                S["p1"] = ("lse", lse1)
            elif pa > 0:
                ctx1, *S["p1"] = attention_fwd_plain(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, U, U, H, causal=True, drop=(pa, seed, sid(32 + l, 0)))
            else:
                ctx1 = ops.attention_general(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, U, U, H, causal=True)
            x1 = resid_add(x, ctx1, p + "wo", p + "bo", 32 + l, 1)
            a2 = e16(M, d)
            LN(x1, lna=(P(p + "lnc_g"), P(p + "lnc_b")), eps2=eps, outa=a2)
            qq = ops.gemm(a2, W(p + "wq"), P(p + "bq"))
            kv = ops.gemm(enc_bf, W(p + "wkv"), P(p + "bkv"))
            if fused_att:
                ctx2, lse2 = T.attention_x_lse(qq, kv[:, :d], kv[:, d:], B, U, T2, H, lengths=key_len, drop=(pa, seed, sid(32 + l, 2)) if pa > 0 else None)
                S["p2"] = ("lse", lse2)
            elif pa > 0:
                ctx2, *S["p2"] = attention_fwd_plain(qq, kv[:, :d], kv[:, d:], B, U, T2, H, lengths=key_len, drop=(pa, seed, sid(32 + l, 2)))
            else:
                ctx2 = ops.attention_general(qq, kv[:, :d], kv[:, d:], B, U, T2, H, lengths=key_len)
            x2 = resid_add(x1, ctx2, p + "wco", p + "bco", 32 + l, 3)
            a3 = e16(M, d)
            LN(x2, lna=(P(p + "ln2_g"), P(p + "ln2_b")), eps2=eps, outa=a3)
            mp = ops.gemm(a3, W(p + "wfc"), P(p + "bfc"))
            mm = T.act_fwd(mp, "gelu_new")
            x3 = resid_add(x2, mm, p + "wpr", p + "bpr", 32 + l, 4)
            S.update(a1=a1, qkv=qkv, ctx1=ctx1, x1=x1, a2=a2, qq=qq, kv=kv, ctx2=ctx2, x2=x2, a3=a3, mp=mp, mm=mm)
            saved.append(S)
            x = x3
            if (l + 1) in locs and l + 1 < L:
                taps[l + 1] = x
        hid = e16(M, d)
        LN(x, lna=(P("lnf_g"), P("lnf_b")), eps2=eps, outa=hid)
        Vp = T.pad64(V)
        lm_name = "wte" if c.get("tie_word_embeddings", False) else "lm_head"

        def head(hb16, wname, weight):
            """logits + CE of one head; returns (logits view, loss, dlogits bf16 (M, Vp))"""
            buf = e32(B, U, Vp)
            ops.gemm(hb16, W(wname), None, out=buf.view(M, Vp))
            lg = buf[..., :V]
            acc = torch.zeros((2,), device=dev, dtype=F32)
            rows = torch.empty((B * (U - 1),), device=dev, dtype=F32)
            ops._lib.check(ops._lib.lib().mi_ce_label_smoothing(lg.data_ptr(), lg.stride(1), labels.data_ptr(), B, U, 1, V, lsm, acc.data_ptr(), rows.data_ptr(),
                                                                 torch.cuda.current_stream().cuda_stream), "mi_ce_label_smoothing")
            dl = T.ce_label_smoothing_bwd(lg, labels, acc, shift=1, eps=lsm, weight=weight * gs, ldo=Vp)
            return lg, acc[0] / acc[1], dl

        wdec = 1.0 - jc["ctc_weight"]
        logits, ce, dl = head(hid, lm_name, wdec * weights[-1])
        dec_loss = weights[-1] * ce
        # ---- backward of the last head
        dhid = ops.gemm(dl, WT(lm_name))
        tnb = T.TnBatch()            # every weight gradient of the decoder's backward: one grouped launch at the end (46 problems, ~150 output tiles at 6 x 256)
        # first backward after zero_grad: the launch writes its targets instead of adding into the zeros (train.EncoderCTCTrainer._forward_backward); the embedding
        # gradient — the one other contribution to a matrix of this store (wte, tied to the lm head) — is therefore added AFTER the flush below
        tnb.overwrite = bool(self.enc.dw_overwrite and getattr(st, "fresh", False))
        st.fresh = False
        T.gemm_tn_(G(lm_name), dl, hid, n_store=V, defer=tnb)
        tap_grads, final_dys = {}, [dhid]
        for k, loc in enumerate(locs):
            src16 = ops.cast_bf16(taps[loc]) if loc in taps else hid
            _, ce_k, dl_k = head(src16, f"head{k}", wdec * weights[k])
            dec_loss = dec_loss + weights[k] * ce_k
            T.gemm_tn_(G(f"head{k}"), dl_k, src16, n_store=V, defer=tnb)
            if loc in taps:
                tap_grads[loc] = ops.gemm(dl_k, WT(f"head{k}"), out_dtype=F32)
            else:                                            # a head on the last hidden state reads ln_f's output
                final_dys.append(ops.gemm(dl_k, WT(f"head{k}")))
        dx = e32(M, d)
        for i, dy in enumerate(final_dys):
            T.layernorm_bwd(x, P("lnf_g"), dy, dx, accumulate=i > 0, dgamma=G("lnf_g"), dbeta=G("lnf_b"), eps=eps)
        denc = torch.zeros((Me, d), device=dev, dtype=F32)
        for l in range(L - 1, -1, -1):
            p = f"h{l}."
            S = saved[l]
            if (l + 1) in tap_grads:
                T.axpy_(dx, tap_grads[l + 1])
            dres = (lambda lay, site: T.dropout_(dx, pr, seed, sid(lay, site), out=e16(M, d))) if pr > 0 else (lambda lay, site: T.add_cast(dx))
            # MLP
            dyb = dres(32 + l, 4)
            dm = T.linear_bwd(dyb, S["mm"], WT(p + "wpr"), dw=G(p + "wpr"), db=G(p + "bpr"), defer=tnb)
            dmp = T.act_bwd(dm, S["mp"], "gelu_new")
            da3 = T.linear_bwd(dmp, S["a3"], WT(p + "wfc"), dw=G(p + "wfc"), db=G(p + "bfc"), defer=tnb)
            T.layernorm_bwd(S["x2"], P(p + "ln2_g"), da3, dx, accumulate=True, dgamma=G(p + "ln2_g"), dbeta=G(p + "ln2_b"), eps=eps)
            # cross-attention
            dyb = dres(32 + l, 3)
            dctx2 = T.linear_bwd(dyb, S["ctx2"], WT(p + "wco"), dw=G(p + "wco"), db=G(p + "bco"), defer=tnb)
            dqq, dkv = e16(M, d), e16(Me, 2 * d)
            kv = S["kv"]
            if isinstance(S["p2"], tuple) and S["p2"][0] == "lse":
                attention_bwd_fused(S["qq"], kv[:, :d], kv[:, d:], S["ctx2"], dctx2, S["p2"][1], dqq, dkv[:, :d], dkv[:, d:], B, U, T2, H, lengths=key_len,
                                    drop=(pa, seed, sid(32 + l, 2)) if pa > 0 else None)
            else:
                attention_bwd_plain(S["qq"], kv[:, :d], kv[:, d:], dctx2, dqq, dkv[:, :d], dkv[:, d:], B, U, T2, H, lengths=key_len,
                                    drop=(pa, seed, sid(32 + l, 2)) if pa > 0 else None, saved=S["p2"])
            da2 = T.linear_bwd(dqq, S["a2"], WT(p + "wq"), dw=G(p + "wq"), db=G(p + "bq"), defer=tnb)
            T.linear_bwd(dkv, enc_bf, WT(p + "wkv"), dw=G(p + "wkv"), db=G(p + "bkv"), need_dx=False, defer=tnb)
            ops.gemm(dkv, WT(p + "wkv")[:, :2 * d], out=denc, resid=denc, alpha=1.0)
            T.layernorm_bwd(S["x1"], P(p + "lnc_g"), da2, dx, accumulate=True, dgamma=G(p + "lnc_g"), dbeta=G(p + "lnc_b"), eps=eps)
            # causal self-attention
            dyb = dres(32 + l, 1)
            dctx1 = T.linear_bwd(dyb, S["ctx1"], WT(p + "wo"), dw=G(p + "wo"), db=G(p + "bo"), defer=tnb)
            qkv = S["qkv"]
            dqkv = e16(M, 3 * d)
            if isinstance(S["p1"], tuple) and S["p1"][0] == "lse":
                attention_bwd_fused(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], S["ctx1"], dctx1, S["p1"][1], dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, U, U, H,
                                    causal=True, drop=(pa, seed, sid(32 + l, 0)) if pa > 0 else None)
            else:
                attention_bwd_plain(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], dctx1, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, U, U, H, causal=True,
                                    drop=(pa, seed, sid(32 + l, 0)) if pa > 0 else None, saved=S["p1"])
            da1 = T.linear_bwd(dqkv, S["a1"], WT(p + "wqkv"), dw=G(p + "wqkv"), db=G(p + "bqkv"), defer=tnb)
            T.layernorm_bwd(S["x"], P(p + "ln1_g"), da1, dx, accumulate=True, dgamma=G(p + "ln1_g"), dbeta=G(p + "ln1_b"), eps=eps)
        if 0 in tap_grads:
            T.axpy_(dx, tap_grads[0])
        if pe > 0:
            T.dropout_(dx, pe, seed, sid(63, 0))
        if self.with_proj:
            dh = T.linear_bwd(T.add_cast(denc), hb, WT("proj_w"), dw=G("proj_w"), db=G("proj_b"), dx_dtype=F32, defer=tnb)
        else:
            dh = denc
        tnb.flush()
        T.embed_tokens_bwd(ids, dx, G("wte"), None if self.pos_fixed is not None else G("wpe"), scale=self.emb_scale, heavy_id=self.jcfg.get("pad_token_id"))
        self.sync.launch(0, st.n)
        out.update(dec_loss=dec_loss, logits=logits, encoder_hidden=enc_bf)
        return dh

    # ------------------------------------------------------------------ step
    def forward_backward(self, feats, feat_lengths, labels):
        with ops.pinned_stream():
            return self._forward_backward(feats, feat_lengths, labels)

    def _forward_backward(self, feats, feat_lengths, labels):
        jc = self.jcfg
        labels = labels.contiguous()
        B = feats.shape[0]
        out = {}
        w = jc["ctc_weight"]

        def hook(last_hidden, outer_len):
            T2 = last_hidden.shape[0] // B
            key_len = torch.clamp(outer_len, max=T2) if feat_lengths is not None else None     # cross mask from the OUTER lengths (quirk 8')
            return self._decoder(last_hidden, B, T2, key_len, labels, out, 1.0 / self.sync.world)

        eo = self.enc.forward_backward(feats, feat_lengths, labels, loss_scale=w, extra_hidden_grad=hook)
        out.update(enc_loss=eo["loss"], encoder_logits=eo["logits"], loss=w * eo["loss"] + (1 - w) * out["dec_loss"])
        return out

    def optimizer_step(self, lr=None):
        with ops.pinned_stream():
            return self._optimizer_step(lr)

    def _optimizer_step(self, lr=None):
        hp = self.hp
        self.enc.sync.wait(); self.sync.wait()
        self.enc.store.zero_frozen_grads()          # frozen encoder parameters (set_frozen): no update, not in the clip norm
        sc = self.enc._scal
        sc.zero_()
        T.sumsq_(sc[0:1], self.enc.store.flat_g)
        T.sumsq_(sc[0:1], self.store.flat_g)
        T.clip_coef(sc[0:1], hp["max_grad_norm"] if hp["max_grad_norm"] else 0.0, sc[1:4], hp.get("grad_norm_skip", 0.0))
        for st in (self.enc.store, self.store):
            st.step_count += 1
            T.adamw_step_(st.flat_p, st.flat_g, st.flat_m, st.flat_v, st.decay, lr=hp["lr"] if lr is None else lr, betas=hp["betas"], eps=hp["eps"],
                          weight_decay=hp["weight_decay"], step=st.step_count, norm_coef=sc[1:4], mirror=st.flat_bf)
            st.refresh_mirrors(cast=False)
        return sc[1]

    def train_step(self, feats, feat_lengths, labels, lr=None):
        self.enc.store.zero_grad(); self.store.zero_grad()
        out = self.forward_backward(feats, feat_lengths, labels)
        out["grad_norm"] = self.optimizer_step(lr)
        return out
