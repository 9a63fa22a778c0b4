"""GPT-2 cross-attention decoder + joint CTC/attention forward on the HIP kernels.

Host-side orchestration (one C-ABI call per op) of
  * reference `src/models/decoders/multi_head_gpt2.py:80-170` (GPT2LMMultiHeadModel: auxiliary lm heads on intermediate
    hidden states, shifted label-smoothed CE),
  * the transformers GPT-2 block it inherits (ln_1 -> causal self-attn -> ln_cross_attn -> cross-attn over the encoder
    frames -> ln_2 -> gelu_new MLP; Conv1D weights are stored (in, out) and are transposed once at pack time),
  * reference `src/models/embeddings.py` (fixed sinusoidal positions + sqrt(d)-scaled embedding when `pos_emb_fixed`),
  * reference `src/models/ctc_encoder_plus_autoregressive_decoder.py:237-358` (JointAED: encoder CTC loss, enc_to_dec_proj,
    cross mask from the OUTER length formula, shift_tokens_right, loss = w*CTC + (1-w)*CE).
Eval-mode semantics; the fp32 residual stream / bf16 GEMM operands precision model is the encoder's.
"""
from __future__ import annotations

import ctypes as C
import time

import torch

from . import _lib, ops
from .engine import EBranchformerEngine

BF16 = torch.bfloat16


class GPT2DecoderEngine:
    def __init__(self, cfg: dict, device="cuda:0"):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        d, H = cfg["n_embd"], cfg["n_head"]
        if d // H not in (64, 128):
            raise NotImplementedError("HIP decoder attention supports head sizes 64 and 128 (the reference's GPT-2 configs use 64)")
        if cfg.get("activation_function", "gelu_new") != "gelu_new":
            raise NotImplementedError("decoder MLP activation other than gelu_new")
        self.w = None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: dict, prefix: str = "decoder."):
        c, dev = self.cfg, self.device
        d, L = c["n_embd"], c["n_layer"]
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        lin = lambda t: t.detach().to(dev, torch.float32).t().to(BF16).contiguous()      # Conv1D (in,out) -> (out,in) bf16
        w = {"layers": []}
        if c.get("pos_emb_fixed", False):
            w["wte"] = f32(sd[prefix + "transformer.wte.emb_layers.0.weight"])
            w["scale"] = float(d) ** 0.5
            n = c.get("n_positions", 1024)
            inv = 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d))
            s = torch.outer(torch.arange(n).float(), inv)
            w["pos"] = torch.cat([s.sin(), s.cos()], -1).to(dev).contiguous()
        else:
            w["wte"] = f32(sd[prefix + "transformer.wte.weight"])
            w["scale"] = 1.0
            w["pos"] = f32(sd[prefix + "transformer.wpe.weight"])
        for l in range(L):
            p = f"{prefix}transformer.h.{l}."
            g = lambda n: f32(sd[p + n])
            w["layers"].append(dict(
                ln1=(g("ln_1.weight"), g("ln_1.bias")), wqkv=lin(sd[p + "attn.c_attn.weight"]), bqkv=g("attn.c_attn.bias"),
                wo=lin(sd[p + "attn.c_proj.weight"]), bo=g("attn.c_proj.bias"),
                lnc=(g("ln_cross_attn.weight"), g("ln_cross_attn.bias")),
                wq=lin(sd[p + "crossattention.q_attn.weight"]), bq=g("crossattention.q_attn.bias"),
                wkv=lin(sd[p + "crossattention.c_attn.weight"]), bkv=g("crossattention.c_attn.bias"),
                wco=lin(sd[p + "crossattention.c_proj.weight"]), bco=g("crossattention.c_proj.bias"),
                ln2=(g("ln_2.weight"), g("ln_2.bias")), wfc=lin(sd[p + "mlp.c_fc.weight"]), bfc=g("mlp.c_fc.bias"),
                wpr=lin(sd[p + "mlp.c_proj.weight"]), bpr=g("mlp.c_proj.bias")))
        w["lnf"] = (f32(sd[prefix + "transformer.ln_f.weight"]), f32(sd[prefix + "transformer.ln_f.bias"]))
        w["heads"] = [sd[f"{prefix}additional_lm_heads.{k}.weight"].detach().to(dev, torch.float32).to(BF16).contiguous()
                      for k in range(len(c.get("head_locations") or []))]
        w["lm_head"] = sd[prefix + "lm_head.weight"].detach().to(dev, torch.float32).to(BF16).contiguous()
        self.w = w
        # pointer table of mi_gpt2_step (csrc/decoder_step.hip): 5 globals, then 18 per layer
        ptrs = [w["wte"], w["pos"], w["lnf"][0], w["lnf"][1], w["lm_head"]]
        for lw in w["layers"]:
            ptrs += [lw["ln1"][0], lw["ln1"][1], lw["wqkv"], lw["bqkv"], lw["wo"], lw["bo"], lw["lnc"][0], lw["lnc"][1], lw["wq"], lw["bq"],
                     lw["wco"], lw["bco"], lw["ln2"][0], lw["ln2"][1], lw["wfc"], lw["bfc"], lw["wpr"], lw["bpr"]]
        self._wtable = (C.c_void_p * len(ptrs))(*[t.data_ptr() for t in ptrs])
        self._gcfg = _lib.Gpt2Config(d=d, H=c["n_head"], L=L, V=w["lm_head"].shape[0], eps=float(c.get("layer_norm_epsilon", 1e-5)))
        self._step_ws = None

    # ------------------------------------------------------------------ building blocks
    def cross_kv(self, enc_bf16: torch.Tensor):
        """Per-layer cross-attention keys/values of the encoder frames: list of (B*T', 2d) bf16 (computed once per utterance)."""
        return [ops.gemm(enc_bf16, lw["wkv"], lw["bkv"]) for lw in self.w["layers"]]

    def _logits(self, hid_bf16, head_w):
        V = head_w.shape[0]
        Vp = (V + 7) // 8 * 8
        buf = torch.empty((hid_bf16.shape[0], Vp), device=self.device, dtype=torch.float32)
        ops.gemm(hid_bf16, head_w, None, out=buf[:, :V])
        return buf[:, :V]

    def _block(self, l, x, B, U, kv, T_enc, enc_len, self_k=None, self_v=None, past=0, Lmax=0):
        """One GPT-2 block on the fp32 stream x (B*U, d).  With a KV cache (self_k/self_v (B, Lmax, d)) U new tokens are
        appended at position `past` and attend to past+U keys."""
        c, lw = self.cfg, self.w["layers"][l]
        d, H, eps = c["n_embd"], c["n_head"], c.get("layer_norm_epsilon", 1e-5)
        M = x.shape[0]
        a = torch.empty((M, d), device=self.device, dtype=BF16)
        ops.layernorm_chain(x, lna=lw["ln1"], eps2=eps, outa=a)
        qkv = ops.gemm(a, lw["wqkv"], lw["bqkv"])
        if self_k is None:
            ctx = ops.attention_general(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, U, U, H, causal=True)
        else:
            self_k[:, past:past + U] = qkv[:, d:2 * d].reshape(B, U, d)
            self_v[:, past:past + U] = qkv[:, 2 * d:].reshape(B, U, d)
            ctx = ops.attention_general(qkv[:, :d], self_k.view(B * Lmax, d), self_v.view(B * Lmax, d), B, U, past + U, H,
                                        causal=True, kv_bstride=Lmax * d)
        ops.gemm(ctx, lw["wo"], lw["bo"], out=x, resid=x, alpha=1.0)
        ops.layernorm_chain(x, lna=lw["lnc"], eps2=eps, outa=a)
        qq = ops.gemm(a, lw["wq"], lw["bq"])
        ctx = ops.attention_general(qq, kv[:, :d], kv[:, d:], B, U, T_enc, H, lengths=enc_len)
        ops.gemm(ctx, lw["wco"], lw["bco"], out=x, resid=x, alpha=1.0)
        ops.layernorm_chain(x, lna=lw["ln2"], eps2=eps, outa=a)
        m = ops.gemm(a, lw["wfc"], lw["bfc"], act="gelu_new")
        ops.gemm(m, lw["wpr"], lw["bpr"], out=x, resid=x, alpha=1.0)
        return x

    # ------------------------------------------------------------------ teacher-forced forward
    def forward(self, ids: torch.Tensor, enc_bf16: torch.Tensor, T_enc: int, enc_len, labels=None):
        """ids (B,U) int64; enc_bf16 (B*T', d) bf16 (already projected to the decoder width); enc_len (B) int32 valid frames
        or None.  Returns dict(logits (B,U,V) fp32, loss | None) = GPT2LMMultiHeadModel.forward."""
        c, w = self.cfg, self.w
        B, U = ids.shape
        d, L, eps = c["n_embd"], c["n_layer"], c.get("layer_norm_epsilon", 1e-5)
        x = ops.embed_tokens(ids, w["wte"], w["pos"], scale=w["scale"])
        kvs = self.cross_kv(enc_bf16)
        locs = list(c.get("head_locations") or [])
        taps = {}
        if 0 in locs:
            taps[0] = x.clone()
        for l in range(L):
            x = self._block(l, x, B, U, kvs[l], T_enc, enc_len)
            if (l + 1) in locs and l + 1 < L:
                taps[l + 1] = x.clone()
        hid = torch.empty((B * U, d), device=self.device, dtype=BF16)
        ops.layernorm_chain(x, lna=w["lnf"], eps2=eps, outa=hid)
        logits = self._logits(hid, w["lm_head"]).view(B, U, -1)
        loss = None
        if labels is not None:
            weights = list(c.get("head_weights") or [1.0])
            loss = weights[-1] * ops.ce_label_smoothing(logits, labels, shift=1, eps=c.get("lsm_factor", 0.0))
            for k, loc in enumerate(locs):
                hb = ops.cast_bf16(taps[loc]) if loc in taps else hid
                lg = self._logits(hb, w["heads"][k]).view(B, U, -1)
                loss = loss + weights[k] * ops.ce_label_smoothing(lg, labels, shift=1, eps=c.get("lsm_factor", 0.0))
        return dict(logits=logits, loss=loss)

    # ------------------------------------------------------------------ incremental decoding
    def init_cache(self, B: int, Lmax: int):
        """KV cache: two sets of (B, Lmax, d) tensors per layer (beam re-ordering copies set A -> set B in one kernel and swaps)."""
        d, L = self.cfg["n_embd"], self.cfg["n_layer"]
        # one allocation and one fill for all 4 L tensors.  Zero, not empty: the MFMA attention kernel stages whole 32-key tiles, and a V row past the last key meets a
        # probability of exactly 0 — which only gives 0 if the row holds finite numbers
        buf = torch.zeros((4, L, B, Lmax, d), device=self.device, dtype=BF16)
        cache = dict(k=list(buf[0].unbind(0)), v=list(buf[1].unbind(0)), k2=list(buf[2].unbind(0)), v2=list(buf[3].unbind(0)), past=0, Lmax=Lmax)
        self._cache_tables(cache)
        return cache

    @staticmethod
    def _cache_tables(cache):
        tab = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        cache["tk"], cache["tv"], cache["tk2"], cache["tv2"] = tab(cache["k"]), tab(cache["v"]), tab(cache["k2"]), tab(cache["v2"])

    def reorder_cache(self, cache, beam_idx: torch.Tensor):
        """transformers' `_reorder_cache`: every layer's K and V rows follow their beam (one kernel for all 2L tensors)."""
        L = len(cache["k"])
        B, Lmax, d = cache["k"][0].shape
        beam_idx = beam_idx.to(device=self.device, dtype=torch.long).contiguous()
        _lib.check(_lib.lib().mi_kv_cache_reorder(cache["tk"], cache["tv"], cache["tk2"], cache["tv2"], beam_idx.data_ptr(), L, B, cache["past"], Lmax, d,
                                                  torch.cuda.current_stream().cuda_stream), "mi_kv_cache_reorder")
        cache["k"], cache["k2"] = cache["k2"], cache["k"]
        cache["v"], cache["v2"] = cache["v2"], cache["v"]
        cache["tk"], cache["tk2"] = cache["tk2"], cache["tk"]
        cache["tv"], cache["tv2"] = cache["tv2"], cache["tv"]

    def step(self, ids_new: torch.Tensor, cache, kvs, T_enc: int, enc_len):
        """ids_new (B, U_new) -> logits (B, V) of the LAST new position; appends to the KV cache.  One C call (mi_gpt2_step)."""
        c, w = self.cfg, self.w
        ids_new = ids_new.contiguous()
        B, U = ids_new.shape
        past, Lmax = cache["past"], cache["Lmax"]
        L_ = _lib.lib()
        nbytes = L_.mi_gpt2_step_workspace_bytes(C.byref(self._gcfg), B, U)
        if self._step_ws is None or self._step_ws.numel() < nbytes:
            self._step_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        if cache.get("kv_id") != id(kvs):                              # pointer table of the per-layer encoder K/V
            cache["tkv"] = (C.c_void_p * len(kvs))(*[t.data_ptr() for t in kvs])
            cache["kv_id"] = id(kvs)
        V = self._gcfg.V
        Vp = (V + 7) // 8 * 8
        buf = torch.empty((B, Vp), device=self.device, dtype=torch.float32)
        _lib.check(L_.mi_gpt2_step(C.byref(self._gcfg), self._wtable, ids_new.data_ptr(), B, U, past, Lmax, cache["tk"], cache["tv"], cache["tkv"], T_enc,
                                   enc_len.data_ptr() if enc_len is not None else None, float(w["scale"]), self._step_ws.data_ptr(), self._step_ws.numel(),
                                   buf.data_ptr(), Vp, torch.cuda.current_stream().cuda_stream), "mi_gpt2_step")
        cache["past"] = past + U
        return buf[:, :V]

    def step_py(self, ids_new: torch.Tensor, cache, kvs, T_enc: int, enc_len):
        """same step driven op by op from Python (kept as the cross-check of the C driver)"""
        c, w = self.cfg, self.w
        B, U = ids_new.shape
        d, L, eps = c["n_embd"], c["n_layer"], c.get("layer_norm_epsilon", 1e-5)
        past, Lmax = cache["past"], cache["Lmax"]
        x = ops.embed_tokens(ids_new, w["wte"], w["pos"], scale=w["scale"], pos_offset=past)
        for l in range(L):
            x = self._block(l, x, B, U, kvs[l], T_enc, enc_len, cache["k"][l], cache["v"][l], past, Lmax)
        cache["past"] = past + U
        last = x.view(B, U, d)[:, -1].contiguous()
        hid = torch.empty((B, d), device=self.device, dtype=BF16)
        ops.layernorm_chain(last, lna=w["lnf"], eps2=eps, outa=hid)
        return self._logits(hid, w["lm_head"])


def shift_tokens_right(labels: torch.Tensor, pad_id: int, start_id: int) -> torch.Tensor:
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


class JointAEDEngine:
    """JointCTCAttentionEncoderDecoder.forward (eval) on the HIP path."""

    def __init__(self, enc_cfg: dict, dec_cfg: dict, joint_cfg: dict, device="cuda:0"):
        self.enc = EBranchformerEngine(enc_cfg, device)
        self.dec = GPT2DecoderEngine(dec_cfg, device)
        self.jcfg = dict(joint_cfg)
        self.device = torch.device(device)
        self.proj = None

    def load_state_dict(self, sd: dict):
        self.enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
        self.dec.load_state_dict(sd, "decoder.")
        if "enc_to_dec_proj.weight" in sd:
            self.proj = (sd["enc_to_dec_proj.weight"].detach().to(self.device, torch.float32).to(BF16).contiguous(),
                         sd["enc_to_dec_proj.bias"].detach().to(self.device, torch.float32).contiguous())

    def encode(self, feats, feat_len):
        """-> (encoder out dict, encoder states for the decoder (B*T', d_dec) bf16, T', cross-attention key lengths)."""
        out = self.enc.forward(feats, feat_len, want_hidden=True)
        B, T2, d = out["last_hidden"].shape
        hid = out["last_hidden"].reshape(B * T2, d)
        if self.proj is not None:
            enc_bf = ops.gemm(ops.cast_bf16(hid), self.proj[0], self.proj[1])       # ctc_encoder_plus...:289-293
        else:
            enc_bf = ops.cast_bf16(hid)
        # cross mask uses the OUTER (un-padded) length formula, quirk 8' (:296-301); no attention_mask -> all frames
        key_len = torch.clamp(out["outer_len"], max=T2) if feat_len is not None else None
        return out, enc_bf, T2, key_len

    def forward(self, feats, feat_len, labels):
        c = self.jcfg
        enc_out, enc_bf, T2, key_len = self.encode(feats, feat_len)
        enc_loss, _, _ = ops.ctc_loss(enc_out["logits"], labels, enc_out["outer_len"],
                                      reduction=self.enc.cfg.get("ctc_loss_reduction", "mean"),
                                      zero_infinity=self.enc.cfg.get("ctc_zero_infinity", False), lse=enc_out.get("lse"))
        dec_ids = shift_tokens_right(labels, c["pad_token_id"], c["decoder_start_token_id"])
        d = self.dec.forward(dec_ids, enc_bf, T2, key_len, labels)
        w = c["ctc_weight"]
        return dict(loss=w * enc_loss + (1 - w) * d["loss"], enc_loss=enc_loss, dec_loss=d["loss"], logits=d["logits"],
                    encoder_logits=enc_out["logits"], encoder_hidden=enc_bf)


# ---------------------------------------------------------------------------------------------------------------------
# Joint CTC/attention decoding (config 5 of BASELINE.json): greedy (W = 1) and beam search with the device-side CTC
# prefix scorer.  Mirrors what the reference obtains from GenerationMixin.generate + its logits processors
# (ctc_encoder_plus_autoregressive_decoder.py:360-482, hf_shared_models/ED_small.py:20-22: ctc_weight 0.3, num_beams 5):
# scores = log_softmax(decoder logits) -> [CTC processor: pad masked, (1-w)*att + w*ctc] -> + running beam score -> top 2W over W*V ->
# candidates that stop (EOS, or max_length reached) among the first W ranks join the kept hypotheses with sum_logprob / generated_tokens**length_penalty,
# the best W are kept; the first W that did not stop run on; the early-stop rule compares the best running beam with the worst kept hypothesis
# (transformers/generation/utils.py `_beam_search` of the installed 5.x — the loop the reference's generate() runs here; pinned by tests/golden/gen_*.npz,
# restated in oracle/generate_ref.py).
_ES_MODE = {False: 0, True: 1, "never": 2}


def _step_denoms(cur_len, max_length, length_penalty, early_stopping):
    """(closing denominator, early-stop denominator) of the step that extends prefixes of `cur_len` tokens (one of them the start token), as fp32 values:
    generated tokens of a hypothesis closed now = cur_len; hypothetical length of the early-stop rule = cur_len, or max_length - 1 for "never" with a positive penalty."""
    import numpy as np
    hyp = (max_length - 1) if (early_stopping == "never" and length_penalty > 0.0) else cur_len
    return float(np.float32(cur_len ** length_penalty)), float(np.float32(hyp ** length_penalty))


def _check_generate_args(num_beams, max_length, early_stopping):
    if early_stopping not in _ES_MODE:
        raise ValueError(f"early_stopping must be False, True or 'never', got {early_stopping!r}")
    if num_beams < 1 or max_length < 2:
        raise ValueError(f"num_beams >= 1 and max_length >= 2 required, got {num_beams}, {max_length}")


def generate(joint: "JointAEDEngine", feats, feat_len, *, num_beams=1, max_length=64, ctc_weight=0.3, length_penalty=1.0, early_stopping=False,
             eos_token_id=1, pad_token_id=None, start_token_id=None, space_token_id=-1, apply_eos_space_trick=False, eos_space_trick_weight=1.0,
             run_ahead=2, stats=None, trace=None):
    """Device-resident decoding loop: per token the decoder step (one C call), the row log-sum-exp and ONE launch that mixes the CTC prefix scores in, takes the top 2W
    candidates, applies the beam loop's rules and moves ids / beam scores / kept hypotheses on the device (csrc/beam_step.hip).  The CTC prefix scorer of step t
    depends on the prefixes only, not on the decoder's logits: it runs on a second stream beside the decoder step.  Nothing is copied to the host until decoding ends, except
    the per-utterance `done` flags (the kernel writes them into pinned, device-mapped memory): the host stays at most `run_ahead` steps in front of the GPU and stops enqueuing once every utterance is done.
    Returns per utterance dict(tokens, score, hypotheses = the kept (score, tokens), best first, at most W).  Same hypotheses, scores and order as `generate_stepwise`
    (same arithmetic, operation for operation).  `stats` (a dict) receives the host time spent enqueuing the token loop and the number of steps enqueued; `trace` (a list)
    receives per step the (B, 2W) candidate values and indices the kernel walked and the (B) done flags before the step (device tensors)."""
    from .decoding import CTCRescorerLogitsProcessor
    _check_generate_args(num_beams, max_length, early_stopping)
    dev = joint.device
    c = joint.jcfg
    pad = c["pad_token_id"] if pad_token_id is None else pad_token_id
    start = c["decoder_start_token_id"] if start_token_id is None else start_token_id
    W = num_beams
    L_ = _lib.lib()
    main = torch.cuda.current_stream()
    V = joint.dec.w["lm_head"].shape[0]
    Lmax = max_length + 1
    if apply_eos_space_trick or W > 16 or W * V >= (1 << 24) or W * (max_length + Lmax) * 8 > 96 * 1024:
        # the eos / space trick (ctc_scorer.py:333-349) lives in the processor the host loop calls; beyond mi_beam_step's limits (beams, candidates, the two id buffers it
        # stages in 96 KiB of LDS) the bookkeeping runs on the host as well (still the HIP kernels for everything else) — decided here, not by an error in the middle of a decode
        return generate_stepwise(joint, feats, feat_len, num_beams=num_beams, max_length=max_length, ctc_weight=ctc_weight, length_penalty=length_penalty,
                                 early_stopping=early_stopping, eos_token_id=eos_token_id, pad_token_id=pad_token_id, start_token_id=start_token_id,
                                 space_token_id=space_token_id, apply_eos_space_trick=apply_eos_space_trick, eos_space_trick_weight=eos_space_trick_weight)
    enc_out, enc_bf, T2, key_len = joint.encode(feats, feat_len)
    B = feats.shape[0]
    d = enc_bf.shape[1]
    enc_rep = enc_bf.view(B, T2, d).repeat_interleave(W, 0).reshape(B * W * T2, d)
    key_rep = key_len.repeat_interleave(W) if key_len is not None else None
    kvs = joint.dec.cross_kv(enc_rep)
    cache = joint.dec.init_cache(B * W, Lmax)
    proc, side = None, None
    if ctc_weight > 0:
        lens = enc_out["outer_len"].clamp(max=T2)
        proc = CTCRescorerLogitsProcessor(enc_out["logits"], lens, pad, eos_token_id, 0, ctc_weight, W, space_token_id, False, 1.0)
        if proc.O != V:
            raise ValueError(f"CTC head has {proc.O} classes, the decoder {V}: joint decoding needs one vocabulary")
        # beside the decoder step the scorer competes with it for the memory system (its step is a chain of latency-bound launches: 358 -> 409 us per token at W = 1 with the
        # scorer running): the form that re-runs the selected chains moves a tenth of the bytes of the form that keeps every chain (425 us), so it is the one used here —
        # for a processor called in sequence (HF generate, generate_stepwise) the single-scan form is the faster one (tools/decode_step_timing.py)
        proc.FULL_STATE_BYTES = 0
        side = torch.cuda.Stream(device=dev)                 # a higher stream priority for the scorer was measured: no better (21.4 -> 25 ms and erratic)
    n_bh = B * W
    ids = torch.full((n_bh, Lmax), pad, dtype=torch.long, device=dev)
    ids[:, 0] = start
    beam_scores = torch.zeros((B, W), device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1).contiguous()
    done = torch.zeros((B,), dtype=torch.int32, device=dev)
    nfin = torch.zeros((B,), dtype=torch.int32, device=dev)
    fin_score = torch.zeros((B, W), dtype=torch.float32, device=dev)
    fin_len = torch.zeros((B, W), dtype=torch.int32, device=dev)
    fin_tok = torch.full((B, W, Lmax), pad, dtype=torch.long, device=dev)
    done_host = torch.zeros((max_length, B), dtype=torch.int32).pin_memory()
    new_tok = ids[:, :1].contiguous()
    flags = []                                 # (event, step) of the done-flag copies
    ev_ids = torch.cuda.Event()
    ev_ids.record(main)
    cur_len, steps = 1, 0
    w_att, w_ctc = float(1 - ctc_weight), float(ctc_weight)
    es_mode = _ES_MODE[early_stopping]
    t_loop = time.perf_counter()
    while cur_len < max_length:
        if len(flags) >= run_ahead:            # bounded run-ahead: wait for the flags of step (now - run_ahead) and stop if everything is done
            ev, t = flags[len(flags) - run_ahead]
            ev.synchronize()
            if bool(done_host[t].all()):
                break
        ctc = None
        if proc is not None:                   # prefix scores of this step on the side stream (needs the ids the previous step left)
            with torch.cuda.stream(side):
                side.wait_event(ev_ids)
                ctc = proc.ctc_scores(ids[:, :cur_len])
                ev_ctc = torch.cuda.Event()
                ev_ctc.record(side)
            ctc.record_stream(main)            # allocated on the side stream, read by the main stream below: the only tensor of the loop that crosses streams
        logits = joint.dec.step(new_tok, cache, kvs, T2, key_rep)                       # (B*W, V), row stride padded to 8
        lse = ops.row_lse(logits)
        if proc is not None:
            main.wait_event(ev_ctc)
        new_tok = torch.empty((n_bh, 1), dtype=torch.long, device=dev)
        beam_idx = torch.empty((n_bh,), dtype=torch.long, device=dev)
        top_s = top_i = None
        if trace is not None:
            top_s, top_i = torch.empty((B, 2 * W), device=dev), torch.empty((B, 2 * W), dtype=torch.int32, device=dev)
            trace.append((top_s, top_i, done.clone()))          # the done flags BEFORE the step
        denom, heur = _step_denoms(cur_len, max_length, length_penalty, early_stopping)
        _lib.check(L_.mi_beam_step(logits.data_ptr(), logits.stride(0), lse.data_ptr(), ctc.data_ptr() if ctc is not None else None, w_att, w_ctc, int(proc is not None), pad,
                                   eos_token_id, B, W, V, cur_len, max_length, Lmax, denom, heur, es_mode, ids.data_ptr(), beam_scores.data_ptr(), new_tok.data_ptr(),
                                   beam_idx.data_ptr(), done.data_ptr(), nfin.data_ptr(), fin_score.data_ptr(), fin_len.data_ptr(), fin_tok.data_ptr(),
                                   top_s.data_ptr() if top_s is not None else None, top_i.data_ptr() if top_i is not None else None, done_host[steps].data_ptr(),
                                   main.cuda_stream), "mi_beam_step")
        ev_ids = torch.cuda.Event()
        ev_ids.record(main)
        if W > 1:
            joint.dec.reorder_cache(cache, beam_idx)
        flags.append((ev_ids, steps))          # the flags of this step are in pinned memory once the step's event has fired
        cur_len += 1
        steps += 1
    if stats is not None:                      # host time spent enqueuing the token loop (the GPU may still be running it)
        stats["host_loop_ms"] = (time.perf_counter() - t_loop) * 1e3
        stats["steps"] = steps
    if side is not None:
        main.wait_stream(side)
    nfin_c, fs_c, fl_c, ft_c = nfin.cpu(), fin_score.cpu(), fin_len.cpu(), fin_tok.cpu()             # the first copy synchronises with everything enqueued
    out = []
    for b in range(B):                         # every utterance ends with kept hypotheses: at max_length the step's first W candidates all stop
        hyps = [(float(fs_c[b, k]), ft_c[b, k, :int(fl_c[b, k])].tolist()) for k in range(int(nfin_c[b]))]
        out.append(dict(tokens=hyps[0][1], score=hyps[0][0], hypotheses=hyps))
    return out


def generate_stepwise(joint: "JointAEDEngine", feats, feat_len, *, num_beams=1, max_length=64, ctc_weight=0.3, length_penalty=1.0, early_stopping=False,
                      eos_token_id=1, pad_token_id=None, start_token_id=None, space_token_id=-1, apply_eos_space_trick=False, eos_space_trick_weight=1.0):
    """The same decoding with the beam bookkeeping on the host, one token at a time (two device -> host copies and three host -> device copies per token): the form the
    reference's generate() has, kept as the cross-check of `generate` (tests/test_gpu_config5.py, tests/test_gpu_aed.py compare the two hypothesis for hypothesis) and as
    the route of the eos / space trick (the processor applies it, ctc_scorer.py:333-349)."""
    import numpy as np
    from .decoding import CTCRescorerLogitsProcessor
    _check_generate_args(num_beams, max_length, early_stopping)
    dev = joint.device
    c = joint.jcfg
    pad = c["pad_token_id"] if pad_token_id is None else pad_token_id
    start = c["decoder_start_token_id"] if start_token_id is None else start_token_id
    W = num_beams
    enc_out, enc_bf, T2, key_len = joint.encode(feats, feat_len)
    B = feats.shape[0]
    d = enc_bf.shape[1]
    # every beam attends to its utterance's encoder frames
    enc_rep = enc_bf.view(B, T2, d).repeat_interleave(W, 0).reshape(B * W * T2, d)
    key_rep = key_len.repeat_interleave(W) if key_len is not None else None
    kvs = joint.dec.cross_kv(enc_rep)
    cache = joint.dec.init_cache(B * W, max_length + 1)
    proc = None
    if ctc_weight > 0:
        lens = enc_out["outer_len"].clamp(max=T2)
        proc = CTCRescorerLogitsProcessor(enc_out["logits"], lens, pad, eos_token_id, 0, ctc_weight, W, space_token_id, bool(apply_eos_space_trick), eos_space_trick_weight)
    ids = torch.full((B * W, 1), start, dtype=torch.long, device=dev)
    beam_scores = torch.zeros((B, W), device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    kept = [[] for _ in range(B)]              # (score fp32, tokens), best first, at most W
    done = [False] * B
    new_tok = ids
    V = joint.dec.w["lm_head"].shape[0]
    NEG = np.float32(-1.0e9)
    while ids.shape[1] < max_length and not all(done):
        logits = joint.dec.step(new_tok, cache, kvs, T2, key_rep)                       # (B*W, V)
        scores = logits - ops.row_lse(logits.contiguous())[:, None]                      # log_softmax
        if proc is not None:
            scores = proc(ids, scores.clone())
        cand = (scores + beam_scores[:, None]).view(B, W * V)
        top_s, top_i = cand.topk(2 * W, dim=1)
        top_s, top_i = top_s.cpu().numpy(), top_i.cpu().numpy()
        cur_len = ids.shape[1]
        at_max = cur_len + 1 >= max_length
        denom, heur = (np.float32(v) for v in _step_denoms(cur_len, max_length, length_penalty, early_stopping))
        nb_scores = torch.zeros((B, W)); nb_tok = torch.zeros((B, W), dtype=torch.long); nb_idx = torch.zeros((B, W), dtype=torch.long)
        ids_cpu = ids.cpu()
        for b in range(B):
            if done[b]:
                nb_scores[b] = 0; nb_tok[b] = pad; nb_idx[b] = b * W
                continue
            rows = [(np.float32(top_s[b, r]), int(top_i[b, r]) // V, int(top_i[b, r]) % V) for r in range(2 * W)]
            hit = [tok == eos_token_id or at_max for _, _, tok in rows]
            nxt = [(s, beam, tok) for (s, beam, tok), h in zip(rows, hit) if not h][:W]
            nxt += [(np.float32(s + NEG), beam, tok) for (s, beam, tok), h in zip(rows, hit) if h][:W - len(nxt)]
            for k, (s, beam, tok) in enumerate(nxt):
                nb_scores[b, k], nb_tok[b, k], nb_idx[b, k] = float(s), tok, b * W + beam
            for r in range(W):
                if hit[r]:
                    s, beam, tok = rows[r]
                    sc = np.float32(s / denom)
                    pos = len(kept[b])
                    while pos > 0 and sc > kept[b][pos - 1][0]:
                        pos -= 1
                    if pos < W:
                        kept[b].insert(pos, (sc, ids_cpu[b * W + beam].tolist() + [tok]))
                        del kept[b][W:]
            best = np.float32(nxt[0][0] / heur)
            unsat = best > (kept[b][W - 1][0] if len(kept[b]) == W else NEG)
            if (not unsat) or (early_stopping is True and len(kept[b]) == W) or at_max:
                done[b] = True
        beam_idx = nb_idx.view(-1).to(dev)
        new_tok = nb_tok.view(-1, 1).to(dev)
        beam_scores = nb_scores.view(-1).to(dev)
        ids = torch.cat([ids.index_select(0, beam_idx), new_tok], 1)
        joint.dec.reorder_cache(cache, beam_idx)
    return [dict(tokens=kept[b][0][1], score=float(kept[b][0][0]), hypotheses=[(float(s), t) for s, t in kept[b]]) for b in range(B)]
