"""Training step of the E-Branchformer encoder + CTC head on the HIP path.

Replaces, for this model, what the reference gets from torch autograd + HF Trainer around `Wav2Vec2EBranchformerForCTC.forward`
(src/models/encoders/e_branchformer.py:422-496; trainer src/utilities/training_utils.py:93-115; recipe hyper-parameters
recipes/librispeech/.../train_small_baseline.sh:43,53-58: bf16 autocast, AdamW, clip 1.0):

  * `ParamStore`   — ONE flat fp32 master buffer (+ flat gradient, Adam moments, bf16 mirror the GEMMs read, and K-major
                     transposed bf16 copies for the dX GEMMs), parameters kept in the packed layouts the kernels want;
                     import / export in the reference's state-dict names.
  * `EncoderCTCTrainer.forward_backward` — forward with saved activations, analytic backward on the HIP kernels of
                     csrc/{train_ops,bgemm,attn_bwd,conv_bwd,loss_bwd}.hip + the forward GEMM kernel (dX = dY·W, dW = dYᵀ·X).
  * `GradSync`     — data-parallel gradient SUM all-reduce (RCCL through torch.distributed): one merged collective after the backward by
                     default, per layer bucket while the earlier layers' backward is still running with overlap=True; the 1/world
                     factor is folded into the loss gradient.
  * `AdamW`        — one fused kernel over the flat buffer (grad-norm clip coefficient and the skip-this-step flag read on device,
                     bf16 mirror refreshed).

Precision model = the reference's autocast recipe: fp32 master weights and residual stream, bf16 GEMM operands (activations AND
activation gradients), fp32 accumulation, fp32 LayerNorm / softmax / CTC, fp32 parameter gradients.
Dropout (all eight sites of the layer, encoder input, feature projection, CTC head) uses counter-based masks that the backward
pass regenerates; in-model SpecAugment draws its masks on the host with transformers' own `_compute_mask_indices` (numpy RNG), exactly as the
reference does (LayerDrop included).  Causal (streaming) encoders train on the same path (left-padded front end, causal attention mask,
the reference's dilated causal CSGU conv).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch

from . import ops
from . import ops_train as T
from .shapes import GATE_SHARE, context_mode, conv_freq_out, gate_geometry

BF16, F32 = torch.bfloat16, torch.float32


# ====================================================================================================== parameter store
@dataclass
class Spec:
    name: str
    shape: tuple
    mat: bool          # bf16 (N,K) GEMM operand -> also keeps the transposed bf16 copy
    decay: bool


def _al(n, a=64):
    return (n + a - 1) // a * a


class ParamStore:
    def __init__(self, specs: list[Spec], device):
        self.specs = {s.name: s for s in specs}
        self.order = [s.name for s in specs]
        self.device = torch.device(device)
        off, offT = 0, 0
        self.off, self.offT = {}, {}
        for s in specs:
            self.off[s.name] = off
            off += _al(math.prod(s.shape))
            if s.mat:
                N, K = s.shape
                self.offT[s.name] = offT
                offT += _al(K * _al(N))
        self.n = off
        z = lambda dt, n=off: torch.zeros(n, dtype=dt, device=self.device)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = z(F32), z(F32), z(F32), z(F32)
        self.flat_bf = z(BF16)
        self.flat_T = z(BF16, max(offT, 64))
        self.decay = torch.zeros(off, dtype=torch.uint8, device=self.device)
        self.frozen_spans = []
        self.set_frozen(())
        self.step_count = 0

    def set_frozen(self, names):
        """Parameters that must not change: no weight decay on them (the mask AdamW reads), and their gradient spans — merged into as few
        contiguous ranges as the layout allows — are cleared before the norm / AdamW (`zero_frozen_grads`), so that with m = v = 0 the update
        is exactly zero and the clipping norm counts trainable parameters only (torch's optimizers never see a frozen parameter)."""
        frozen = set(names)
        mask = torch.zeros(self.n if hasattr(self, "n") else self.decay.numel(), dtype=torch.uint8)
        spans = []
        for name in self.order:
            s, o = self.specs[name], self.off[name]
            if name in frozen:
                hi = o + _al(math.prod(s.shape))
                if spans and spans[-1][1] == o:
                    spans[-1][1] = hi
                else:
                    spans.append([o, hi])
            elif s.decay:
                mask[o:o + math.prod(s.shape)] = 1
        self.decay.copy_(mask)
        self.frozen_spans = [(lo, hi) for lo, hi in spans]

    def zero_frozen_grads(self):
        for lo, hi in self.frozen_spans:
            self.flat_g[lo:hi].zero_()

    def _view(self, flat, name):
        s = self.specs[name]
        o = self.off[name]
        return flat[o:o + math.prod(s.shape)].view(*s.shape)

    def p(self, name): return self._view(self.flat_p, name)
    def g(self, name): return self._view(self.flat_g, name)
    def bf(self, name): return self._view(self.flat_bf, name)

    def bfT(self, name):
        """(K, pad64(N)) bf16 transposed copy of the (N,K) matrix (columns N.. zero)."""
        N, K = self.specs[name].shape
        o = self.offT[name]
        return self.flat_T[o:o + K * _al(N)].view(K, _al(N))

    def range_of(self, names):
        lo = min(self.off[n] for n in names)
        hi = max(self.off[n] + _al(math.prod(self.specs[n].shape)) for n in names)
        return lo, hi

    def refresh_mirrors(self, cast=True):
        """bf16 mirror (if not already written by the optimizer kernel) + transposed copies of every matrix (ONE launch for all of them)."""
        if cast:
            ops_cast_flat(self.flat_p, self.flat_bf)
        if getattr(self, "_tr_descs", None) is None:
            import numpy as np
            rec = np.zeros(len(self.offT), dtype=np.dtype([("in", "<u8"), ("out", "<u8"), ("M", "<i4"), ("N", "<i4"), ("Mp", "<i4"), ("pad", "<i4")]))
            ok = self.device.type == "cuda"
            for i, name in enumerate(self.offT):
                N, K = self.specs[name].shape
                rec[i] = (self.bf(name).data_ptr(), self.bfT(name).data_ptr(), N, K, _al(N), 0)
                ok = ok and K % 8 == 0
            self._tr_descs = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device) if ok and len(rec) else False
            self._tr_count = len(rec)
        if self._tr_descs is not False:
            ops._lib.check(ops._lib.lib().mi_transpose_many_bf16(self._tr_descs.data_ptr(), self._tr_count, torch.cuda.current_stream().cuda_stream),
                           "mi_transpose_many_bf16")
            return
        for name, s in self.specs.items():
            if s.mat:
                N, K = s.shape
                T.transpose(self.bf(name), Mp=_al(N), out=self.bfT(name))

    def zero_grad(self):
        self.flat_g.zero_()
        self.fresh = True               # every gradient is exactly zero: the next backward's grouped weight-gradient launches may WRITE their targets (EncoderCTCTrainer.dw_overwrite)


def ops_cast_flat(src_f32, dst_bf16):
    n = src_f32.numel()
    w = 4096
    rows = n // w
    if rows:
        ops_cast = ops._lib.lib().mi_cast_f32_bf16
        ops._lib.check(ops_cast(src_f32.data_ptr(), w, dst_bf16.data_ptr(), w, rows, w, torch.cuda.current_stream().cuda_stream), "mi_cast_f32_bf16")
    rem = n - rows * w
    if rem:
        ops._lib.check(ops._lib.lib().mi_cast_f32_bf16(src_f32[rows * w:].data_ptr(), rem, dst_bf16[rows * w:].data_ptr(), rem, 1, rem,
                                                       torch.cuda.current_stream().cuda_stream), "mi_cast_f32_bf16")


# ====================================================================================================== encoder parameters
def encoder_specs(c: dict, head: bool = True) -> list[Spec]:
    d, I, L, V1 = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"], c["vocab_size"] + 1
    C1, C2 = c["conv_dim"]
    K = c["conv_kernel"][0]
    F2 = conv_freq_out(c.get("num_fbanks", 80), c["conv_kernel"], c["conv_stride"], c["conv_padding"])
    kc, km = c.get("csgu_kernel_size", 31), c.get("merge_conv_kernel", 31)
    rel = c.get("position_embeddings_type", "relative") == "relative"
    S = []
    mat = lambda n, *sh: S.append(Spec(n, tuple(sh), True, True))
    vec = lambda n, *sh, decay=False: S.append(Spec(n, tuple(sh), False, decay))
    vec("masked_spec_embed", d)     # SpecAugment fill vector: carried for state-dict parity, never receives a gradient on this path (torch skips it too)
    mode = context_mode(c)                                   # context-aware front end (extractors.py:23-65): 0 plain, 1 gated, 2 gated_shared
    gkh = K * (GATE_SHARE if mode == 2 else 1)               # the shared gate's kernel is (4K, K)
    vec("conv1_w", C1, K * K, decay=True); vec("conv1_b", C1)
    if mode:
        vec("gate1_w", C1, gkh * K, decay=True); vec("gate1_b", C1)
    if mode == 1:                                            # conv rows, then gate rows: ONE implicit GEMM forward, one dW / dX GEMM pair backward
        mat("conv2_w", 2 * C2, K * K * C1); vec("conv2_b", 2 * C2)
    else:
        mat("conv2_w", C2, K * K * C1); vec("conv2_b", C2)
    if mode == 2:
        mat("gate2_w", C2, gkh * K * C1); vec("gate2_b", C2)
    mat("feout_w", d, F2 * C2); vec("feout_b", d)
    vec("fp_ln_g", d); vec("fp_ln_b", d); mat("fp_w", d, d); vec("fp_b", d)
    for l in range(L + int(bool(c.get("finetune_with_additional_layer", False)))):     # layer L = the fine-tuning head's `additional_layer` (bestrq.py:199-200)
        p = f"l{l}."
        ffs = ("ff1", "ff2") if c.get("use_macaron_ff", True) else ()
        for ff in ffs[:1]:
            vec(p + ff + "_ln_g", d); vec(p + ff + "_ln_b", d); mat(p + ff + "_w1", I, d); vec(p + ff + "_b1", I); mat(p + ff + "_w2", d, I); vec(p + ff + "_b2", d)
        vec(p + "att_ln_g", d); vec(p + "att_ln_b", d)
        mat(p + "att_wqkv", 3 * d, d); vec(p + "att_bqkv", 3 * d); mat(p + "att_wo", d, d); vec(p + "att_bo", d)
        if rel:
            mat(p + "att_wpos", d, d); vec(p + "att_u", d); vec(p + "att_v", d)          # pos_bias_* : "bias" in the name -> no decay
        vec(p + "mlp_ln_g", d); vec(p + "mlp_ln_b", d); mat(p + "mlp_w1", I, d); vec(p + "mlp_b1", I)
        vec(p + "csgu_ln_g", I // 2); vec(p + "csgu_ln_b", I // 2); vec(p + "csgu_w", I // 2, kc, decay=True); vec(p + "csgu_b", I // 2)
        if c.get("csgu_use_linear_after_conv", False):
            mat(p + "csgu_lin_w", I // 2, I // 2); vec(p + "csgu_lin_b", I // 2)
        mat(p + "mlp_w2", d, I // 2); vec(p + "mlp_b2", d)
        vec(p + "mrg_dw_w", 2 * d, km, decay=True); vec(p + "mrg_dw_b", 2 * d); mat(p + "mrg_w", d, 2 * d); vec(p + "mrg_b", d)
        for ff in ffs[1:]:
            vec(p + ff + "_ln_g", d); vec(p + ff + "_ln_b", d); mat(p + ff + "_w1", I, d); vec(p + ff + "_b1", I); mat(p + ff + "_w2", d, I); vec(p + ff + "_b2", d)
        vec(p + "fin_ln_g", d); vec(p + "fin_ln_b", d)
    vec("enc_ln_g", d); vec("enc_ln_b", d)
    if c.get("finetune_with_layer_mixing", False):
        vec("mix_w", L + 1, decay=True)          # `per_layer_weights` (bestrq.py:202-205): a plain nn.Parameter, so weight decay applies
    if head:
        mat("head_w", V1, d); vec("head_b", V1)
    return S


def _enc_map(c: dict, head: bool = True):
    """packed name -> (to_packed(sd) -> tensor, [(reference key, from_packed(tensor) -> tensor), ...])"""
    d, L = c["hidden_size"], c["num_hidden_layers"]
    C1, C2 = c["conv_dim"]
    K = c["conv_kernel"][0]
    V = c["vocab_size"]
    F2 = conv_freq_out(c.get("num_fbanks", 80), c["conv_kernel"], c["conv_stride"], c["conv_padding"])
    kc, km = c.get("csgu_kernel_size", 31), c.get("merge_conv_kernel", 31)
    fe, fp = "wav2vec2.feature_extractor.", "wav2vec2.feature_projection."
    cw = "" if c.get("is_causal", False) else ".conv"
    m = {}
    one = lambda name, key, fwd=lambda t: t, bwd=lambda t: t: m.__setitem__(name, (lambda sd: fwd(sd[key]), [(key, bwd)]))
    # optional in the reference (present iff mask_time_prob > 0 or mask_feature_prob > 0): absent -> zeros, not exported
    m["masked_spec_embed"] = (lambda sd: sd["wav2vec2.masked_spec_embed"] if "wav2vec2.masked_spec_embed" in sd else torch.zeros(d),
                              [("wav2vec2.masked_spec_embed", lambda t: t)])
    mode = context_mode(c)
    if mode:        # ContextAwareConv2d.conv is a Gated* module: keys ...conv.N.0.conv.{conv,gate}.{weight,bias} (extractors.py:23-54)
        gkh = K * (GATE_SHARE if mode == 2 else 1)
        c1, c2 = f"{fe}conv.0.0.conv.", f"{fe}conv.1.0.conv."
        cl = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)                                  # (Cout, Cin, KH, KW) -> (Cout, (kh, kw, cin))
        uncl = lambda kh: (lambda t: t.reshape(C2, kh, K, C1).permute(0, 3, 1, 2))
        one("conv1_w", c1 + "conv.weight", lambda t: t.reshape(C1, K * K), lambda t: t.reshape(C1, 1, K, K))
        one("conv1_b", c1 + "conv.bias")
        one("gate1_w", c1 + "gate.weight", lambda t: t.reshape(C1, gkh * K), lambda t: t.reshape(C1, 1, gkh, K))
        one("gate1_b", c1 + "gate.bias")
        if mode == 1:
            m["conv2_w"] = (lambda sd: torch.cat([cl(sd[c2 + "conv.weight"]), cl(sd[c2 + "gate.weight"])], 0),
                            [(c2 + "conv.weight", lambda t: uncl(K)(t[:C2])), (c2 + "gate.weight", lambda t: uncl(K)(t[C2:]))])
            m["conv2_b"] = (lambda sd: torch.cat([sd[c2 + "conv.bias"], sd[c2 + "gate.bias"]], 0),
                            [(c2 + "conv.bias", lambda t: t[:C2]), (c2 + "gate.bias", lambda t: t[C2:])])
        else:
            one("conv2_w", c2 + "conv.weight", cl, uncl(K)); one("conv2_b", c2 + "conv.bias")
            one("gate2_w", c2 + "gate.weight", cl, uncl(gkh)); one("gate2_b", c2 + "gate.bias")
    else:
        one("conv1_w", f"{fe}conv.0.0{cw}.weight", lambda t: t.reshape(C1, K * K), lambda t: t.reshape(C1, 1, K, K))
        one("conv1_b", f"{fe}conv.0.0{cw}.bias")
        one("conv2_w", f"{fe}conv.1.0{cw}.weight", lambda t: t.permute(0, 2, 3, 1).reshape(C2, K * K * C1),
            lambda t: t.reshape(C2, K, K, C1).permute(0, 3, 1, 2))
        one("conv2_b", f"{fe}conv.1.0{cw}.bias")
    one("feout_w", fe + "out.weight", lambda t: t.reshape(d, C2, F2).permute(0, 2, 1).reshape(d, F2 * C2),
        lambda t: t.reshape(d, F2, C2).permute(0, 2, 1).reshape(d, C2 * F2))
    one("feout_b", fe + "out.bias")
    one("fp_ln_g", fp + "layer_norm.weight"); one("fp_ln_b", fp + "layer_norm.bias")
    one("fp_w", fp + "projection.weight"); one("fp_b", fp + "projection.bias")
    one("enc_ln_g", "wav2vec2.encoder.layer_norm.weight"); one("enc_ln_b", "wav2vec2.encoder.layer_norm.bias")
    if head:
        m["head_w"] = (lambda sd: torch.cat([sd["lm_head.weight"], sd["blank_projection.weight"]], 0),
                       [("lm_head.weight", lambda t: t[:V]), ("blank_projection.weight", lambda t: t[V:])])
        m["head_b"] = (lambda sd: torch.cat([sd["lm_head.bias"], sd["blank_projection.bias"]], 0),
                       [("lm_head.bias", lambda t: t[:V]), ("blank_projection.bias", lambda t: t[V:])])
    if c.get("finetune_with_layer_mixing", False):
        one("mix_w", "per_layer_weights")
    for l in range(L + int(bool(c.get("finetune_with_additional_layer", False)))):
        p, r = f"l{l}.", (f"wav2vec2.encoder.layers.{l}." if l < L else "additional_layer.")
        if c.get("use_macaron_ff", True):
            for ff in ("ff1", "ff2"):
                one(p + ff + "_ln_g", r + ff + ".0.weight"); one(p + ff + "_ln_b", r + ff + ".0.bias")
                one(p + ff + "_w1", r + ff + ".1.intermediate_dense.weight"); one(p + ff + "_b1", r + ff + ".1.intermediate_dense.bias")
                one(p + ff + "_w2", r + ff + ".1.output_dense.weight"); one(p + ff + "_b2", r + ff + ".1.output_dense.bias")
        one(p + "att_ln_g", r + "self_attn_layer_norm.weight"); one(p + "att_ln_b", r + "self_attn_layer_norm.bias")
        a = r + "self_attn."
        m[p + "att_wqkv"] = (lambda sd, a=a: torch.cat([sd[a + f"linear_{n}.weight"] for n in "qkv"], 0),
                             [(a + f"linear_{n}.weight", (lambda t, i=i: t[i * d:(i + 1) * d])) for i, n in enumerate("qkv")])
        m[p + "att_bqkv"] = (lambda sd, a=a: torch.cat([sd[a + f"linear_{n}.bias"] for n in "qkv"], 0),
                             [(a + f"linear_{n}.bias", (lambda t, i=i: t[i * d:(i + 1) * d])) for i, n in enumerate("qkv")])
        one(p + "att_wo", a + "linear_out.weight"); one(p + "att_bo", a + "linear_out.bias")
        if c.get("position_embeddings_type", "relative") == "relative":
            H = c["num_attention_heads"]
            one(p + "att_wpos", a + "linear_pos.weight")
            one(p + "att_u", a + "pos_bias_u", lambda t: t.reshape(d), lambda t: t.reshape(H, d // H))
            one(p + "att_v", a + "pos_bias_v", lambda t: t.reshape(d), lambda t: t.reshape(H, d // H))
        one(p + "mlp_ln_g", r + "cgMLP_layer_norm.weight"); one(p + "mlp_ln_b", r + "cgMLP_layer_norm.bias")
        g = r + "cgMLP."
        one(p + "mlp_w1", g + "channel_proj1.0.weight"); one(p + "mlp_b1", g + "channel_proj1.0.bias")
        one(p + "csgu_ln_g", g + "csgu.norm.weight"); one(p + "csgu_ln_b", g + "csgu.norm.bias")
        one(p + "csgu_w", g + "csgu.conv.weight", lambda t: t.reshape(-1, kc), lambda t: t.reshape(-1, 1, kc))
        one(p + "csgu_b", g + "csgu.conv.bias")
        if c.get("csgu_use_linear_after_conv", False):
            one(p + "csgu_lin_w", g + "csgu.linear.weight"); one(p + "csgu_lin_b", g + "csgu.linear.bias")
        one(p + "mlp_w2", g + "channel_proj2.weight"); one(p + "mlp_b2", g + "channel_proj2.bias")
        one(p + "mrg_dw_w", r + "depthwise_conv_fusion.weight", lambda t: t.reshape(-1, km), lambda t: t.reshape(-1, 1, km))
        one(p + "mrg_dw_b", r + "depthwise_conv_fusion.bias")
        one(p + "mrg_w", r + "merge_proj.weight"); one(p + "mrg_b", r + "merge_proj.bias")
        one(p + "fin_ln_g", r + "final_layer_norm.weight"); one(p + "fin_ln_b", r + "final_layer_norm.bias")
    return m


# ====================================================================================================== gradient sync (DP)
class GradSync:
    """Data-parallel SUM all-reduce of the flat gradient.  RCCL over xGMI on MI355X (backend 'nccl'), gloo in the CPU tests.

    Two schedules.  Default (`overlap=False`): the ranges reported by `launch()` are merged and reduced by ONE collective over the
    contiguous span when the backward is done (`wait()`), ordered on the compute stream — 516 MB of fp32 gradients for the 129 M-parameter base
    model, of the order of 1–3 ms over xGMI against a 35 ms step.  `overlap=True` (or HFASR_DP_OVERLAP=1): each range is
    all-reduced as soon as it is final (reverse layer order), the collective of layer l running beside the backward of layers < l.
    That puts RCCL's kernels on the same CUs as the LDS-DMA GEMMs; a kernel of ours sharing a CU with those GEMMs from another
    stream produced wrong values on MI355X ('Concurrent kernels': a packed-f32 VALU instruction loses a product; our library is built without packed
    ops, librccl is not ours to rebuild), so co-scheduling stays opt-in and warns on the nccl backend."""

    def __init__(self, flat_g: torch.Tensor, group=None, enabled=True, overlap=None):
        import torch.distributed as dist
        self.dist = dist
        self.on = bool(enabled) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.on else 1
        self.flat_g, self.group, self.pending = flat_g, group, []
        self.overlap = (os.environ.get("HFASR_DP_OVERLAP", "0") == "1") if overlap is None else bool(overlap)
        self._span = None
        if self.on and self.overlap and dist.get_backend(group) == "nccl":
            import warnings
            warnings.warn("GradSync(overlap=True) on RCCL: the all-reduce kernels (librccl's gfx950 code objects contain v_pk_{add,mul,fma}_f32) then run beside the "
                          "backward's LDS-DMA GEMMs on shared CUs; a packed-f32 kernel in that position lost products on MI355X (DESIGN.md 'Concurrent kernels', "
                          "tools/dbg/README.md).  Unverified pairing: gradients may be silently wrong; the default (one collective after the backward) avoids it.",
                          RuntimeWarning, stacklevel=2)

    def launch(self, lo: int, hi: int):
        if not self.on or hi <= lo:
            return
        if self.overlap:
            self.pending.append(self.dist.all_reduce(self.flat_g[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._span = (lo, hi) if self._span is None else (min(lo, self._span[0]), max(hi, self._span[1]))

    def wait(self):
        for h in self.pending:
            h.wait()
        self.pending = []
        if self._span is not None:
            lo, hi = self._span
            self._span = None
            self.dist.all_reduce(self.flat_g[lo:hi], op=self.dist.ReduceOp.SUM, group=self.group)


def _u01(seed: int, stream: int) -> float:
    """one uniform number in [0, 1) from the counter-based generator of the dropout masks (splitmix64 of (stream << 32) ^ seed)"""
    m = (1 << 64) - 1
    z = ((((stream & 0xFFFFFFFF) << 32) ^ (seed & 0xFFFFFFFF)) + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return (z >> 11) / float(1 << 53)


# ====================================================================================================== trainer
class EncoderCTCTrainer:
    """forward + backward + AdamW for Wav2Vec2EBranchformerForCTC on one GPU (one process per GPU under DP)."""

    def __init__(self, cfg: dict, device="cuda:0", *, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 dp_sync=True, seed=0, head=True, grad_norm_skip=0.0):
        """grad_norm_skip: > 0: a step whose global gradient norm exceeds it is dropped — parameters and moments untouched — which is what
        GradAwareTrainer does with its fixed threshold of 100 (training_utils.py:81,101-115; pass 100.0 to mirror it).  Off by default: on
        randomly initialised models the norm sits above 100 and every step would be dropped silently.  A non-finite norm always drops the step.
        `optimizer_step` leaves the decision in `self.last_step_flags` ([norm, clip coefficient, skipped] on the device).
        dp_sync=False: no gradient all-reduce here (the caller, e.g. HF Trainer's DDP wrapper, owns data parallelism).
        seed: dropout mask seed (masks are counter-based: f(seed, step, layer, site, element), csrc/dropout.hip); give every DP rank its own."""
        c = self.cfg = dict(cfg)
        self.causal = bool(c.get("is_causal", False))               # streaming encoder: left-padded front end, triu attention mask, dilated causal CSGU conv
        if len(c["conv_dim"]) != 2 or c["conv_kernel"][0] != 3 or len(set(c["conv_kernel"])) != 1:
            raise NotImplementedError("training path: 2-layer 3x3 Conv2d sub-sampling only")
        from .engine import ACT
        if c.get("csgu_activation", "identity") not in ACT:
            raise NotImplementedError(f"csgu_activation {c['csgu_activation']}")
        self.csgu_act = ACT[c.get("csgu_activation", "identity")]
        self.csgu_lin = bool(c.get("csgu_use_linear_after_conv", False))
        # fused CSGU kernels cover the reference recipes' form (identity activation, no Linear); anything else runs split: conv -> [Linear] -> act * gate
        self.csgu_split = self.csgu_lin or self.csgu_act != 0
        self.dual_ln = True                       # the two branch norms' backward in one pass (tools/train_bench.py --no-dual-ln measures the two-pass form beside it)
        self.dw_overwrite = os.environ.get("HFASR_DW_OVERWRITE", "1") != "0"      # see _forward_backward (HFASR_DW_OVERWRITE=0 / tools/train_bench.py --no-dw-overwrite: always accumulate)
        self.ctc_from_bwd = True                  # the CTC loss out of the backward's own alpha recursion (tools/train_bench.py --no-ctc-from-bwd: forward loss kernel + backward)
        self.walk_qb = True                       # q + u / q + v of the attention backward from the fused walk's prologue (--no-walk-qb: the pass of their own)
        self.sparse_attn_bwd = os.environ.get("HFASR_ATTN_BWD_SPARSE", "1") != "0"       # round 5: the fused walk does not write the zeros nobody reads (ops_train.attn_bwd_probs)
        self.frozen = set()
        self.layerdrop = float(c.get("layerdrop", 0.0) or 0.0)      # tf:models/wav2vec2_conformer/modeling_wav2vec2_conformer.py:686-690
        g = lambda k: float(c.get(k, 0.0) or 0.0)
        # dropout probabilities by site (which config value feeds which nn.Dropout: e_branchformer.py:132,182,229-246,451; tf :353,356,674)
        self.pdrop = dict(act=g("activation_dropout"), hidden=g("hidden_dropout"), att=g("attention_dropout"), csgu=g("csgu_conv_dropout"),
                          final=g("final_dropout"), fp=g("feat_proj_dropout"))
        self.seed = int(seed) & 0xFFFFFFFF
        self.train_steps_seen = 0
        self.specaug = bool(c.get("apply_spec_augment", False)) and (float(c.get("mask_time_prob", 0.0) or 0.0) > 0.0 or
                                                                     float(c.get("mask_feature_prob", 0.0) or 0.0) > 0.0)
        self.device = torch.device(device)
        self.head = bool(head)            # False: bare encoder (BEST-RQ pre-training puts its own classifier on the last hidden state)
        self.store = ParamStore(encoder_specs(c, self.head), self.device)
        self.map = _enc_map(c, self.head)
        self.hp = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm, grad_norm_skip=grad_norm_skip)
        self.sync = GradSync(self.store.flat_g, group, enabled=dp_sync)
        self._ranges_waiting = []                 # gradient ranges whose weight-gradient GEMMs are still recorded (ops_train.TnBatch), see _range_done
        self._pos = {}
        self._scal = torch.zeros(4, dtype=F32, device=self.device)       # [sumsq, norm, coef, -]
        L = c["num_hidden_layers"]
        # CTC fine-tuning head of a BEST-RQ encoder (bestrq.py:192-279): softmax-weighted mix of all hidden states and / or one more layer on top
        self.mix = bool(c.get("finetune_with_layer_mixing", False))
        self.extra = bool(c.get("finetune_with_additional_layer", False))
        if (self.mix or self.extra) and not self.head:
            raise NotImplementedError("layer mixing / additional layer belong to the CTC fine-tuning head")
        names = self.store.order
        self._layer_names = [[n for n in names if n.startswith(f"l{l}.")] for l in range(L + int(self.extra))]
        self._front_names = [n for n in names if n.startswith(("masked_spec", "conv", "gate", "feout", "fp_"))]
        self.ctx_mode = context_mode(c)                              # context-aware Conv2d front end (extractors.py:23-65): 0 plain, 1 gated, 2 gated_shared
        self._encln_names = ["enc_ln_g", "enc_ln_b"] + (["mix_w"] if self.mix else [])
        self._head_names = self._encln_names + (["head_w", "head_b"] if self.head else [])

    # ------------------------------------------------------------------ weights in / out
    def load_state_dict(self, sd: dict):
        dev = self.device
        sdd = {k: v.detach().to(dev, F32) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
        self._has_mse = "wav2vec2.masked_spec_embed" in sdd
        for name in self.store.order:
            self.store.p(name).copy_(self.map[name][0](sdd).reshape(self.store.specs[name].shape))
        self.store.refresh_mirrors(cast=True)

    def _export(self, view):
        out = {}
        for name in self.store.order:
            if name == "masked_spec_embed" and not getattr(self, "_has_mse", True):
                continue
            t = view(name)
            for key, fn in self.map[name][1]:
                out[key] = fn(t).clone(memory_format=torch.contiguous_format)
        return out

    def state_dict(self) -> dict:
        return self._export(self.store.p)

    def alias_views(self, which: str = "p", prefix: str = "") -> dict:
        """reference name -> a tensor that ALIASES the flat parameter ("p") or gradient ("g") store in the reference's shape (possibly strided: conv2's
        channels-last weight is a permuted view), for every piece whose reference layout is a view of the packed layout; None where it is not
        (the front end's `out` Linear, whose columns are re-ordered): those pieces are copied.  The autograd bridge makes the model's nn.Parameters
        these views, so the HF route neither imports the state dict nor exports gradients per step (autograd_bridge.py)."""
        flat = self.store.flat_p if which == "p" else self.store.flat_g
        base = flat.untyped_storage().data_ptr()
        out = {}
        for name in self.store.order:
            if name == "masked_spec_embed" and not getattr(self, "_has_mse", True):
                continue
            t = self.store._view(flat, name)
            for key, fn in self.map[name][1]:
                v = fn(t)
                out[prefix + key] = v if v.untyped_storage().data_ptr() == base else None
        return out

    def stores(self):
        return [self.store]

    def import_piece(self, key: str, value: torch.Tensor):
        """copy ONE reference tensor into its (non-aliasable) packed slot"""
        for name in self.store.order:
            if len(self.map[name][1]) == 1 and self.map[name][1][0][0] == key:
                self.store.p(name).copy_(self.map[name][0]({key: value.detach().to(self.device, F32)}).reshape(self.store.specs[name].shape))
                return
        raise KeyError(key)

    def export_grad_piece(self, key: str) -> torch.Tensor:
        for name in self.store.order:
            for k, fn in self.map[name][1]:
                if k == key:
                    return fn(self.store.g(name)).clone(memory_format=torch.contiguous_format)
        raise KeyError(key)

    def import_grad_piece(self, key: str, grad: torch.Tensor):
        """the gradient of ONE (non-aliasable) reference tensor into its packed gradient slot (optim.StoreAdamW: autograd owns that `.grad`, the step reads the flat store)"""
        for name in self.store.order:
            if len(self.map[name][1]) == 1 and self.map[name][1][0][0] == key:
                self.store.g(name).copy_(self.map[name][0]({key: grad.detach().to(self.device, F32)}).reshape(self.store.specs[name].shape))
                return
        raise KeyError(key)

    def export_piece(self, key: str) -> torch.Tensor:
        """the current master value of ONE reference tensor, in the reference layout"""
        for name in self.store.order:
            for k, fn in self.map[name][1]:
                if k == key:
                    return fn(self.store.p(name)).clone(memory_format=torch.contiguous_format)
        raise KeyError(key)

    def set_frozen(self, reference_names):
        """Names (reference state-dict keys) of parameters that do not train (`requires_grad False`: `freeze_encoder()`, train_ctc_asr.py:51-52).
        A packed store parameter counts as frozen when all of its reference pieces are; the weight / bias gradient GEMMs of frozen linears
        are then skipped in the backward (their input gradients are still computed: something upstream may train).  On the native route
        (`train_step`) frozen parameters stay bit-identical: no weight decay, gradients cleared before the clip norm and AdamW."""
        ref = set(reference_names or ())
        frozen = {name for name in self.store.order if name in self.map and self.map[name][1] and all(k in ref for k, _ in self.map[name][1])}
        if frozen != self.frozen:                 # the autograd bridge calls this every step: rebuild the device mask only when the set changes
            self.store.set_frozen(frozen)         # native route: AdamW leaves them bit-identical (no decay, zero gradient), the clip norm skips them
        self.frozen = frozen

    def _range_done(self, lo, hi, final=False):
        """the gradients of [lo, hi) are final once the deferred LayerNorm reductions and weight-gradient GEMMs have run: flush them, then hand the range to the
        data-parallel all-reduce.  The weight-gradient batch may decline a non-final flush (too few output tiles to fill the chip: ops_train.TnBatch) — the range
        then waits, with any earlier ones, for the flush that does run; the backward's last range is `final`."""
        pending = self._ranges_waiting
        pending.append((lo, hi))
        tnb = getattr(self, "_tnb", None)
        if tnb is None or tnb.flush(final=final):
            if getattr(self, "_lnred", None) is not None:       # the LayerNorm reductions ride the same cadence (a full batch flushes itself): fewer, fuller launches
                self._lnred.flush()
            for r in pending:
                self.sync.launch(*r)
            pending.clear()

    def _dwred(self):
        """the batch the depthwise-conv backward passes defer their tap-gradient reductions to (the LayerNorm one: flushed in _range_done, before a range is handed on)"""
        if getattr(self, "_lnred", None) is None:
            self._lnred = T.LnReduceBatch(self.device)
        return self._lnred

    def _lng(self, gname, bname):
        """gradient targets of a LayerNorm's affine pair: none when both are frozen (the cross-row reduction is then skipped)"""
        if gname in self.frozen and bname in self.frozen:
            return dict(dgamma=None, dbeta=None)
        if getattr(self, "_lnred", None) is None:
            self._lnred = T.LnReduceBatch(self.device)        # the (dgamma | dbeta) reductions of a layer's LayerNorms: one launch per flush, not one per LayerNorm
        return dict(dgamma=self.store.g(gname), dbeta=self.store.g(bname), defer=self._lnred)

    def grad_dict(self) -> dict:
        """gradients in the reference's parameter names / shapes (tests, checkpoint tooling, the autograd bridge)."""
        return self._export(self.store.g)

    # ------------------------------------------------------------------ tables
    def out_frames(self, Tn):
        c = self.cfg
        k, s, p = c["conv_kernel"][0], c["conv_stride"][0], c["conv_padding"][0]
        out = []
        for _ in range(2):
            Tn = (Tn + 2 * p - k) // s + 1
            out.append(Tn)
        return out

    def _pos_table(self, T2):
        c = self.cfg
        ptype = c.get("position_embeddings_type", "relative")
        key = (ptype, T2)
        if key not in self._pos:
            d, H = c["hidden_size"], c["num_attention_heads"]
            if ptype == "relative":
                pos = torch.arange(T2 - 1, -T2, -1, dtype=F32)[:, None]
                div = torch.exp(torch.arange(0, d, 2, dtype=torch.int64).float() * -(math.log(10000.0) / d))
                pe = torch.zeros(2 * T2 - 1, d)
                pe[:, 0::2] = torch.sin(pos * div); pe[:, 1::2] = torch.cos(pos * div)
                t = pe.to(self.device).to(BF16).contiguous()
                self._pos[key] = (t,)
            elif ptype == "rotary":
                hd = d // H
                inv = 1.0 / (c.get("rotary_embedding_base", 10000) ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
                fr = torch.einsum("i,j->ij", torch.arange(T2).float(), inv)
                emb = torch.cat((fr, fr), dim=-1)
                self._pos[key] = (emb.cos().contiguous().to(self.device), emb.sin().contiguous().to(self.device), (-emb.sin()).contiguous().to(self.device))
            else:
                self._pos[key] = None
        return self._pos[key]

    # ------------------------------------------------------------------ forward + backward
    def forward_backward(self, *args, **kwargs):
        """see `_forward_backward`; the whole step runs with torch's current stream resolved once (ops.pinned_stream)"""
        with ops.pinned_stream():
            return self._forward_backward(*args, **kwargs)

    def _forward_backward(self, feats, feat_lengths, labels, *, loss_scale=1.0, extra_hidden_grad=None, backward=True, keep_hidden=False,
                          train_mode=False, step_index=None, noise_mask=None, skip_layers=None):
        """feats (B,T,F) f32 device; feat_lengths (B) int32 or None; labels (B,U) int64 (<0 = padding).
        Returns dict(loss, logits (B,T2,V+1) f32, outer_len, last_hidden).  Gradients of loss_scale/world * loss accumulate into the store.
        `extra_hidden_grad`: optional callable(last_hidden f32 (M,d), outer_len (B) int32) -> f32 (M,d) gradient to add at the encoder output
        (the attention decoder of the joint model hooks in here).
        `noise_mask` = (time mask (B*T2) uint8, std): BEST-RQ masking — those frames of the encoder input are replaced by N(0, std) noise
        (bestrq.py:84-97), counter-based like the dropout masks (global stream site 3).
        `skip_layers`: LayerDrop decisions for this step (iterable of layer indices that are skipped: the layer is the identity and its
        parameters get no gradient).  None = draw them as the reference does — one uniform number per layer and step, layer skipped when it
        is below `config.layerdrop` (every rank draws its own, from its dropout seed) — in training mode; nothing is skipped in eval."""
        c, st = self.cfg, self.store
        if getattr(self, "_lnred", None) is not None:
            self._lnred.items, self._lnred.keep = [], []              # a step that raised part-way leaves deferred LayerNorm reductions behind: they must not land in this step's gradients
        if getattr(self, "_tnb", None) is None:
            self._tnb = T.TnBatch()
        self._tnb.items = []
        # first backward after zero_grad: every layer matrix is the target of exactly one weight-gradient problem, so the grouped launches write dW / db instead of adding
        # into the zeros (their epilogue otherwise ends with a dependent read of the output tile: 516 MB per step at the base size); any later backward accumulates
        self._tnb.overwrite = bool(backward and self.dw_overwrite and getattr(self.store, "fresh", False))
        if backward:
            self.store.fresh = False
        self._ranges_waiting = []
        P, G, W, WT = st.p, st.g, st.bf, st.bfT
        GL = lambda n, sl=None: None if n in self.frozen else (st.g(n) if sl is None else st.g(n)[sl])      # gradient of a linear's weight / bias, None when frozen
        dev = self.device
        feats = feats.to(F32).contiguous()
        B, Tn, Fq = feats.shape
        d, H, I, L, V1 = c["hidden_size"], c["num_attention_heads"], c["intermediate_size"], c["num_hidden_layers"], c["vocab_size"] + 1
        C1, C2 = c["conv_dim"]
        K, s_, pad = c["conv_kernel"][0], c["conv_stride"][0], c["conv_padding"][0]
        T1, T2 = self.out_frames(Tn)
        F1 = (Fq + 2 * pad - K) // s_ + 1
        F2 = (F1 + 2 * pad - K) // s_ + 1
        M, hd = B * T2, d // H
        ptype = c.get("position_embeddings_type", "relative")
        macaron = c.get("use_macaron_ff", True)
        eps_e = float(c.get("layer_norm_eps", 1e-5))
        kc = c.get("csgu_kernel_size", 31)
        scale = 1.0 / math.sqrt(hd)
        if feat_lengths is not None:
            feat_lengths = feat_lengths.to(device=dev, dtype=torch.int32).contiguous()
            inner, outer = self._lengths(feat_lengths, T2)
        else:
            inner = None
            outer = torch.full((B,), self._outer_len(Tn), dtype=torch.int32, device=dev)
        e32 = lambda *sh: torch.empty(sh, device=dev, dtype=F32)
        e16 = lambda *sh: torch.empty(sh, device=dev, dtype=BF16)
        LN = ops.layernorm_chain
        pd, seed = (self.pdrop if (backward or train_mode) else dict.fromkeys(self.pdrop, 0.0)), self.seed
        self._step_idx = self.train_steps_seen if step_index is None else int(step_index)
        if backward or train_mode:
            self.train_steps_seen += 1

        # ---------------- front end
        causal = self.causal
        padl = 2 * pad if causal else pad                  # CausalConv2d: all of the padding on the top / left (streaming_modules.py:31-55)
        # e_branchformer.py:153-160 hands (K-1)//2 to CausalConv1d's dilation slot: the causal CSGU conv is dilated by 15 with a left pad of (K-1)*15
        cs_dil = (kc - 1) // 2 if causal else 1
        cs_pad = (kc - 1) * cs_dil if causal else (kc - 1) // 2
        cm = self.ctx_mode
        if cm == 0:
            act1 = ops.conv2d_first_gelu(feats, P("conv1_w"), P("conv1_b"), stride=s_, pad=pad, causal=causal)
            pre2 = ops.conv2d_cl(act1, W("conv2_w"), P("conv2_b"), K=K, stride=s_, pad=pad, causal=causal, act="none").view(B * T2 * F2, C2)
            act2 = T.act_fwd(pre2).view(M, F2 * C2)
        else:
            # GatedConv2d / GatedConv2dShared (extractors.py:23-54): GELU(conv(x) * sigmoid(gate(x))), the shared gate being one row per four conv rows from a
            # (4K, K) / stride (4s, s) / padding (4p, p) conv.  Training keeps the raw conv / gate outputs (bf16) for the backward: un-fused passes.
            share = GATE_SHARE if cm == 2 else 1
            gkh, gkw, gst, gsf, gpt, gpf = gate_geometry(K, s_, pad, cm)
            gK, gS, gP, cK, cS, cP = (gkh, gkw), (gst, gsf), (gpt, gpf), (K, K), (s_, s_), (pad, pad)
            if cm == 2:
                for Tc, Tg in ((T1, (Tn + 2 * gpt - gkh) // gst + 1), (T2, (T1 + 2 * gpt - gkh) // gst + 1)):
                    if Tc % share or Tc // share != Tg:
                        raise RuntimeError(f"gated_shared front end: conv time axis {Tc} vs gate {Tg}: the reference's view(B, C, -1, {share}, F) * gate needs "
                                           f"{Tc} % {share} == 0 and {Tc} // {share} == {Tg} (extractors.py:49-54)")
            z1 = ops.conv2d_first_geo(feats, P("conv1_w"), P("conv1_b"), K=cK, stride=cS, pad=cP, act="none")
            g1 = ops.conv2d_first_geo(feats, P("gate1_w"), P("gate1_b"), K=gK, stride=gS, pad=gP, act="none")
            act1 = ops.gated_act(z1, g1, B, T1, F1, C1, share).view(B, T1, F1, C1)
            if cm == 1:
                pre2 = ops.conv2d_cl(act1, W("conv2_w"), P("conv2_b"), K=K, stride=s_, pad=pad, act="none").view(B * T2 * F2, 2 * C2)
                z2, g2 = pre2[:, :C2], pre2[:, C2:]
            else:
                z2 = ops.conv2d_cl(act1, W("conv2_w"), P("conv2_b"), K=K, stride=s_, pad=pad, act="none").view(B * T2 * F2, C2)
                g2 = ops.conv2d_cl_geo(act1, W("gate2_w"), P("gate2_b"), K=gK, stride=gS, pad=gP, act="none").view(-1, C2)
            act2 = ops.gated_act(z2, g2, B, T2, F2, C2, share).view(M, F2 * C2)
        feo = ops.gemm(act2, W("feout_w"), P("feout_b"), out_dtype=F32)
        a_fp = e16(M, d)
        LN(feo, lna=(P("fp_ln_g"), P("fp_ln_b")), eps2=eps_e, outa=a_fp)
        x = ops.gemm(a_fp, W("fp_w"), P("fp_b"), out_dtype=F32)
        if pd["fp"] > 0:
            T.dropout_(x, pd["fp"], seed, self._sid(L, 0))
        tmask = fmask = None
        if noise_mask is not None:
            from . import _lib
            nm, nstd = noise_mask
            _lib.check(_lib.lib().mi_mask_noise_f32(x.data_ptr(), x.stride(0), nm.data_ptr(), M, d, float(nstd), seed, self._sid(L, 3),
                                                    torch.cuda.current_stream().cuda_stream), "mi_mask_noise_f32")
        elif self.specaug and (backward or train_mode):
            tmask, fmask = self._spec_masks(B, T2, d, inner)
            T.spec_mask_apply_(x, tmask, P("masked_spec_embed"), fmask, T2)
        if inner is not None:
            T.mask_rows_(x, inner, T2)
        if pd["hidden"] > 0:
            T.dropout_(x, pd["hidden"], seed, self._sid(L, 1))
        pos = self._pos_table(T2)
        saved = []
        if skip_layers is None:
            skip_layers = [l for l in range(L) if _u01(seed, self._sid(l, 15)) < self.layerdrop] if (self.layerdrop > 0 and (backward or train_mode)) else []
        skip = set(int(l) for l in skip_layers)
        self.last_skipped = sorted(skip)
        # ---------------- layers
        def layer_fwd(x, l):
            p, sl = f"l{l}.", l + int(l >= L)          # sl: dropout-stream layer id (L itself names the global sites; the additional layer takes L + 1)
            S = {"x_in": x}
            if macaron:
                x, S["ff1"] = self._ffn_fwd(x, p + "ff1", LN, e16, pd, sl, (0, 1))
            S["x1"] = x
            a1, a2 = e16(M, d), e16(M, d)
            LN(x, lna=(P(p + "att_ln_g"), P(p + "att_ln_b")), outa=a1, lnb=(P(p + "mlp_ln_g"), P(p + "mlp_ln_b")), outb=a2)
            cat = e16(M, 2 * d)
            # global branch
            qkv = e16(M, 3 * d)
            if ptype == "rotary":
                a1r = ops.rotary(a1, pos[0].reshape(-1), pos[1].reshape(-1), T2, H)
                ops.gemm(a1r, W(p + "att_wqkv")[:2 * d], P(p + "att_bqkv")[:2 * d], out=qkv[:, :2 * d])
                ops.gemm(a1, W(p + "att_wqkv")[2 * d:], P(p + "att_bqkv")[2 * d:], out=qkv[:, 2 * d:])
                S["a1r"] = a1r
            else:
                ops.gemm(a1, W(p + "att_wqkv"), P(p + "att_bqkv"), out=qkv)
            posp = None
            if ptype == "relative":
                posp = ops.gemm(pos[0], W(p + "att_wpos"))
            ctx = self._attention_fwd(qkv, posp, P(p + "att_u") if posp is not None else None, P(p + "att_v") if posp is not None else None,
                                      inner, B, T2, H, S, (pd["att"], seed, self._sid(sl, 2)) if pd["att"] > 0 else None)
            if pd["att"] > 0:       # linear_out + self_attn_dropout (e_branchformer.py:288): the dropout rides the GEMM's epilogue
                T.gemm_dropout(ctx, W(p + "att_wo"), P(p + "att_bo"), pd["att"], seed, self._sid(sl, 3), out=cat[:, :d])
            else:
                ops.gemm(ctx, W(p + "att_wo"), P(p + "att_bo"), out=cat[:, :d])
            # local branch (cgMLP)
            hp, h = T.gemm_act_fwd(a2, W(p + "mlp_w1"), P(p + "mlp_b1"))
            stats = ops.row_stats(h[:, I // 2:])
            cv = lin = None
            if self.csgu_split:                              # e_branchformer.py:196-201: conv -> [Linear] -> act -> gate
                cv = lin = ops.csgu_conv(h, stats, P(p + "csgu_ln_g"), P(p + "csgu_ln_b"), P(p + "csgu_w"), P(p + "csgu_b"), B, T2, pad_left=cs_pad, dilation=cs_dil)
                if self.csgu_lin:
                    lin = ops.gemm(cv, W(p + "csgu_lin_w"), P(p + "csgu_lin_b"))
                sg = ops.gate_act_mul(h[:, :I // 2], lin, self.csgu_act)
            else:
                sg = ops.csgu(h, P(p + "csgu_ln_g"), P(p + "csgu_ln_b"), P(p + "csgu_w"), P(p + "csgu_b"), B, T2, pad_left=cs_pad, dilation=cs_dil, stats=stats)
            if pd["csgu"] > 0:
                T.dropout_(sg, pd["csgu"], seed, self._sid(sl, 4))
            ops.gemm(sg, W(p + "mlp_w2"), P(p + "mlp_b2"), out=cat[:, d:])
            # merge
            m2 = ops.dwconv_residual(cat, P(p + "mrg_dw_w"), P(p + "mrg_dw_b"), B, T2)
            if pd["att"] > 0:       # the layer's `final_dropout` module takes config.attention_dropout (e_branchformer.py:229,246)
                x2 = T.gemm_dropout(m2, W(p + "mrg_w"), P(p + "mrg_b"), pd["att"], seed, self._sid(sl, 5), resid=x, alpha=1.0)
            else:
                x2 = ops.gemm(m2, W(p + "mrg_w"), P(p + "mrg_b"), out_dtype=F32, resid=x, alpha=1.0)
            S.update(a1=a1, a2=a2, qkv=qkv, posp=posp, ctx=ctx, hp=hp, h=h, stats=stats, sg=sg, cat=cat, m2=m2, x2=x2, cv=cv, lin=lin)
            x = x2
            if macaron:
                x, S["ff2"] = self._ffn_fwd(x, p + "ff2", LN, e16, pd, sl, (6, 7))
            S["x3"] = x
            xo = e32(M, d)
            LN(x, ln1=(P(p + "fin_ln_g"), P(p + "fin_ln_b")), store_y=xo)
            x = xo
            return x, S

        hs = [] if self.mix else None       # HF's `hidden_states` tuple: the INPUT of every layer, then the final LayerNorm's output (tf:679-680,714-715)
        for l in range(L):
            if hs is not None:
                hs.append(x)
            if l in skip:                   # LayerDrop: identity, nothing saved, no gradient
                saved.append(None)
                continue
            x, S = layer_fwd(x, l)
            saved.append(S)
        # ---------------- head + CTC
        hid = e16(M, d)
        last_hidden = e32(M, d)
        LN(x, lna=(P("enc_ln_g"), P("enc_ln_b")), eps2=eps_e, outa=hid, outa32=last_hidden)
        sw = mixed = S_extra = None
        if self.mix or self.extra:
            if extra_hidden_grad is not None:
                raise NotImplementedError("layer mixing / additional layer: CTC fine-tuning head only (no attention decoder on top)")
            top = last_hidden
            if self.mix:            # bestrq.py:239-245 — the weights never leave the device
                hs.append(last_hidden)
                sw = T.softmax_vec(P("mix_w"))
                mixed = e32(M, d)
                for i, h in enumerate(hs):
                    T.axpy_dev_(mixed, h, sw[i:i + 1], overwrite=(i == 0))
                top = mixed
            if self.extra:          # bestrq.py:247-274: padded frames zeroed, then one more layer with the encoder's mask and position table; no LayerNorm after it
                xin = top if top is mixed else top.clone()
                if inner is not None:
                    T.mask_rows_(xin, inner, T2)
                top, S_extra = layer_fwd(xin, L)
            hid = T.add_cast(top)
        loss = logits = lse = nll = None
        red = c.get("ctc_loss_reduction", "mean")
        ldl = T.pad64(V1)
        if self.head:
            if pd["final"] > 0:
                T.dropout_(hid, pd["final"], seed, self._sid(L, 2))
            lbuf = e32(B, T2, ldl)
            if os.environ.get("HFASR_TRAIN_HEAD_LSE", "1") != "0":
                lse = ops.gemm_lse(hid, W("head_w"), P("head_b"), lbuf.view(M, ldl))       # logits and their row log-sum-exp from one pass (the GEMM's epilogue)
            else:                                                                          # (the two passes: A/B and trajectory comparisons)
                ops.gemm(hid, W("head_w"), P("head_b"), out=lbuf.view(M, ldl))
                lse = ops.row_lse(lbuf.view(M, ldl)[:, :V1])
            logits = lbuf[..., :V1]
            if labels is not None:
                if not (backward and self.ctc_from_bwd and red in ("mean", "sum")):       # with a backward pass the loss comes out of ITS alpha recursion (below): no forward loss kernel
                    loss, nll, _ = ops.ctc_loss(logits, labels, outer, reduction=red, zero_infinity=bool(c.get("ctc_zero_infinity", False)), lse=lse)
            elif backward:
                raise ValueError("forward_backward: the backward pass of the CTC head needs `labels`")
        out = dict(loss=loss, logits=logits, outer_len=outer, inner_len=inner,
                   last_hidden=last_hidden.view(B, T2, d) if keep_hidden or extra_hidden_grad or not self.head else None)
        if not backward:
            return out

        # =================================================================== backward
        def layer_bwd(dx, S, l):
            p, sl = f"l{l}.", l + int(l >= L)
            # final_layer_norm
            # Every LayerNorm backward whose dx is next turned into a bf16 GEMM operand (scaled, dropped) writes that operand itself (`cast=`): no pass of its own
            hdrop = lambda site: (pd["hidden"], seed, self._sid(sl, site)) if pd["hidden"] > 0 else None
            mdrop = (pd["att"], seed, self._sid(sl, 5)) if pd["att"] > 0 else None
            d3 = e32(M, d)
            if macaron:
                _, dyb = T.layernorm_bwd(S["x3"], P(p + "fin_ln_g"), dx, d3, accumulate=False, **self._lng(p + "fin_ln_g", p + "fin_ln_b"), cast=(0.5, hdrop(7)))
                dx = d3
                dyb = self._ffn_bwd(dx, S["x2"], S["ff2"], p + "ff2", pd, sl, (6, 7), dyb=dyb, cast_next=(1.0, mdrop))
            else:
                _, dyb = T.layernorm_bwd(S["x3"], P(p + "fin_ln_g"), dx, d3, accumulate=False, **self._lng(p + "fin_ln_g", p + "fin_ln_b"), cast=(1.0, mdrop))
                dx = d3
            # merge:  x2 = x1 + dropout(merge_proj(m2));  dyb = dropout(dx) as bf16
            dm2 = T.linear_bwd(dyb, S["m2"], WT(p + "mrg_w"), dw=GL(p + "mrg_w"), db=GL(p + "mrg_b"), defer=self._tnb)
            dcat = e16(M, 2 * d)
            dwred = self._dwred()                       # the depthwise convs' cross-utterance tap-gradient sums ride the deferred LayerNorm reductions' launches
            T.dwconv_residual_bwd(S["cat"], P(p + "mrg_dw_w"), dm2, dcat, G(p + "mrg_dw_w"), G(p + "mrg_dw_b"), B, T2, defer=dwred)
            # local branch
            dsg = T.linear_bwd(dcat[:, d:], S["sg"], WT(p + "mlp_w2"), dw=GL(p + "mlp_w2"), db=GL(p + "mlp_b2"), defer=self._tnb)
            if pd["csgu"] > 0:
                T.dropout_(dsg, pd["csgu"], seed, self._sid(sl, 4))
            dh = e16(M, I)
            dgn = e16(M, I // 2)
            if self.csgu_split:
                dlin = T.gate_act_mul_bwd(S["h"][:, :I // 2], S["lin"], dsg, dh[:, :I // 2], self.csgu_act)
                dcv = T.linear_bwd(dlin, S["cv"], WT(p + "csgu_lin_w"), dw=GL(p + "csgu_lin_w"), db=GL(p + "csgu_lin_b"), defer=self._tnb) if self.csgu_lin else dlin
                T.csgu_bwd(S["h"], S["stats"], P(p + "csgu_ln_g"), P(p + "csgu_ln_b"), P(p + "csgu_w"), P(p + "csgu_b"), dcv, None, dgn,
                           G(p + "csgu_w"), G(p + "csgu_b"), B, T2, pad_left=cs_pad, dilation=cs_dil, defer=dwred)
            else:
                T.csgu_bwd(S["h"], S["stats"], P(p + "csgu_ln_g"), P(p + "csgu_ln_b"), P(p + "csgu_w"), P(p + "csgu_b"), dsg, dh[:, :I // 2], dgn,
                           G(p + "csgu_w"), G(p + "csgu_b"), B, T2, pad_left=cs_pad, dilation=cs_dil, defer=dwred)
            T.layernorm_bwd(S["h"][:, I // 2:], P(p + "csgu_ln_g"), dgn, dh[:, I // 2:], accumulate=False, **self._lng(p + "csgu_ln_g", p + "csgu_ln_b"))
            dhp = T.act_bwd(dh, S["hp"])       # (folding this pass into the two kernels above was built and measured: each slows by what its share of this one costs — DESIGN §7)
            da2 = T.linear_bwd(dhp, S["a2"], WT(p + "mlp_w1"), dw=GL(p + "mlp_w1"), db=GL(p + "mlp_b1"), defer=self._tnb)
            # the two branch norms read the same x1: one pass for both when their affine pairs train (the gradient w.r.t. x1 is linear in dy * gamma)
            lng_m, lng_a = self._lng(p + "mlp_ln_g", p + "mlp_ln_b"), self._lng(p + "att_ln_g", p + "att_ln_b")
            dual = self.dual_ln and ptype != "rotary" and d <= 512 and lng_m["dgamma"] is not None and lng_a["dgamma"] is not None
            if not dual:
                T.layernorm_bwd(S["x1"], P(p + "mlp_ln_g"), da2, dx, accumulate=True, **lng_m)
            # global branch
            if pd["att"] > 0:
                T.dropout_(dcat[:, :d], pd["att"], seed, self._sid(sl, 3))
            dctx = T.linear_bwd(dcat[:, :d], S["ctx"], WT(p + "att_wo"), dw=GL(p + "att_wo"), db=GL(p + "att_bo"), defer=self._tnb)
            dqkv = self._attention_bwd(dctx, S, p, pos, inner, B, T2, H, (pd["att"], seed, self._sid(sl, 2)) if pd["att"] > 0 else None)
            if ptype == "rotary":
                wt = WT(p + "att_wqkv")
                da1r = T.linear_bwd(dqkv[:, :2 * d], S["a1r"], wt[:, :2 * d], dw=GL(p + "att_wqkv", slice(0, 2 * d)), db=GL(p + "att_bqkv", slice(0, 2 * d)), defer=self._tnb)
                da1 = T.linear_bwd(dqkv[:, 2 * d:], S["a1"], wt[:, 2 * d:3 * d], dw=GL(p + "att_wqkv", slice(2 * d, None)), db=GL(p + "att_bqkv", slice(2 * d, None)), dx_dtype=F32, defer=self._tnb)
                rot = ops.rotary(da1r, pos[0].reshape(-1), pos[2].reshape(-1), T2, H)             # R^T = rotation by -theta
                T.layernorm_bwd(S["x1"], P(p + "att_ln_g"), da1, dx, accumulate=True, **self._lng(p + "att_ln_g", p + "att_ln_b"))
                last = dict(x=S["x1"], g=P(p + "att_ln_g"), dy=rot)
            else:
                da1 = T.linear_bwd(dqkv, S["a1"], WT(p + "att_wqkv")[:, :3 * d], dw=GL(p + "att_wqkv"), db=GL(p + "att_bqkv"), defer=self._tnb)
                last = dict(x=S["x1"], g=P(p + "att_ln_g"), dy=da1)
                if dual:
                    r = T.layernorm_bwd_dual(S["x1"], P(p + "att_ln_g"), da1, P(p + "mlp_ln_g"), da2, dx, accumulate=True, dgamma=lng_a["dgamma"], dbeta=lng_a["dbeta"],
                                             dgamma2=lng_m["dgamma"], dbeta2=lng_m["dbeta"], defer=lng_a["defer"], cast=(0.5, hdrop(1)) if macaron else None)
                    if macaron:
                        self._ffn_bwd(dx, S["x_in"], S["ff1"], p + "ff1", pd, sl, (0, 1), dyb=r[1])
                    return dx
            if macaron:             # the layer's last accumulation into dx also leaves the first FFN's bf16 operand
                _, dyb = T.layernorm_bwd(last["x"], last["g"], last["dy"], dx, accumulate=True, **self._lng(p + "att_ln_g", p + "att_ln_b"), cast=(0.5, hdrop(1)))
                self._ffn_bwd(dx, S["x_in"], S["ff1"], p + "ff1", pd, sl, (0, 1), dyb=dyb)
            else:
                T.layernorm_bwd(last["x"], last["g"], last["dy"], dx, accumulate=True, **self._lng(p + "att_ln_g", p + "att_ln_b"))
            return dx

        gs = float(loss_scale) / self.sync.world
        dhid = None
        if self.head:
            dlog = None
            if nll is None:         # loss, per-utterance nll and the gradient from one pair of recursions; a target too long for that kernel takes the two calls
                zi = bool(c.get("ctc_zero_infinity", False))
                r = T.ctc_loss_bwd_nll(logits, lse, labels, outer, reduction=red, zero_infinity=zi, gscale=gs, ldo=ldl)
                if r is not None:
                    dlog, loss, nll = r
                else:
                    loss, nll, _ = ops.ctc_loss(logits, labels, outer, reduction=red, zero_infinity=zi, lse=lse)
                out["loss"] = loss
            if dlog is None:
                dlog = T.ctc_loss_bwd(logits, lse, labels, outer, nll, reduction=red, gscale=gs, ldo=ldl)         # (M, ldl) bf16
            dhid = T.gemm(dlog, WT("head_w"), out_dtype=F32 if (self.mix or self.extra) else BF16)           # (M, d)
            if pd["final"] > 0:
                T.dropout_(dhid, pd["final"], seed, self._sid(L, 2))
            T.gemm_tn_(G("head_w"), dlog, hid, n_store=V1, db=G("head_b"))
        dx = e32(M, d)
        dmix = None
        if self.mix or self.extra:
            self._range_done(*st.range_of(["head_w", "head_b"]))
            dtop = dhid                                               # f32: gradient at the head's input
            if self.extra:
                dtop = layer_bwd(dtop, S_extra, L)
                if inner is not None:
                    T.mask_rows_(dtop, inner, T2)
                self._range_done(*st.range_of(self._layer_names[L]))
            if self.mix:            # d hidden_l = s_l * d mixed;  d per_layer_weights = s * (g - <s, g>),  g_l = <d mixed, hidden_l>
                dmix = dtop
                gdot = torch.zeros(L + 1, device=dev, dtype=F32)
                for i, h in enumerate(hs):
                    T.dot_(gdot[i:i + 1], dmix, h.reshape(M, d))
                T.softmax_vec_bwd_(G("mix_w"), sw, gdot)
                dtop = T.axpy_dev_(e32(M, d), dmix, sw[L:L + 1], overwrite=True)
            T.layernorm_bwd(x, P("enc_ln_g"), dtop, dx, accumulate=False, **self._lng("enc_ln_g", "enc_ln_b"), eps=eps_e)
        elif dhid is not None:
            T.layernorm_bwd(x, P("enc_ln_g"), dhid, dx, accumulate=False, **self._lng("enc_ln_g", "enc_ln_b"), eps=eps_e)
        if extra_hidden_grad is not None:
            dh32 = extra_hidden_grad(last_hidden, outer)
            if dh32 is not None:
                T.layernorm_bwd(x, P("enc_ln_g"), dh32, dx, accumulate=dhid is not None, **self._lng("enc_ln_g", "enc_ln_b"), eps=eps_e)
            elif dhid is None:
                dx.zero_()
        self._range_done(*st.range_of(self._encln_names if (self.mix or self.extra) else self._head_names))
        for l in range(L - 1, -1, -1):
            S = saved[l]
            if S is not None:               # (a dropped layer: dx passes through, its gradient range stays zero — still reduced: other ranks may have run it)
                dx = layer_bwd(dx, S, l)
            if dmix is not None:            # layer mixing: hidden_states[l] is this layer's input
                T.axpy_dev_(dx, dmix, sw[l:l + 1])
            self._range_done(*st.range_of(self._layer_names[l]))
        # ---------------- front end
        if pd["hidden"] > 0:
            T.dropout_(dx, pd["hidden"], seed, self._sid(L, 1))
        if inner is not None:
            T.mask_rows_(dx, inner, T2)
        if noise_mask is not None:
            T.spec_mask_bwd_(dx, noise_mask[0], None, None, T2)          # replaced frames pass no gradient upstream
        elif tmask is not None or fmask is not None:
            T.spec_mask_bwd_(dx, tmask, G("masked_spec_embed"), fmask, T2)
        dyb = T.dropout_(dx, pd["fp"], seed, self._sid(L, 0), out=e16(M, d)) if pd["fp"] > 0 else T.add_cast(dx)
        da = T.linear_bwd(dyb, a_fp, WT("fp_w"), dw=GL("fp_w"), db=GL("fp_b"))
        dfeo = e32(M, d)
        T.layernorm_bwd(feo, P("fp_ln_g"), da, dfeo, accumulate=False, **self._lng("fp_ln_g", "fp_ln_b"), eps=eps_e)
        dact2 = T.linear_bwd(T.add_cast(dfeo), act2, WT("feout_w"), dw=GL("feout_w"), db=GL("feout_b"))      # (M, F2*C2)
        if cm == 0:
            dpre2 = T.act_bwd(dact2.view(B * T2 * F2, C2), pre2)
            T.conv2d_wgrad_(G("conv2_w"), dpre2, act1, K, s_, padl, T2, F2, db=G("conv2_b"))      # the im2col operand is gathered inside the GEMM, never written
            if K == 3 and s_ == 2 and T.conv2d_s2k3_dgrad_supported(B, T1, F1, C1, T2, F2, padl):
                # conv2's input gradient as four stride-1 convolutions (one per parity of the position) into phase buffers that conv1's backward reads directly
                T.conv2d_first_bwd_phases(feats, P("conv1_w"), P("conv1_b"), dpre2, WT("conv2_w"), G("conv1_w"), G("conv1_b"), K, s_, padl, T1, F1, padl, T2, F2)
            else:
                dcol = ops.gemm(dpre2, WT("conv2_w")[:, :C2])
                T.conv2d_first_bwd(feats, P("conv1_w"), P("conv1_b"), dcol, G("conv1_w"), G("conv1_b"), K, s_, padl, T1, F1, K, s_, padl, T2, F2)
        else:
            # y = z * sigmoid(g), out = GELU(y): dz = dout GELU'(y) sigmoid(g), dg = sum over the rows sharing g of dout GELU'(y) z sigmoid(g)(1 - sigmoid(g));
            # then conv and gate are two plain convs (ONE for "gated": conv and gate rows stacked) — dW = dY^T col, dX = col2im(dY W)
            dact2 = dact2.view(B * T2 * F2, C2)
            col = T.im2col(act1, K, s_, pad, T2, F2)
            if cm == 1:
                dzg = e16(B * T2 * F2, 2 * C2)
                T.gated_act_bwd(dact2, z2, g2, B, T2, F2, C2, 1, dz=dzg[:, :C2], dg=dzg[:, C2:])
                T.gemm_tn_(G("conv2_w"), dzg, col, db=G("conv2_b"))
                dcol = ops.gemm(dzg, WT("conv2_w")[:, :2 * C2])
                del col
                dact1 = T.col2im(dcol, (B, T1, F1, C1), cK, cS, cP, T2, F2)
            else:
                dz2, dg2 = T.gated_act_bwd(dact2, z2, g2, B, T2, F2, C2, share)
                T.gemm_tn_(G("conv2_w"), dz2, col, db=G("conv2_b"))
                dcol = ops.gemm(dz2, WT("conv2_w")[:, :C2])
                del col
                dact1 = T.col2im(dcol, (B, T1, F1, C1), cK, cS, cP, T2, F2)
                colg = T.im2col_geo(act1, gK, gS, gP, T2 // share, F2)
                T.gemm_tn_(G("gate2_w"), dg2, colg, db=G("gate2_b"))
                dcol = ops.gemm(dg2, WT("gate2_w")[:, :C2])
                del colg
                T.col2im(dcol, (B, T1, F1, C1), gK, gS, gP, T2 // share, F2, out=dact1)
            del dcol
            dz1, dg1 = T.gated_act_bwd(dact1.view(-1, C1), z1.view(-1, C1), g1.view(-1, C1), B, T1, F1, C1, share)
            T.conv2d_first_wgrad(feats, dz1.view(B, T1, F1, C1), G("conv1_w"), G("conv1_b"), cK, cS, cP)
            T.conv2d_first_wgrad(feats, dg1.view(B, T1 // share, F1, C1), G("gate1_w"), G("gate1_b"), gK, gS, gP)
        self._range_done(*st.range_of(self._front_names), final=True)
        return out

    # ------------------------------------------------------------------ pieces
    def _sid(self, layer: int, site: int) -> int:
        """dropout stream id of (this step, layer, site); layer = num_hidden_layers for the global sites (0 feat-proj, 1 encoder input, 2 head)"""
        return ((self._step_idx * 64 + layer) * 16 + site) & 0xFFFFFFFF

    def _spec_masks(self, B, T2, d, inner):
        """SpecAugment masks of tf `_mask_hidden_states` (:1086-1130), drawn on the host by transformers' own helper with numpy's global RNG
        (so a run seeded like the reference's draws the reference's masks).  -> (time mask (B*T2) u8 | None, feature mask (B, d) u8 | None)"""
        import numpy as np
        from transformers.models.wav2vec2.modeling_wav2vec2 import _compute_mask_indices
        c = self.cfg
        tm = fm = None
        if float(c.get("mask_time_prob", 0.0) or 0.0) > 0:
            am = None
            if inner is not None:
                am = torch.arange(T2)[None, :] < inner.cpu()[:, None].long()
            m = _compute_mask_indices((B, T2), mask_prob=c["mask_time_prob"], mask_length=c.get("mask_time_length", 10), attention_mask=am,
                                      min_masks=c.get("mask_time_min_masks", 2))
            tm = torch.from_numpy(np.ascontiguousarray(m).astype(np.uint8)).reshape(-1).to(self.device)
        if float(c.get("mask_feature_prob", 0.0) or 0.0) > 0:
            m = _compute_mask_indices((B, d), mask_prob=c["mask_feature_prob"], mask_length=c.get("mask_feature_length", 10),
                                      min_masks=c.get("mask_feature_min_masks", 0))
            fm = torch.from_numpy(np.ascontiguousarray(m).astype(np.uint8)).to(self.device)
        return tm, fm

    def _outer_len(self, n):
        k, s = self.cfg["conv_kernel"][0], self.cfg["conv_stride"][0]
        for _ in range(2):
            n = (n - k) // s + 1
        return n

    def _lengths(self, feat_lengths, T2):
        c = self.cfg
        k, s, p = c["conv_kernel"][0], c["conv_stride"][0], c["conv_padding"][0]
        if feat_lengths.is_cuda and feat_lengths.dtype == torch.int32:                 # one launch instead of fifteen one-block torch kernels
            from . import _lib
            inner, outer = torch.empty_like(feat_lengths), torch.empty_like(feat_lengths)
            _lib.check(_lib.lib().mi_subsampled_lengths_i32(feat_lengths.data_ptr(), feat_lengths.numel(), int(k), int(s), int(p), 2, int(T2), inner.data_ptr(), outer.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream), "mi_subsampled_lengths_i32")
            return inner, outer
        li, lo = feat_lengths.clone(), feat_lengths.clone()
        for _ in range(2):
            li = torch.div(li + 2 * p - k, s, rounding_mode="floor") + 1
            lo = torch.div(lo - k, s, rounding_mode="floor") + 1
        return torch.clamp(li, max=T2).to(torch.int32), lo.to(torch.int32)

    def _ffn_fwd(self, x, pre, LN, e16, pd, l, sites):
        """x + 0.5 * dropout(W2 dropout(gelu(W1 LN(x))))  (e_branchformer.py:271-273, 307-309; tf:350-357)"""
        P, W = self.store.p, self.store.bf
        M, d = x.shape
        a = e16(M, d)
        LN(x, lna=(P(pre + "_ln_g"), P(pre + "_ln_b")), outa=a)
        # intermediate_dense + GELU + activation dropout: one launch (the GEMM's training epilogue leaves the pre-activation for the backward and the activation)
        hp, h = T.gemm_act_fwd(a, W(pre + "_w1"), P(pre + "_b1"), drop=(pd["act"], self.seed, self._sid(l, sites[0])) if pd["act"] > 0 else None)
        if pd["hidden"] > 0:
            y = T.gemm_dropout(h, W(pre + "_w2"), P(pre + "_b2"), pd["hidden"], self.seed, self._sid(l, sites[1]), resid=x, alpha=0.5)
        else:
            y = ops.gemm(h, W(pre + "_w2"), P(pre + "_b2"), out_dtype=F32, resid=x, alpha=0.5)
        return y, dict(a=a, hp=hp, h=h)

    def _ffn_bwd(self, dx, x_in, S, pre, pd, l, sites, dyb=None, cast_next=None):
        """dx (f32, in place): gradient w.r.t. the block output -> gradient w.r.t. its input (residual + LN path).
        dyb: bf16(0.5 * dropout(dx)) when the producer of dx already wrote it; cast_next = (alpha, drop): -> the same operand of the NEXT linear backward,
        written by this block's LayerNorm backward."""
        P, G, WT = self.store.p, self.store.g, self.store.bfT
        GL = lambda n: None if n in self.frozen else self.store.g(n)
        if dyb is None:
            if pd["hidden"] > 0:
                dyb = T.dropout_(dx, pd["hidden"], self.seed, self._sid(l, sites[1]), out=torch.empty(dx.shape, device=dx.device, dtype=BF16), alpha=0.5)
            else:
                dyb = T.add_cast(dx, alpha=0.5)
        # dh = dy W2 with the activation (+ dropout) backward in the GEMM's epilogue; the weight / bias gradients of W2 go to the layer's grouped launch
        dhp = T.gemm_act_bwd(dyb, WT(pre + "_w2")[:, :dyb.shape[1]], S["hp"], drop=(pd["act"], self.seed, self._sid(l, sites[0])) if pd["act"] > 0 else None)
        T.linear_bwd(dyb, S["h"], WT(pre + "_w2"), dw=GL(pre + "_w2"), db=GL(pre + "_b2"), need_dx=False, defer=self._tnb)
        da = T.linear_bwd(dhp, S["a"], WT(pre + "_w1"), dw=GL(pre + "_w1"), db=GL(pre + "_b1"), defer=self._tnb)
        if cast_next is not None:
            return T.layernorm_bwd(x_in, P(pre + "_ln_g"), da, dx, accumulate=True, **self._lng(pre + "_ln_g", pre + "_ln_b"), cast=cast_next)[1]
        T.layernorm_bwd(x_in, P(pre + "_ln_g"), da, dx, accumulate=True, **self._lng(pre + "_ln_g", pre + "_ln_b"))
        return None

    def _attention_fwd(self, qkv, posp, u, v, lengths, B, Tt, H, S, drop=None):
        d = qkv.shape[1] // 3
        hd = d // H
        if hd in (64, 128):
            # the fused kernel (probability dropout, e_branchformer.py:132, included); it leaves the rows' log-sum-exp for the backward's recomputation (ops_train.attn_bwd_probs)
            S["lse"] = torch.empty((B, H, Tt), device=qkv.device, dtype=F32)
            return ops.attention_qkv(qkv, B, Tt, H, pos=posp, bias_u=u, bias_v=v, lengths=lengths, causal=self.causal, lse=S["lse"], drop=drop)
        # small heads (test configs): probabilities through the generic pieces,
        # kept for the backward pass; the dropped copy feeds the PV product
        prob = self._probs(qkv, posp, u, v, lengths, B, Tt, H, S, drop)
        if drop is not None:
            S["prob_pre"], prob = prob
        ctx = torch.empty((B * Tt, d), device=qkv.device, dtype=BF16)
        vv = qkv[:, 2 * d:]
        Ts = prob.shape[-1]
        T.bgemm(prob, (B * Tt * Ts, Tt * Ts, Ts, 1), vv, (hd, Tt * 3 * d, 1, 3 * d), ctx, (hd, Tt * d, d), H, B, Tt, hd, Tt)
        S["prob"] = prob
        return ctx

    def _probs(self, qkv, posp, u, v, lengths, B, Tt, H, S, drop=None):
        d = qkv.shape[1] // 3
        hd = d // H
        dev = qkv.device
        q, k = qkv[:, :d], qkv[:, d:2 * d]
        if posp is not None:
            qu, qv = T.add_rowvec2(q, u, v)
            a_q, a_str = qu, (hd, Tt * d, d, 1)
        else:
            qu = qv = None
            a_q, a_str = q, (hd, Tt * 3 * d, 3 * d, 1)
        Ts = T.pad8(Tt)                                    # padded row strides: 16-B aligned rows for the 16-B staging loads
        ac = torch.empty((H, B, Tt, Ts), device=dev, dtype=F32)
        T.bgemm(a_q, a_str, k, (hd, Tt * 3 * d, 3 * d, 1), ac, (B * Tt * Ts, Tt * Ts, Ts), H, B, Tt, Tt, hd)
        bd = None
        if posp is not None:
            Pn = 2 * Tt - 1
            Ps = T.pad8(Pn)
            bd = torch.empty((H, B, Tt, Ps), device=dev, dtype=F32)
            T.bgemm(qv, (hd, Tt * d, d, 1), posp, (hd, 0, d, 1), bd, (B * Tt * Ps, Tt * Ps, Ps), H, B, Tt, Pn, hd)
        S["qu"], S["qv"] = qu, qv
        return T.attn_softmax_fwd(ac, bd, lengths, H, B, Tt, Tt, 1.0 / math.sqrt(hd), causal=self.causal, drop=drop)

    def _attention_bwd(self, dctx, S, p, pos, lengths, B, Tt, H, drop=None):
        """-> dqkv (M, 3d) bf16; accumulates pos_bias_u / pos_bias_v / linear_pos gradients."""
        G, P = self.store.g, self.store.p
        qkv, posp = S["qkv"], S["posp"]
        d = qkv.shape[1] // 3
        hd = d // H
        M = B * Tt
        dev = qkv.device
        scale = 1.0 / math.sqrt(hd)
        rel = posp is not None
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        dqkv = torch.empty((M, 3 * d), device=dev, dtype=BF16)
        Pn = 2 * Tt - 1
        if "lse" in S:
            # fused forward: ONE walk over the keys recomputes the scores and leaves P, dS and the un-shifted dBD (bf16) — no fp32 score-sized tensors — and
            # accumulates dQ = dS K + dBD P on the way (its two terms' column sums are the position-bias gradients)
            # (with positions the walk also leaves q + u and q + v, its own A operands, for the dK and d(positions) products below)
            prob, ds, dbd, su, sv, qu, qv = T.attn_bwd_probs(qkv, B, Tt, H, S["ctx"], dctx, S["lse"], dqkv[:, :d], pos=posp, bias_u=P(p + "att_u") if rel else None,
                                                             bias_v=P(p + "att_v") if rel else None, lengths=lengths, causal=self.causal, drop=drop, qb=self.walk_qb, sparse=self.sparse_attn_bwd)[:7] + ((None, None) if not self.walk_qb else ())
            fused = True
            if rel:
                if qu is None:
                    qu, qv = T.add_rowvec2(q, P(p + "att_u"), P(p + "att_v"))
                off, Kp = T.band_geometry(Tt)                 # dbd's columns are relative positions + off
        else:
            fused = False
            prob = S.get("prob")                           # what multiplied V in the forward (dropped copy under dropout)
            if prob is None:
                prob = self._probs(qkv, posp, P(p + "att_u") if rel else None, P(p + "att_v") if rel else None, lengths, B, Tt, H, S)
            prob_pre = S.get("prob_pre", prob)             # un-dropped probabilities: the softmax Jacobian
            qu, qv = S["qu"], S["qv"]
            # dP = dctx · V^T
            Ts0 = prob.shape[-1]
            dp = torch.empty((H, B, Tt, Ts0), device=dev, dtype=F32)
            T.bgemm(dctx, (hd, Tt * d, d, 1), v, (hd, Tt * 3 * d, 3 * d, 1), dp, (B * Tt * Ts0, Tt * Ts0, Ts0), H, B, Tt, Tt, hd)
            ds, dbd = T.attn_softmax_bwd(prob_pre, dp, H, B, Tt, Tt, scale, want_dbd=rel, drop=drop)
            off, Kp = 0, Pn
        Ts = prob.shape[-1]
        sTT = (B * Tt * Ts, Tt * Ts)
        # dV = P^T · dctx
        # (keys beyond an utterance's length have all-zero rows in P^T and dS^T: their tiles are stored as zeros without being read — `m_valid`; at BASELINE config 3,
        # clips of 1-20 s padded to 20 s, that is half of both products)
        T.bgemm(prob, (*sTT, 1, Ts), dctx, (hd, Tt * d, 1, d), dqkv[:, 2 * d:], (hd, Tt * 3 * d, 3 * d), H, B, Tt, hd, Tt, m_valid=lengths)
        # dK = dS^T · (q + u)
        aq, aq_str = (qu, (hd, Tt * d, 1, d)) if rel else (q, (hd, Tt * 3 * d, 1, 3 * d))
        T.bgemm(ds, (*sTT, 1, Ts), aq, aq_str, dqkv[:, d:2 * d], (hd, Tt * 3 * d, 3 * d), H, B, Tt, hd, Tt, m_valid=lengths)
        if not rel:
            if not fused:      # dQ = dS · K
                T.bgemm(ds, (*sTT, Ts, 1), k, (hd, Tt * 3 * d, 1, 3 * d), dqkv[:, :d], (hd, Tt * 3 * d, 3 * d), H, B, Tt, hd, Tt)
            return dqkv
        Ps = dbd.shape[-1]
        if fused:
            self._dwred().add_rows2(su, G(p + "att_u"), G(p + "att_v"))       # the per-wave rows' column sums leave with the deferred LayerNorm reductions' next launch
        else:
            dqu = torch.empty((M, d), device=dev, dtype=F32)
            dqv = torch.empty((M, d), device=dev, dtype=F32)
            T.bgemm(ds, (*sTT, Ts, 1), k, (hd, Tt * 3 * d, 1, 3 * d), dqu, (hd, Tt * d, d), H, B, Tt, hd, Tt)
            T.bgemm(dbd, (B * Tt * Ps, Tt * Ps, Ps, 1), posp, (hd, 0, 1, d), dqv, (hd, Tt * d, d), H, B, Tt, hd, Pn)
            T.add_cast(dqu, dqv, out=dqkv[:, :d])
            T.colsum_(G(p + "att_u"), dqu)
            T.colsum_(G(p + "att_v"), dqv)
        # d(posp) (P, d) = sum_b dBD^T · (q + v): K runs over the (b, t) rows of one head.  Partials per group of `cg` utterances (B / cg, P, d), then a column
        # sum over the groups: one long-K product per head would occupy 64 blocks only, per-utterance partials are 4x the fp32 traffic of groups of four
        # (the largest group that still leaves the 128 x 128 tile >= 256 blocks: with fewer the batched GEMM falls back to 64-wide tiles and reads dBD twice)
        cg = next((c for c in (4, 2) if B % c == 0 and H * (B // c) * -(-Kp // 128) >= 256), 1)
        dpp = torch.empty((B // cg, Kp * d), device=dev, dtype=F32)
        T.bgemm(dbd, (B * Tt * Ps, cg * Tt * Ps, 1, Ps), qv, (hd, cg * Tt * d, 1, d), dpp, (hd, Kp * d, d), H, B // cg, Kp, hd, cg * Tt,
                band=(Tt, Tt - 1 + off, cg))               # row i of dBD is non-zero in columns [T - 1 - i + off, 2T - 1 - i + off) only: half of the k tiles are skipped
        # linear_pos: posp = table · Wpos^T  ->  dWpos += dposp^T · table.  dposp = the groups' partials summed in order, straight to the bf16 operand (rows off .. off + Pn)
        dpb = T.colsum_cast(dpp[:, off * d:(off + Pn) * d]).view(Pn, d)
        T.gemm_tn_(G(p + "att_wpos"), dpb, pos[0], defer=self._tnb)
        return dqkv

    # ------------------------------------------------------------------ optimizer
    def optimizer_step(self, lr=None):
        """waits for the gradient all-reduces, clips by global norm, applies AdamW, refreshes the bf16 mirrors."""
        with ops.pinned_stream():
            return self._optimizer_step(lr)

    def _optimizer_step(self, lr=None):
        st, hp = self.store, self.hp
        self.sync.wait()
        st.zero_frozen_grads()
        sc = self._scal
        sc.zero_()
        T.sumsq_(sc[0:1], st.flat_g)
        T.clip_coef(sc[0:1], hp["max_grad_norm"] if hp["max_grad_norm"] else 0.0, sc[1:4], hp.get("grad_norm_skip", 0.0))
        st.step_count += 1
        T.adamw_step_(st.flat_p, st.flat_g, st.flat_m, st.flat_v, st.decay, lr=hp["lr"] if lr is None else lr, betas=hp["betas"], eps=hp["eps"],
                      weight_decay=hp["weight_decay"], step=st.step_count, norm_coef=sc[1:4], mirror=st.flat_bf)
        st.refresh_mirrors(cast=False)
        self.last_step_flags = sc[1:4]
        return sc[1]          # gradient norm (device scalar)

    def train_step(self, feats, feat_lengths, labels, lr=None):
        self.store.zero_grad()
        out = self.forward_backward(feats, feat_lengths, labels)
        out["grad_norm"] = self.optimizer_step(lr)
        return out
