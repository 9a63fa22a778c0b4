"""Whisper-style encoder on the HIP kernels (BASELINE.json config 4).

The reference uses HuggingFace's Whisper classes as they are (`src/utilities/model_utils.py:183`,
`src/trainers/train_enc_dec_asr.py:82-83`, `configs/default_data_preprocessing_whisper.json`), so the drop-in unit here is an
engine that takes a `WhisperEncoder` state dict and reproduces `WhisperEncoder.forward` / `WhisperFeatureExtractor`:
Conv1d(80->d,k3)+GELU and Conv1d(d->d,k3,s2)+GELU as implicit GEMMs over a channels-last layout, + learned positions,
pre-LN layers (fused QKV GEMM with a zero bias block for k_proj, LDS-staged attention hd=64, out-proj / FFN GEMMs with the
residual fused), final LayerNorm."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops

BF16 = torch.bfloat16


class WhisperFrontend:
    """log-mel on device: (B, N) fp32 waveforms -> (B, 80, 3000) fp32 `input_features` (+ channels-last bf16 for conv1)."""

    def __init__(self, num_mel=80, n_samples=480000, sr=16000):
        self.num_mel, self.n_samples = num_mel, n_samples
        n = np.arange(400)
        self.window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / 400)                 # periodic hann
        self.twiddle = np.stack([np.cos(2 * np.pi * n / 400), np.sin(2 * np.pi * n / 400)], 1)
        self.filters = self._slaney(num_mel, sr)
        self._dev = {}

    @staticmethod
    def _slaney(num_mel, sr, fmin=0.0, fmax=8000.0):
        def hz2mel(f):
            f = np.asarray(f, dtype=np.float64)
            return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

        def mel2hz(m):
            m = np.asarray(m, dtype=np.float64)
            return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)
        ff = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), num_mel + 2))
        fft = np.linspace(0, sr // 2, 201)
        diff = np.diff(ff)
        sl = ff[None, :] - fft[:, None]
        fb = np.maximum(0.0, np.minimum(-sl[:, :-2] / diff[:-1], sl[:, 2:] / diff[1:]))
        return fb * (2.0 / (ff[2: num_mel + 2] - ff[:num_mel]))[None, :]

    def _tables(self, device):
        k = str(device)
        if k not in self._dev:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
            self._dev[k] = (t(self.window), t(self.twiddle), t(self.filters.T.copy()))
        return self._dev[k]

    def __call__(self, wave: torch.Tensor, num_samples=None, want_features=True):
        if not wave.is_cuda:
            raise RuntimeError("WhisperFrontend needs device tensors (no CPU fallback)")
        B, N = wave.shape
        win, tw, mel = self._tables(wave.device)
        frames = self.n_samples // 160
        ns = num_samples.to(torch.int32) if num_samples is not None else torch.full((B,), min(N, self.n_samples), dtype=torch.int32, device=wave.device)
        scratch = torch.empty(B * frames * self.num_mel + B, dtype=torch.float32, device=wave.device)       # log-mel before the clamp + the per-clip maxima
        feats = torch.empty((B, self.num_mel, frames), dtype=torch.float32, device=wave.device) if want_features else None
        cl = torch.empty((B, frames, self.num_mel), dtype=BF16, device=wave.device)
        rc = _lib.lib().mi_whisper_logmel(wave.data_ptr(), wave.stride(0), ns.data_ptr(), self.n_samples, win.data_ptr(), tw.data_ptr(),
                                          mel.data_ptr(), self.num_mel, B, scratch.data_ptr(), 0 if feats is None else feats.data_ptr(),
                                          cl.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mi_whisper_logmel")
        return feats, cl


class WhisperEncoderEngine:
    def __init__(self, cfg: dict, device="cuda:0"):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if cfg["d_model"] // cfg["encoder_attention_heads"] not in (64, 128):
            raise NotImplementedError("head size must be 64 or 128")
        self.w = None
        self.tail_split = True          # GEMM dispatch: give an under-filled last round of 256 x 256 tiles to the 128 x 128 kernel (gemm_glds.hip); set False when several batches are in flight on streams

    def load_state_dict(self, sd: dict, prefix: str = ""):
        dev, c = self.device, self.cfg
        d = c["d_model"]
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        bf = lambda t: t.detach().to(dev, torch.float32).to(BF16).contiguous()
        g = lambda n: sd[prefix + n]
        w = dict(
            c1w=bf(g("conv1.weight").permute(0, 2, 1).reshape(d, -1)), c1b=f32(g("conv1.bias")),          # (d, k*mel): k-major, channel-minor
            c2w=bf(g("conv2.weight").permute(0, 2, 1).reshape(d, -1)), c2b=f32(g("conv2.bias")),
            pos=f32(g("embed_positions.weight")), lnf=(f32(g("layer_norm.weight")), f32(g("layer_norm.bias"))), layers=[])
        for l in range(c["encoder_layers"]):
            p = f"{prefix}layers.{l}."
            wqkv = bf(torch.cat([sd[p + f"self_attn.{n}_proj.weight"].detach().to(dev) for n in "qkv"], 0))
            bqkv = f32(torch.cat([sd[p + "self_attn.q_proj.bias"].detach().to(dev), torch.zeros(d, device=dev),
                                  sd[p + "self_attn.v_proj.bias"].detach().to(dev)], 0))                   # k_proj has no bias
            w["layers"].append(dict(
                ln1=(f32(sd[p + "self_attn_layer_norm.weight"]), f32(sd[p + "self_attn_layer_norm.bias"])), wqkv=wqkv, bqkv=bqkv,
                wo=bf(sd[p + "self_attn.out_proj.weight"]), bo=f32(sd[p + "self_attn.out_proj.bias"]),
                ln2=(f32(sd[p + "final_layer_norm.weight"]), f32(sd[p + "final_layer_norm.bias"])),
                w1=bf(sd[p + "fc1.weight"]), b1=f32(sd[p + "fc1.bias"]), w2=bf(sd[p + "fc2.weight"]), b2=f32(sd[p + "fc2.bias"])))
        self.w = w

    def forward(self, input_features: torch.Tensor | None = None, features_cl: torch.Tensor | None = None) -> torch.Tensor:
        """input_features (B, mel, T) fp32 (HF layout) or features_cl (B, T, mel) bf16 -> last_hidden_state (B, T/2, d) fp32."""
        c, w, dev = self.cfg, self.w, self.device
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        d, H = c["d_model"], c["encoder_attention_heads"]
        if features_cl is None:
            if not input_features.is_cuda:
                raise RuntimeError("WhisperEncoderEngine needs device tensors (no CPU fallback)")
            B, mel, T = input_features.shape
            x32 = input_features.to(torch.float32).contiguous()
            features_cl = torch.empty((B, T, mel), dtype=BF16, device=dev)
            _lib.check(L.mi_transpose_cast_bct_btc(x32.data_ptr(), features_cl.data_ptr(), B, mel, T, st), "mi_transpose_cast_bct_btc")
        B, T, mel = features_cl.shape
        P = w["pos"].shape[0]
        if T != 2 * P:
            raise ValueError(f"Whisper expects the mel input features to be of length {2 * P}, but found {T}")
        h1 = ops.conv2d_cl(features_cl.view(B, T, 1, mel), w["c1w"], w["c1b"], K=(3, 1), stride=1, pad=(1, 0))
        h2 = ops.conv2d_cl(h1, w["c2w"], w["c2b"], K=(3, 1), stride=2, pad=(1, 0))
        T2 = h2.shape[1]
        M = B * T2
        x = torch.empty((M, d), dtype=torch.float32, device=dev)
        _lib.check(L.mi_add_positions(h2.data_ptr(), w["pos"].data_ptr(), x.data_ptr(), M, T2, d, st), "mi_add_positions")
        a = torch.empty((M, d), dtype=BF16, device=dev)
        gv = 0 if self.tail_split else 48
        for lw in w["layers"]:
            ops.layernorm_chain(x, lna=lw["ln1"], outa=a)
            qkv = ops.gemm(a, lw["wqkv"], lw["bqkv"], variant=gv)
            ctx = ops.attention_qkv(qkv, B, T2, H)
            ops.gemm(ctx, lw["wo"], lw["bo"], out=x, resid=x, alpha=1.0, variant=gv)
            ops.layernorm_chain(x, lna=lw["ln2"], outa=a)
            m = ops.gemm(a, lw["w1"], lw["b1"], act="gelu", variant=gv)
            ops.gemm(m, lw["w2"], lw["b2"], out=x, resid=x, alpha=1.0, variant=gv)
        out = torch.empty((M, d), dtype=torch.float32, device=dev)
        ops.layernorm_chain(x, lna=w["lnf"], outa32=out)
        return out.view(B, T2, d)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# The Whisper branch of the reference (`src/utilities/model_utils.py:183`: `AutoModelForSpeechSeq2Seq.from_pretrained(...)` on a Whisper checkpoint, e.g.
# `recipes_v0.0.1/decred/out_of_domain/decode_whisper_lumi.sh:60-66` `--from_pretrained=openai/whisper-medium --predict_with_generate`) runs HuggingFace's
# `WhisperForConditionalGeneration` as it is, and `src/trainers/train_enc_dec_asr.py:82-83` tests `isinstance(model, WhisperForConditionalGeneration)` — so the class has
# to stay HuggingFace's.  The drop-in is therefore a replacement of `WhisperEncoder.forward`: the encoder (the whole cost of config 4) runs on the HIP engine above, the
# decoder, `generate`, the loss and the checkpoint format stay transformers' own.
def _encoder_cfg(enc) -> dict:
    c = enc.config
    return dict(d_model=c.d_model, encoder_layers=c.encoder_layers, encoder_attention_heads=c.encoder_attention_heads, encoder_ffn_dim=c.encoder_ffn_dim)


def _engine_for(enc) -> WhisperEncoderEngine:
    """the HIP engine of a `WhisperEncoder` module: built at the first forward, rebuilt when a parameter was replaced, moved or written in place"""
    params = list(enc.parameters())
    key = (str(params[0].device), tuple((p.data_ptr(), p._version) for p in params))
    cached = enc.__dict__.get("_hfasr_engine")
    if cached is not None and cached[0] == key:
        return cached[1]
    eng = WhisperEncoderEngine(_encoder_cfg(enc), params[0].device)
    eng.load_state_dict(enc.state_dict())
    enc.__dict__["_hfasr_engine"] = (key, eng)
    return eng


def _stock_forward(self, why, input_features, attention_mask, kwargs):
    """What the HIP engine does not cover runs transformers' OWN `WhisperEncoder.forward` — the code the reference runs for its Whisper branch anyway (the class is
    transformers', not ours: replacing its forward process-wide must not take away what worked before).  Said once per reason; `HFASR_WHISPER_STRICT=1` raises
    instead (what the GPU parity tests run under, so that a silent PyTorch pass cannot stand in for the HIP encoder)."""
    import os
    import warnings
    if os.environ.get("HFASR_WHISPER_STRICT") == "1":
        raise NotImplementedError(f"huggingface_asr_amd: the HIP Whisper encoder does not cover this call ({why}) and HFASR_WHISPER_STRICT=1 forbids transformers' own forward")
    if why not in _stock_forward.said:
        _stock_forward.said.add(why)
        warnings.warn(f"huggingface_asr_amd: WhisperEncoder.forward runs transformers' own PyTorch implementation for this call ({why}); the HIP engine covers "
                      "inference on GPU tensors returning last_hidden_state", stacklevel=3)
    from transformers.models.whisper import modeling_whisper as MW
    return MW.WhisperEncoder._hfasr_reference_forward(self, input_features, attention_mask=attention_mask, **kwargs)


_stock_forward.said = set()


def hip_whisper_encoder_forward(self, input_features, attention_mask=None, **kwargs):
    """`transformers.models.whisper.modeling_whisper.WhisperEncoder.forward` on the HIP engine: inference (`eval()`, or no gradient wanted) on GPU tensors returning
    `last_hidden_state`.  Every other call — training mode (dropout / LayerDrop / autograd through the encoder: `train_enc_dec_asr.py --do_train` on a Whisper checkpoint,
    `recipes_v0.0.1/librispeech_whisper_ctc`), attention or hidden-state outputs, head masks, CPU tensors — is handed to transformers' own forward (`_stock_forward`)."""
    from transformers.modeling_outputs import BaseModelOutput
    cfg = self.config
    want = {k: kwargs.get(k) for k in ("output_attentions", "output_hidden_states", "head_mask")}
    if want["output_attentions"] or want["output_hidden_states"] or getattr(cfg, "output_attentions", False) or getattr(cfg, "output_hidden_states", False) \
            or want["head_mask"] is not None:
        return _stock_forward(self, "attentions / hidden states / head mask requested", input_features, attention_mask, kwargs)
    if not input_features.is_cuda:
        return _stock_forward(self, "CPU tensors", input_features, attention_mask, kwargs)
    if self.training:
        return _stock_forward(self, "training mode", input_features, attention_mask, kwargs)
    if torch.is_grad_enabled() and (input_features.requires_grad or any(p.requires_grad for p in self.parameters())):
        return _stock_forward(self, "a gradient through the encoder is wanted (wrap inference in torch.no_grad())", input_features, attention_mask, kwargs)
    out = _engine_for(self).forward(input_features=input_features)                       # (B, T/2, d) fp32; same length check / message as transformers
    return BaseModelOutput(last_hidden_state=out.to(self.layer_norm.weight.dtype))


def install_whisper():
    """Give transformers' `WhisperEncoder` the HIP forward (idempotent).  The original stays reachable as `WhisperEncoder._hfasr_reference_forward` (tests compare against it)."""
    from transformers.models.whisper import modeling_whisper as MW
    if getattr(MW.WhisperEncoder.forward, "_hfasr_hip", False):
        return
    MW.WhisperEncoder._hfasr_reference_forward = MW.WhisperEncoder.forward
    hip_whisper_encoder_forward._hfasr_hip = True
    MW.WhisperEncoder.forward = hip_whisper_encoder_forward
