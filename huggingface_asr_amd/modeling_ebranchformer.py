"""Drop-in `Wav2Vec2EBranchformerForCTC` whose forward runs on the HIP engine.

Mirrors the reference class surface (src/models/encoders/e_branchformer.py:408-496): same constructor, same
state-dict keys (huggingface_asr_amd/shapes.py), same forward signature / `CausalLMOutput`, same ValueError for
out-of-vocabulary labels, `freeze_encoder()`, `_get_feat_extract_output_lengths` (the UN-padded outer formula the
reference inherits from Wav2Vec2ForCTC) and `_get_feature_vector_attention_mask`.  The nn.Modules below only HOLD the
parameters (for state_dict / optimizers / checkpoint averaging); no tensor op of the forward goes through them.

Eval mode / no_grad: the inference engine (engine.py).  Training mode with labels: forward AND backward run on the HIP trainer
(train.py) behind a torch.autograd bridge (autograd_bridge.py), so `loss.backward()` fills ordinary `.grad`s for HF Trainer's
optimizer; dropout, LayerDrop, in-model SpecAugment (counter-based masks) and causal (streaming) encoders are supported; configurations the HIP training
step does not cover raise NotImplementedError rather than silently falling back to PyTorch."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn
from transformers import PreTrainedModel
from transformers.modeling_outputs import CausalLMOutput

from . import ops
from .configuration_ebranchformer import Wav2Vec2EBranchformerConfig
from .engine import EBranchformerEngine, cfg_from_hf


def _dropout_seed() -> int:
    """dropout-mask seed of the HIP training step: torch's seed (set by HF Trainer's set_seed) offset by the data-parallel rank"""
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    return (torch.initial_seed() + 7919 * rank) & 0xFFFFFFFF


class _Holder(nn.Module):
    """parameter container; calling it is a bug"""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the forward pass runs in the HIP engine")


def _ln(d):
    return nn.LayerNorm(d)


class _FFN(_Holder):
    def __init__(self, d, i):
        super().__init__()
        self.intermediate_dense = nn.Linear(d, i)
        self.output_dense = nn.Linear(i, d)


class _SelfAttn(_Holder):
    def __init__(self, cfg):
        super().__init__()
        d, H = cfg.hidden_size, cfg.num_attention_heads
        self.linear_q, self.linear_k, self.linear_v, self.linear_out = (nn.Linear(d, d) for _ in range(4))
        if cfg.position_embeddings_type == "relative":
            self.linear_pos = nn.Linear(d, d, bias=False)
            self.pos_bias_u = nn.Parameter(torch.zeros(H, d // H))
            self.pos_bias_v = nn.Parameter(torch.zeros(H, d // H))


class _CSGU(_Holder):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.intermediate_size // 2
        self.norm = _ln(c)
        self.conv = nn.Conv1d(c, c, cfg.csgu_kernel_size, 1, (cfg.csgu_kernel_size - 1) // 2, groups=c)
        if cfg.csgu_use_linear_after_conv:
            self.linear = nn.Linear(c, c)


class _CgMLP(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.channel_proj1 = nn.Sequential(nn.Linear(cfg.hidden_size, cfg.intermediate_size), nn.GELU())
        self.csgu = _CSGU(cfg)
        self.channel_proj2 = nn.Linear(cfg.intermediate_size // 2, cfg.hidden_size)


class _Layer(_Holder):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        if cfg.use_macaron_ff:
            self.ff1 = nn.Sequential(_ln(d), _FFN(d, cfg.intermediate_size))
        self.self_attn_layer_norm = _ln(d)
        self.self_attn = _SelfAttn(cfg)
        self.cgMLP = _CgMLP(cfg)
        self.cgMLP_layer_norm = _ln(d)
        self.merge_proj = nn.Linear(2 * d, d)
        self.depthwise_conv_fusion = nn.Conv1d(2 * d, 2 * d, cfg.merge_conv_kernel, 1, (cfg.merge_conv_kernel - 1) // 2, groups=2 * d)
        self.final_layer_norm = _ln(d)
        if cfg.use_macaron_ff:
            self.ff2 = nn.Sequential(_ln(d), _FFN(d, cfg.intermediate_size))


class _Gated(_Holder):      # GatedConv2d / GatedConv2dShared (extractors.py:23-54): .conv and .gate; the shared gate is a (4k, k) / stride (4s, s) / padding (4p, p) conv
    def __init__(self, cin, cout, k, s, p, share):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, (k, k), stride=(s, s), padding=p)
        self.gate = nn.Conv2d(cin, cout, (k * share, k), stride=(s * share, s), padding=(p * share, p))


class _ConvWrap(_Holder):   # ContextAwareConv2d: the extra ".conv" level of the non-causal key (extractors.py:57-65)
    def __init__(self, cin, cout, k, s, p=1, mode=0):
        super().__init__()
        # the reference's dict lookup: "gated" / "gated_shared" select a gate, anything else (None, the recipes' `shared_gated`, ...) is the plain nn.Conv2d
        self.conv = _Gated(cin, cout, k, s, p, 4 if mode == 2 else 1) if mode else nn.Conv2d(cin, cout, k, stride=s)


class _FeatureExtractor(_Holder):
    def __init__(self, cfg):
        super().__init__()
        from .shapes import context_mode, conv_freq_out
        blocks, cin = [], 1
        mode = context_mode(dict(is_causal=cfg.is_causal, context_awareness_type=getattr(cfg, "context_awareness_type", None)))
        for c, k, s, p in zip(cfg.conv_dim, cfg.conv_kernel, cfg.conv_stride, cfg.conv_padding):
            conv = nn.Conv2d(cin, c, k, stride=s) if cfg.is_causal else _ConvWrap(cin, c, k, s, p, mode)
            blocks.append(nn.Sequential(conv, nn.GELU()))
            cin = c
        self.conv = nn.Sequential(*blocks)
        fo = conv_freq_out(cfg.num_fbanks, cfg.conv_kernel, cfg.conv_stride, cfg.conv_padding)
        self.out = nn.Linear(cfg.conv_dim[-1] * fo, cfg.hidden_size)


class _FeatureProjection(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.projection = nn.Linear(cfg.hidden_size, cfg.hidden_size)


class _RotaryBuf(_Holder):
    def __init__(self, cfg):
        super().__init__()
        hd = cfg.hidden_size // cfg.num_attention_heads
        self.register_buffer("inv_freq", 1.0 / (cfg.rotary_embedding_base ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd)))


class _Encoder(_Holder):
    def __init__(self, cfg):
        super().__init__()
        if cfg.position_embeddings_type == "rotary":
            self.embed_positions = _RotaryBuf(cfg)       # persistent `inv_freq` buffer is part of the reference state dict
        self.layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _Wav2Vec2(_Holder):
    def __init__(self, cfg):
        super().__init__()
        self.feature_extractor = _FeatureExtractor(cfg)
        self.feature_projection = _FeatureProjection(cfg)
        if cfg.mask_time_prob > 0.0 or cfg.mask_feature_prob > 0.0:
            self.masked_spec_embed = nn.Parameter(torch.Tensor(cfg.hidden_size).uniform_())
        self.encoder = _Encoder(cfg)


class Wav2Vec2EBranchformerForCTC(PreTrainedModel):
    config_class = Wav2Vec2EBranchformerConfig
    base_model_prefix = "wav2vec2"
    main_input_name = "input_values"
    supports_gradient_checkpointing = False

    def __init__(self, config: Wav2Vec2EBranchformerConfig):
        super().__init__(config)
        self.wav2vec2 = _Wav2Vec2(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size)
        self.blank_projection = nn.Linear(config.hidden_size, 1)
        self._engine: Optional[EBranchformerEngine] = None
        self._engine_key = None
        from .autograd_bridge import detach_state_dict_views
        self._register_state_dict_hook(detach_state_dict_views)      # adopted parameters (views of the trainer's flat store) leave state_dict() / save_pretrained() as private copies
        self.post_init()

    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, (nn.Linear, nn.Conv1d, nn.Conv2d)):
            nn.init.normal_(module.weight, mean=0.0, std=std) if isinstance(module, nn.Linear) else nn.init.kaiming_normal_(module.weight)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight); nn.init.zeros_(module.bias)

    # ---- surface used by the reference's trainers / collators
    def freeze_encoder(self):
        for p in self.wav2vec2.encoder.parameters():
            p.requires_grad = False

    def freeze_feature_encoder(self):
        for p in self.wav2vec2.feature_extractor.parameters():
            p.requires_grad = False

    def get_output_embeddings(self):
        return None

    def _get_feat_extract_output_lengths(self, input_lengths, add_adapter=None):
        """Wav2Vec2ForCTC's UN-padded formula (quirk of SURVEY.md §8a row 8'): floor((L - k)/s) + 1 per conv."""
        for k, s in zip(self.config.conv_kernel, self.config.conv_stride):
            input_lengths = torch.div(input_lengths - k, s, rounding_mode="floor") + 1 if torch.is_tensor(input_lengths) \
                else (input_lengths - k) // s + 1
        return input_lengths

    def _get_feature_vector_attention_mask(self, feature_vector_length: int, attention_mask: torch.LongTensor, add_adapter=None):
        lens = self._get_feat_extract_output_lengths(attention_mask.sum(-1)).to(torch.long)
        return torch.arange(feature_vector_length, device=attention_mask.device)[None, :] < lens[:, None]

    # ---- engine plumbing
    def _param_list(self):
        """cached flat list of the parameters (walking the module tree per forward costs more than the check it feeds); parameters are never
        added or removed after construction"""
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = list(self.parameters())
            self.__dict__["_plist"] = pl
        return pl

    def _weights_version(self):
        return sum(p._version for p in self._param_list())

    def _get_engine(self, device) -> EBranchformerEngine:
        if self._engine is None or self._engine.device != torch.device(device):
            self._engine = EBranchformerEngine(cfg_from_hf(self.config), device)
            self._engine_key = None
        pl = self._param_list()
        from .autograd_bridge import bridge_generation
        key = (self._weights_version(), bridge_generation(self), pl[0].data_ptr(), pl[-1].data_ptr(), pl[len(pl) // 2].data_ptr())
        if key != self._engine_key:
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._engine_key = key
        return self._engine

    def _get_trainer(self, device):
        from .train import EncoderCTCTrainer
        if getattr(self, "_trainer", None) is None or self._trainer.device != torch.device(device):
            self._trainer = EncoderCTCTrainer(cfg_from_hf(self.config), device, dp_sync=False, seed=_dropout_seed())      # optimizer / all-reduce stay with the caller (HF Trainer)
        return self._trainer

    def _training_forward(self, input_values, attention_mask, labels, output_hidden_states, return_dict):
        """training step through the HIP trainer; gradients reach the nn.Parameters via autograd_bridge.HipStep."""
        from .autograd_bridge import run_training_forward
        if labels is None:
            raise NotImplementedError("HIP training forward needs `labels` (the CTC loss is the only differentiable output of this head)")
        chk = getattr(self, "_label_check", None)
        if chk is None:
            from .autograd_bridge import LabelRangeCheck
            chk = LabelRangeCheck(self.config.vocab_size, "vocab_size")
            object.__setattr__(self, "_label_check", chk)
        chk.submit(labels)                                 # the reference's range check (e_branchformer.py:461-462) without its per-step host sync
        labels = labels.clamp(max=self.config.vocab_size - 1)   # a bad batch is reported one forward later: until then its labels must stay inside the logits row
        tr = self._get_trainer(input_values.device)
        feat_len = attention_mask.sum(-1).to(torch.int32) if attention_mask is not None else None

        def step(t):
            t.store.zero_grad()
            return t.forward_backward(input_values, feat_len, labels.to(input_values.device), keep_hidden=bool(output_hidden_states))
        loss, out = run_training_forward(self, tr, step)
        hidden_states = (out["last_hidden"],) if output_hidden_states else None
        if not return_dict:
            return (loss, out["logits"]) + ((hidden_states,) if hidden_states is not None else ())
        return CausalLMOutput(loss=loss, logits=out["logits"], hidden_states=hidden_states, attentions=None)

    def forward(
        self,
        input_values: Optional[torch.Tensor],
        attention_mask: Optional[torch.Tensor] = None,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        return_dict: Optional[bool] = None,
        labels: Optional[torch.Tensor] = None,
        **kwargs,
    ) -> Union[Tuple, CausalLMOutput]:
        return_dict = return_dict if return_dict is not None else getattr(self.config, "return_dict", True)
        if output_attentions:
            raise NotImplementedError("attention probabilities are never materialised by the fused HIP attention kernel")
        if not input_values.is_cuda:
            raise RuntimeError("Wav2Vec2EBranchformerForCTC (HIP): inputs must be on the GPU; there is no CPU fallback")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._training_forward(input_values, attention_mask, labels, output_hidden_states, return_dict)
        eng = self._get_engine(input_values.device)
        feat_len = attention_mask.sum(-1).to(torch.int32) if attention_mask is not None else None
        out = eng.forward(input_values, feat_len, want_hidden=True, want_all_hidden=bool(output_hidden_states))
        logits = out["logits"]
        loss = None
        if labels is not None:
            if getattr(self, "_label_check", None) is not None:
                self._label_check.flush()                   # a bad TRAINING batch still pending: report it now
            if labels.max() >= self.config.vocab_size:      # same check / same sync point as e_branchformer.py:461-462
                raise ValueError(f"Label values must be <= vocab_size: {self.config.vocab_size}")
            loss, _, _ = ops.ctc_loss(logits, labels.to(logits.device), out["outer_len"],
                                      reduction=self.config.ctc_loss_reduction, zero_infinity=self.config.ctc_zero_infinity, lse=out.get("lse"))
        hidden_states = out["hidden_states"] if output_hidden_states else None          # L + 1 tensors, as HF (tf:685-713)
        if not return_dict:
            output = (logits,) + ((hidden_states,) if hidden_states is not None else ())
            return ((loss,) + output) if loss is not None else output
        return CausalLMOutput(loss=loss, logits=logits, hidden_states=hidden_states, attentions=None)
