"""Drop-in `JointCTCAttentionEncoderDecoder` (+ configs) for the reference's joint CTC/attention model surface.

Mirrors reference `src/models/ctc_encoder_plus_autoregressive_decoder.py` (config :38-40, output dataclass :43-47, model :55-482)
and the decoder config of `src/models/decoders/multi_head_gpt2.py:12-29`: same model types (so `AutoConfig` /
`AutoModelForSpeechSeq2Seq` route here after `bind_all()`), same state-dict keys (encoder.* = our E-Branchformer CTC model,
decoder.* = GPT-2 multi-head keys, enc_to_dec_proj.*), same forward arguments and `Seq2SeqLMOutputLosses` fields, and a
`generate()` that performs the reference's joint CTC/attention greedy / beam decoding.  The modules only hold parameters;
all tensor work runs in `huggingface_asr_amd.decoder` (HIP kernels); training-mode forwards run forward AND backward on the HIP trainer (autograd_bridge.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn
from transformers import GPT2Config, PreTrainedModel
from transformers.modeling_outputs import Seq2SeqLMOutput
from transformers.models.speech_encoder_decoder.configuration_speech_encoder_decoder import SpeechEncoderDecoderConfig

from .configuration_ebranchformer import Wav2Vec2EBranchformerConfig
from .decoder import JointAEDEngine, generate as _generate
from .engine import cfg_from_hf
from .modeling_ebranchformer import _dropout_seed, Wav2Vec2EBranchformerForCTC, _Holder


class GPT2MultiHeadConfig(GPT2Config):
    model_type = "gpt2-multi-head"

    def __init__(self, head_locations=None, head_weights=None, tie_additional_weights=False, average_logits=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.head_locations = head_locations
        self.head_weights = head_weights
        self.tie_additional_weights = tie_additional_weights
        self.average_logits = average_logits


class JointCTCAttentionEncoderDecoderConfig(SpeechEncoderDecoderConfig):
    model_type = "joint_aed_ctc_speech-encoder-decoder"
    is_composition = True


@dataclass
class Seq2SeqLMOutputLosses(Seq2SeqLMOutput):
    enc_loss: Optional[torch.FloatTensor] = None
    dec_loss: Optional[torch.FloatTensor] = None
    encoder_logits: Optional[torch.FloatTensor] = None


class _Conv1D(_Holder):          # transformers Conv1D parameter layout: weight (in, out)
    def __init__(self, nf, nx):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))


class _Attn(_Holder):
    def __init__(self, d, cross):
        super().__init__()
        if cross:
            self.c_attn = _Conv1D(2 * d, d)
            self.q_attn = _Conv1D(d, d)
        else:
            self.c_attn = _Conv1D(3 * d, d)
        self.c_proj = _Conv1D(d, d)


class _MLP(_Holder):
    def __init__(self, d, inner):
        super().__init__()
        self.c_fc = _Conv1D(inner, d)
        self.c_proj = _Conv1D(d, inner)


class _Block(_Holder):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        eps = cfg.layer_norm_epsilon
        self.ln_1 = nn.LayerNorm(d, eps=eps)
        self.attn = _Attn(d, False)
        self.ln_2 = nn.LayerNorm(d, eps=eps)
        self.crossattention = _Attn(d, True)
        self.ln_cross_attn = nn.LayerNorm(d, eps=eps)
        self.mlp = _MLP(d, cfg.n_inner if cfg.n_inner is not None else 4 * d)


class _AdaptiveEmb(_Holder):     # reference src/models/embeddings.py:5-31 (div_val = 1): key `emb_layers.0.weight`
    def __init__(self, n_token, d):
        super().__init__()
        self.emb_layers = nn.ModuleList([nn.Embedding(n_token, d)])
        self.emb_projs = nn.ParameterList()


class _FixedPos(_Holder):        # reference src/models/embeddings.py:65-86: persistent buffer `inv_freq`
    def __init__(self, d):
        super().__init__()
        self.register_buffer("inv_freq", 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d)))


class _GPT2Body(_Holder):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.hidden_size
        if getattr(cfg, "pos_emb_fixed", False):
            self.wte = _AdaptiveEmb(cfg.vocab_size, d)
            self.wpe = _FixedPos(d)
        else:
            self.wte = nn.Embedding(cfg.vocab_size, d)
            self.wpe = nn.Embedding(cfg.max_position_embeddings, d)
        self.h = nn.ModuleList([_Block(cfg) for _ in range(cfg.num_hidden_layers)])
        self.ln_f = nn.LayerNorm(d, eps=cfg.layer_norm_epsilon)


class GPT2LMMultiHeadModel(PreTrainedModel):
    """Parameter holder with the reference decoder's class name, config class and state-dict keys (`src/models/decoders/multi_head_gpt2.py:31-49`;
    with `pos_emb_fixed` the embeddings the reference's `PositionalEncodingInitModifier` swaps in, `src/models/auto_wrappers.py:186-209`).  It is what
    `CustomModelForCausalLM.from_config / from_pretrained` return after `bind.install()`; the tensor work of the decoder runs in
    `huggingface_asr_amd.decoder` when the module sits inside `JointCTCAttentionEncoderDecoder` — called on its own it raises."""
    config_class = GPT2MultiHeadConfig
    base_model_prefix = "transformer"
    main_input_name = "input_ids"
    _no_split_modules = []

    def __init__(self, config):
        super().__init__(config)
        if config.head_locations is not None and len(config.head_locations) + 1 != len(config.head_weights or []):
            raise ValueError("The number of head locations should be equal to the number of head weights minus 1")     # reference :36-37
        self.transformer = _GPT2Body(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.additional_lm_heads = nn.ModuleList([nn.Linear(config.hidden_size, config.vocab_size, bias=False) for _ in (config.head_locations or [])])
        self.head_locations = list(config.head_locations or [])
        self.head_weights = list(config.head_weights or [1.0])
        self.lsm_factor = getattr(config, "lsm_factor", 0.0)
        self.post_init()

    def _init_weights(self, module):          # GPT-2's initialiser (tf:models/gpt2/modeling_gpt2.py `_init_weights`): N(0, initializer_range), LayerNorm 1 / 0
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=std)
            if getattr(module, "bias", None) is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.weight.data.fill_(1.0)
            module.bias.data.zero_()

    def get_input_embeddings(self):
        return self.transformer.wte

    def get_output_embeddings(self):
        return self.lm_head

    def tie_weights(self, *a, **k):           # the joint model forces tie_word_embeddings = False (reference ctc_encoder_plus...:90)
        pass

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("GPT2LMMultiHeadModel (HIP): the decoder runs inside JointCTCAttentionEncoderDecoder (huggingface_asr_amd.decoder); "
                           "it has no stand-alone forward and no CPU fallback")


_Decoder = GPT2LMMultiHeadModel


def _dec_cfg_dict(c) -> dict:
    return dict(vocab_size=c.vocab_size, n_embd=c.hidden_size, n_layer=c.num_hidden_layers, n_head=c.num_attention_heads,
                n_positions=c.max_position_embeddings, head_locations=list(c.head_locations or []),
                head_weights=list(c.head_weights or [1.0]), lsm_factor=getattr(c, "lsm_factor", 0.0),
                layer_norm_epsilon=c.layer_norm_epsilon, pos_emb_fixed=bool(getattr(c, "pos_emb_fixed", False)),
                activation_function=c.activation_function)


class JointCTCAttentionEncoderDecoder(PreTrainedModel):
    config_class = JointCTCAttentionEncoderDecoderConfig
    base_model_prefix = "joint_aed_ctc_speech-encoder-decoder"
    main_input_name = "inputs"

    def __init__(self, config=None, encoder=None, decoder=None):
        if config is None and (encoder is None or decoder is None):
            raise ValueError("Either a configuration or an encoder and a decoder has to be provided.")
        if config is None:
            config = JointCTCAttentionEncoderDecoderConfig.from_encoder_decoder_configs(encoder.config, decoder.config)
        elif not isinstance(config, self.config_class):
            raise ValueError(f"Config: {config} has to be of type {self.config_class}")
        for k, v in (("ctc_weight", 0.0), ("shared_lm_head", False)):      # set by the reference's instantiate_aed_model (model_utils.py:159-174)
            if not hasattr(config, k):
                setattr(config, k, v)
        if not hasattr(config.decoder, "lsm_factor"):
            config.decoder.lsm_factor = 0.0
        xh = getattr(config.decoder, "cross_attention_hidden_size", None)      # a PretrainedConfig default in transformers 4.x, absent in 5.x
        if xh is not None and xh != config.encoder.hidden_size:
            raise ValueError("If `cross_attention_hidden_size` is specified in the decoder's configuration, it has to be equal to the "
                             "encoder's `hidden_size`.")
        config.tie_word_embeddings = False
        super().__init__(config)
        if encoder is not None and not isinstance(encoder, Wav2Vec2EBranchformerForCTC):
            raise TypeError(f"encoder must be the HIP Wav2Vec2EBranchformerForCTC, got {type(encoder)} (no PyTorch fallback)")
        if decoder is not None and not isinstance(decoder, GPT2LMMultiHeadModel):
            raise TypeError(f"decoder must be the HIP GPT2LMMultiHeadModel, got {type(decoder)} (no PyTorch fallback)")
        self.encoder = encoder if encoder is not None else Wav2Vec2EBranchformerForCTC(config.encoder)
        self.decoder = decoder if decoder is not None else GPT2LMMultiHeadModel(config.decoder)
        self.encoder.config = self.config.encoder
        self.decoder.config = self.config.decoder
        self.encoder_output_dim = getattr(config.encoder, "output_hidden_size", config.encoder.hidden_size)
        if self.encoder_output_dim != self.decoder.config.hidden_size and xh is None:
            self.enc_to_dec_proj = nn.Linear(self.encoder.config.hidden_size, self.decoder.config.hidden_size)
        if self.encoder.get_output_embeddings() is not None:
            raise ValueError(f"The encoder {self.encoder} should not have a LM Head. Please use a model without LM Head")
        self.enc_loss_weight = config.ctc_weight
        self.dec_loss_weight = 1 - config.ctc_weight
        self.lsm_factor = config.decoder.lsm_factor
        if getattr(config, "shared_lm_head", False):                  # reference :132-133
            self.encoder.lm_head.weight = self.decoder.lm_head.weight
        self._engine = None
        self._engine_key = None
        from .autograd_bridge import detach_state_dict_views
        self._register_state_dict_hook(detach_state_dict_views)      # adopted parameters (views of the trainers' flat stores, GPT-2 Conv1D `.t()` views) leave as private contiguous copies
        from .decoding import GenerationConfigCustom
        # trainers overwrite this (train_enc_dec_asr.py:85); the default carries the ids of the model config
        self.generation_config = GenerationConfigCustom(pad_token_id=config.pad_token_id, eos_token_id=config.decoder.eos_token_id,
                                                        decoder_start_token_id=config.decoder_start_token_id, num_beams=1, max_length=64)
        self.post_init()

    def _init_weights(self, module):
        pass

    @classmethod
    def from_encoder_decoder_pretrained(cls, encoder_pretrained_model_name_or_path=None, decoder_pretrained_model_name_or_path=None, *model_args, **kwargs):
        """Reference `ctc_encoder_plus_autoregressive_decoder.py:138-235` (called by `instantiate_aed_model`, `model_utils.py:199-204`): kwargs
        prefixed `encoder_` / `decoder_` (except `decoder_start_token_id`) go to the sub-model configs, the rest to the joint config; `encoder_model` /
        `decoder_model` pass ready modules.  Sub-models are loaded through the Auto registry (`bind_all()` must have run) and must resolve to the HIP classes."""
        from transformers import AutoConfig, AutoModelForCTC
        kw_enc = {k[len("encoder_"):]: v for k, v in kwargs.items() if k.startswith("encoder_")}
        kw_dec = {k[len("decoder_"):]: v for k, v in kwargs.items() if k.startswith("decoder_") and k != "decoder_start_token_id"}
        for k in kw_enc:
            del kwargs["encoder_" + k]
        for k in kw_dec:
            del kwargs["decoder_" + k]
        encoder = kw_enc.pop("model", None)
        if encoder is None:
            if encoder_pretrained_model_name_or_path is None:
                raise ValueError("If `encoder_model` is not defined as an argument, a `encoder_pretrained_model_name_or_path` has to be defined.")
            if "config" not in kw_enc:
                enc_cfg, kw_enc = AutoConfig.from_pretrained(encoder_pretrained_model_name_or_path, **kw_enc, return_unused_kwargs=True)
                if getattr(enc_cfg, "is_decoder", False) or getattr(enc_cfg, "add_cross_attention", False):
                    enc_cfg.is_decoder = False
                    enc_cfg.add_cross_attention = False
                kw_enc["config"] = enc_cfg
            encoder = AutoModelForCTC.from_pretrained(encoder_pretrained_model_name_or_path, *model_args, **kw_enc)
        decoder = kw_dec.pop("model", None)
        if decoder is None:
            if decoder_pretrained_model_name_or_path is None:
                raise ValueError("If `decoder_model` is not defined as an argument, a `decoder_pretrained_model_name_or_path` has to be defined.")
            if "config" not in kw_dec:
                dec_cfg, kw_dec = AutoConfig.from_pretrained(decoder_pretrained_model_name_or_path, **kw_dec, return_unused_kwargs=True)
                dec_cfg.is_decoder = True
                dec_cfg.add_cross_attention = True
                kw_dec["config"] = dec_cfg
            if not isinstance(kw_dec["config"], GPT2MultiHeadConfig):
                raise ValueError(f"decoder config {type(kw_dec['config'])} is not a GPT2MultiHeadConfig: only the multi-head GPT-2 decoder has a HIP implementation")
            decoder = GPT2LMMultiHeadModel.from_pretrained(decoder_pretrained_model_name_or_path, **kw_dec)
        config = JointCTCAttentionEncoderDecoderConfig.from_encoder_decoder_configs(encoder.config, decoder.config, **kwargs)
        config.tie_word_embeddings = False
        return cls(encoder=encoder, decoder=decoder, config=config)

    def get_encoder(self):
        return self.encoder

    def get_decoder(self):
        return self.decoder

    def freeze_feature_encoder(self):
        self.encoder.freeze_feature_encoder()

    def _get_engine(self, device) -> JointAEDEngine:
        if self._engine is None or self._engine.device != torch.device(device):
            jc = dict(ctc_weight=self.config.ctc_weight, pad_token_id=self.config.pad_token_id,
                      decoder_start_token_id=self.config.decoder_start_token_id)
            self._engine = JointAEDEngine(cfg_from_hf(self.config.encoder), _dec_cfg_dict(self.config.decoder), jc, device)
            self._engine_key = None
        from .autograd_bridge import bridge_generation
        key = (sum(p._version for p in self.parameters()), bridge_generation(self), tuple(p.data_ptr() for p in self.parameters()))
        if key != self._engine_key:
            self._engine.load_state_dict(dict(self.state_dict()))
            self._engine_key = key
        return self._engine

    def _get_trainer(self, device):
        from .train_aed import JointAEDTrainer
        if getattr(self, "_trainer", None) is None or self._trainer.device != torch.device(device):
            jc = dict(ctc_weight=self.config.ctc_weight, pad_token_id=self.config.pad_token_id,
                      decoder_start_token_id=self.config.decoder_start_token_id)
            dc = dict(_dec_cfg_dict(self.config.decoder), tie_word_embeddings=False,
                      resid_pdrop=getattr(self.config.decoder, "resid_pdrop", 0.0), embd_pdrop=getattr(self.config.decoder, "embd_pdrop", 0.0),
                      attn_pdrop=getattr(self.config.decoder, "attn_pdrop", 0.0))
            self._trainer = JointAEDTrainer(cfg_from_hf(self.config.encoder), dc, jc, device, with_proj=hasattr(self, "enc_to_dec_proj"), dp_sync=False, seed=_dropout_seed())
        return self._trainer

    @staticmethod
    def _pick_inputs(inputs, input_values, input_features):
        if inputs is None:
            if input_values is not None and input_features is not None:
                raise ValueError("You cannot specify both input_values and input_features at the same time")
            inputs = input_values if input_values is not None else input_features
            if inputs is None:
                raise ValueError("You have to specify either input_values or input_features")
        return inputs

    def forward(self, inputs=None, attention_mask=None, decoder_input_ids=None, decoder_attention_mask=None, encoder_outputs=None,
                past_key_values=None, decoder_inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, input_values=None, input_features=None, return_dict=None, **kwargs):
        training = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if labels is None or decoder_input_ids is not None or encoder_outputs is not None or decoder_inputs_embeds is not None:
            raise NotImplementedError("HIP joint forward implements the teacher-forced path driven by `labels` (reference :303-304); "
                                      "use generate() for decoding")
        inputs = self._pick_inputs(inputs, input_values, input_features)
        if not inputs.is_cuda:
            raise RuntimeError("JointCTCAttentionEncoderDecoder (HIP): inputs must be on the GPU; there is no CPU fallback")
        chk = getattr(self, "_label_check", None)
        if training:                                       # the range check without a host sync per step (autograd_bridge.LabelRangeCheck): reported one forward later
            if chk is None:
                from .autograd_bridge import LabelRangeCheck
                chk = LabelRangeCheck(self.config.encoder.vocab_size, "vocab_size")
                object.__setattr__(self, "_label_check", chk)
            chk.submit(labels)
            labels = labels.clamp(max=self.config.encoder.vocab_size - 1)      # reported one forward later: until then the labels must stay inside the vocabulary
        else:
            if chk is not None:
                chk.flush()
            if labels.max() >= self.config.encoder.vocab_size:
                raise ValueError(f"Label values must be <= vocab_size: {self.config.encoder.vocab_size}")
        fl = attention_mask.sum(-1).to(torch.int32) if attention_mask is not None else None
        if training:                                       # forward + backward on the HIP trainer, gradients handed to autograd
            from .autograd_bridge import run_training_forward
            tr = self._get_trainer(inputs.device)

            def step(t):
                t.enc.store.zero_grad(); t.store.zero_grad()
                return t.forward_backward(inputs, fl, labels.to(inputs.device))
            loss, out = run_training_forward(self, tr, step)
            out = dict(out, loss=loss)
        else:
            out = self._get_engine(inputs.device).forward(inputs, fl, labels.to(inputs.device))
        B, T2 = out["encoder_logits"].shape[:2]
        return Seq2SeqLMOutputLosses(loss=out["loss"], enc_loss=out["enc_loss"], dec_loss=out["dec_loss"], logits=out["logits"],
                                     encoder_last_hidden_state=out["encoder_hidden"].view(B, T2, -1), encoder_logits=out["encoder_logits"])

    # generation options that change the decoding and that the HIP loop does not implement: name -> the values that leave them off
    _GENERATION_OFF = {"do_sample": (None, False), "num_beam_groups": (None, 1), "penalty_alpha": (None,), "dola_layers": (None,), "min_length": (None, 0),
                       "min_new_tokens": (None, 0), "repetition_penalty": (None, 1.0), "no_repeat_ngram_size": (None, 0), "encoder_no_repeat_ngram_size": (None, 0),
                       "bad_words_ids": (None,), "force_words_ids": (None,), "constraints": (None,), "forced_bos_token_id": (None,), "forced_eos_token_id": (None,),
                       "remove_invalid_values": (None, False), "exponential_decay_length_penalty": (None,), "suppress_tokens": (None,), "begin_suppress_tokens": (None,),
                       "sequence_bias": (None,), "guidance_scale": (None, 1.0), "renormalize_logits": (None, False), "diversity_penalty": (None, 0.0),
                       "lm_weight": (None, 0, 0.0), "output_logits": (None, False), "low_memory": (None, False), "token_healing": (None, False),
                       "max_time": (None,), "stop_strings": (None,), "prompt_lookup_num_tokens": (None,), "watermarking_config": (None,)}
    _MODEL_KWARGS = ("attention_mask", "labels", "use_cache", "output_attentions", "output_hidden_states", "return_dict")

    @torch.no_grad()
    def generate(self, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None, prefix_allowed_tokens_fn=None, synced_gpus=None,
                 assistant_model=None, streamer=None, input_values=None, input_features=None, **kwargs):
        """Joint CTC / attention decoding with the reference's call contract (`generate`, ctc_encoder_plus_autoregressive_decoder.py:450-482, its processors :360-404;
        call sites: `do_generate`, src/utilities/general_utils.py:198-218 — `generation_config=..., **batch` incl. `labels`, `num_return_sequences`,
        `return_dict_in_generate`, `output_scores`, reads `.sequences` / `.sequences_scores`; `Seq2SeqTrainer.prediction_step` through `do_evaluate` :148-158 —
        `**batch, max_length=, num_beams=, output_hidden_states=True`; the configuration the trainer assigns, src/trainers/train_enc_dec_asr.py:61-85).
        Returns what transformers' generate returns for the same request: a (B * num_return_sequences, L) LongTensor padded with `pad_token_id`, or — with
        `return_dict_in_generate` — `GenerateBeamEncoderDecoderOutput(sequences, sequences_scores)` (beam search; best first per utterance) /
        `GenerateEncoderDecoderOutput(sequences)` (greedy: no sequence scores, as in transformers).  Per-step `scores` are not collected (None).
        As in the reference, the CTC processor's parameters come from the MODEL's `generation_config` (:385-396) while the passed configuration only gates it (:382).
        Every other option that would change the decoding raises instead of being ignored."""
        import copy

        from transformers.generation.utils import GenerateBeamEncoderDecoderOutput, GenerateEncoderDecoderOutput
        for name, v in (("logits_processor", logits_processor), ("stopping_criteria", stopping_criteria), ("prefix_allowed_tokens_fn", prefix_allowed_tokens_fn),
                        ("assistant_model", assistant_model), ("streamer", streamer)):
            if v is not None and not (hasattr(v, "__len__") and len(v) == 0):
                raise NotImplementedError(f"JointCTCAttentionEncoderDecoder.generate (HIP): `{name}` is not supported")
        if synced_gpus:
            raise NotImplementedError("JointCTCAttentionEncoderDecoder.generate (HIP): synced_gpus=True is not supported (every rank decodes its own batch to the end)")
        g = copy.deepcopy(generation_config if generation_config is not None else self.generation_config)
        model_kwargs = {}
        for k, v in kwargs.items():                       # transformers: generation attributes passed as keyword arguments override the configuration, the rest is for the model
            if hasattr(g, k):
                setattr(g, k, v)
            else:
                model_kwargs[k] = v
        for k in ("decoder_input_ids", "decoder_attention_mask", "encoder_outputs", "decoder_inputs_embeds", "past_key_values"):
            if model_kwargs.get(k) is not None:
                raise NotImplementedError(f"JointCTCAttentionEncoderDecoder.generate (HIP): `{k}` is not supported (decoding starts from `decoder_start_token_id`)")
        unused = [k for k, v in model_kwargs.items() if k not in self._MODEL_KWARGS and v is not None]
        if unused:                                        # transformers `_validate_model_kwargs`
            raise ValueError(f"The following `model_kwargs` are not used by the model: {unused} (note: typos in the generate arguments will also show up in this list)")
        for name, off in self._GENERATION_OFF.items():
            v = getattr(g, name, None)
            if v not in off:
                raise NotImplementedError(f"JointCTCAttentionEncoderDecoder.generate (HIP): generation option `{name}={v!r}` is not implemented "
                                          f"(implemented: num_beams, max_length / max_new_tokens, length_penalty, early_stopping, num_return_sequences, ctc_weight, "
                                          f"ctc_margin, space_token_id, apply_eos_space_trick, eos_space_trick_weight, return_dict_in_generate, output_scores)")
        inputs = self._pick_inputs(inputs, input_values, input_features)
        attention_mask = model_kwargs.get("attention_mask")
        W = int(getattr(g, "num_beams", None) or 1)
        nret = int(getattr(g, "num_return_sequences", None) or 1)
        if W == 1 and nret != 1:
            raise ValueError(f"Greedy methods without beam search do not support `num_return_sequences` different than 1 (got {nret}).")
        if nret > W:
            raise ValueError(f"`num_return_sequences` ({nret}) has to be smaller or equal to `num_beams` ({W}).")
        ret_dict = bool(getattr(g, "return_dict_in_generate", False))
        if ret_dict and (getattr(g, "output_attentions", False) or getattr(g, "output_hidden_states", False)):
            raise NotImplementedError("JointCTCAttentionEncoderDecoder.generate (HIP): attention / hidden-state outputs are not collected")
        if getattr(g, "max_new_tokens", None) is not None:
            max_length = int(g.max_new_tokens) + 1         # the decoder prompt is the start token
        else:
            max_length = int(getattr(g, "max_length", None) or 20)
        lp = getattr(g, "length_penalty", None)
        lp = 1.0 if lp is None else float(lp)
        es = getattr(g, "early_stopping", None)
        es = False if es is None else es
        eos = getattr(g, "eos_token_id", None)
        if isinstance(eos, (list, tuple)):
            if len(eos) != 1:
                raise NotImplementedError("JointCTCAttentionEncoderDecoder.generate (HIP): one eos_token_id")
            eos = eos[0]
        eos = self.config.decoder.eos_token_id if eos is None else int(eos)
        pad = getattr(g, "pad_token_id", None)
        pad = self.config.pad_token_id if pad is None else int(pad)
        start = getattr(g, "decoder_start_token_id", None)
        start = self.config.decoder_start_token_id if start is None else int(start)
        ctc = dict(ctc_weight=0.0, space_token_id=-1, apply_eos_space_trick=False, eos_space_trick_weight=1.0)
        if getattr(g, "ctc_weight", None) is not None and g.ctc_weight > 0:            # reference :382 gates on the PASSED configuration ...
            mg = self.generation_config                                                # ... and reads the processor's parameters from the MODEL's (:385-396)
            missing = [k for k in ("ctc_weight", "ctc_margin", "space_token_id", "apply_eos_space_trick", "eos_space_trick_weight") if not hasattr(mg, k)]
            if missing:
                raise AttributeError(f"model.generation_config has no {missing}: assign a GenerationConfigCustom to the model (train_enc_dec_asr.py:61-85); "
                                     f"the CTC processor reads its parameters from there (ctc_encoder_plus_autoregressive_decoder.py:385-396)")
            if int(mg.num_beams or 1) != W:
                raise ValueError(f"model.generation_config.num_beams = {mg.num_beams} but decoding runs {W} beams: the reference builds its CTC prefix scorer for the "
                                 f"model configuration's beam count (:392) and cannot decode another one (do_generate's eval_beam_factor must be 1)")
            if (mg.pad_token_id is not None and int(mg.pad_token_id) != pad) or (mg.eos_token_id is not None and int(mg.eos_token_id) != eos):
                raise ValueError("model.generation_config and the passed generation configuration disagree on pad_token_id / eos_token_id")
            ctc = dict(ctc_weight=float(mg.ctc_weight), space_token_id=int(mg.space_token_id), apply_eos_space_trick=bool(mg.apply_eos_space_trick),
                       eos_space_trick_weight=float(mg.eos_space_trick_weight))
        if not inputs.is_cuda:
            raise RuntimeError("JointCTCAttentionEncoderDecoder (HIP): inputs must be on the GPU; there is no CPU fallback")
        eng = self._get_engine(inputs.device)
        fl = attention_mask.sum(-1).to(torch.int32) if attention_mask is not None else None
        hyps = _generate(eng, inputs, fl, num_beams=W, max_length=max_length, length_penalty=lp, early_stopping=es, eos_token_id=eos, pad_token_id=pad,
                         start_token_id=start, **ctc)
        rows = [h["hypotheses"][k] for h in hyps for k in range(nret)]
        L = max(len(t) for _, t in rows)                  # (greedy: transformers runs every row until all have stopped and closed rows take pad tokens — the same layout)
        seq = torch.full((len(rows), L), pad, dtype=torch.long)
        for i, (_, t) in enumerate(rows):
            seq[i, : len(t)] = torch.tensor(t)
        seq = seq.to(inputs.device)
        if not ret_dict:
            return seq
        if W == 1:
            return GenerateEncoderDecoderOutput(sequences=seq)
        scores = torch.tensor([s for s, _ in rows], dtype=torch.float32, device=inputs.device) if getattr(g, "output_scores", False) else None
        return GenerateBeamEncoderDecoderOutput(sequences=seq, sequences_scores=scores)
