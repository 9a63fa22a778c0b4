"""Drop-in BEST-RQ classes (reference src/models/bestrq.py; registered by bind.py:44-46 of the reference under "bestrq-ebranchformer"):
`BestRQEBranchformerForPreTrainingConfig`, `BestRQEBranchformerForPreTraining` (AutoModelForPreTraining) and
`BestRQEBranchformerForCTC` (AutoModelForCTC fine-tuning head).  Same constructor arguments, parameter / buffer names
(`classifiers.k.{weight,bias}`, `rpq.P`, `rpq.CB`, `wav2vec2.*` without `masked_spec_embed`) and output dataclass; forward and the
training step run on the HIP trainer (train_bestrq.BestRQTrainer), gradients reach autograd through autograd_bridge.HipStep."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn
from transformers import PreTrainedModel
from transformers.models.wav2vec2.modeling_wav2vec2 import Wav2Vec2ForPreTrainingOutput

from .configuration_ebranchformer import Wav2Vec2EBranchformerConfig
from .engine import cfg_from_hf
from .modeling_ebranchformer import Wav2Vec2EBranchformerForCTC, _dropout_seed, _Holder, _Wav2Vec2


class BestRQEBranchformerForPreTrainingConfig(Wav2Vec2EBranchformerConfig):
    """bestrq.py:30-41,154-172: the E-Branchformer config + quantizer sizes + fine-tuning switches"""
    model_type = "bestrq-ebranchformer"

    def __init__(self, best_rq_codebook_size=8192, best_rq_codebook_dim=16, best_rq_num_books=1, best_rq_in_dim=320,
                 finetune_with_additional_layer=False, finetune_with_layer_mixing=False, freeze_norm_for_finetunning=False, **kwargs):
        super().__init__(**kwargs)
        self.best_rq_codebook_size = best_rq_codebook_size
        self.best_rq_codebook_dim = best_rq_codebook_dim
        self.best_rq_num_books = best_rq_num_books
        self.best_rq_in_dim = best_rq_in_dim
        self.finetune_with_additional_layer = finetune_with_additional_layer
        self.finetune_with_layer_mixing = finetune_with_layer_mixing
        self.freeze_norm_for_finetunning = freeze_norm_for_finetunning


class _RPQ(_Holder):
    """RandomProjectionQuantizer buffers (bestrq.py:66-78): xavier-uniform projection, codebook = F.normalize(randn) (default dim, as there)"""

    def __init__(self, cfg):
        super().__init__()
        nb, ind, cd, C = cfg.best_rq_num_books, cfg.best_rq_in_dim, cfg.best_rq_codebook_dim, cfg.best_rq_codebook_size
        amp = math.sqrt(3.0) * math.sqrt(2.0 / float(ind * cd + nb * cd))          # fan_in = in_dim*cd, fan_out = books*cd for a (books, in_dim, cd) tensor
        self.register_buffer("P", torch.empty(nb, ind, cd).uniform_(-amp, amp))
        self.register_buffer("CB", F.normalize(torch.randn(nb, C, cd)))


def _bestrq_cfg(config) -> dict:
    c = cfg_from_hf(config)
    c.update(best_rq_codebook_size=config.best_rq_codebook_size, best_rq_codebook_dim=config.best_rq_codebook_dim,
             best_rq_num_books=config.best_rq_num_books, best_rq_in_dim=config.best_rq_in_dim)
    return c


class BestRQEBranchformerForPreTraining(PreTrainedModel):
    config_class = BestRQEBranchformerForPreTrainingConfig
    base_model_prefix = "wav2vec2"
    main_input_name = "input_values"

    def __init__(self, config: BestRQEBranchformerForPreTrainingConfig):
        super().__init__(config)
        self.rpq = _RPQ(config)
        self.classifiers = nn.ModuleList(nn.Linear(config.hidden_size, config.best_rq_codebook_size) for _ in range(config.best_rq_num_books))
        self.wav2vec2 = _Wav2Vec2(config)
        if hasattr(self.wav2vec2, "masked_spec_embed"):
            del self.wav2vec2.masked_spec_embed                       # bestrq.py:174-176
        self._trainer = None
        from .autograd_bridge import detach_state_dict_views
        self._register_state_dict_hook(detach_state_dict_views)
        self.post_init()

    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            nn.init.normal_(module.weight, mean=0.0, std=std)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            nn.init.ones_(module.weight); nn.init.zeros_(module.bias)

    def _get_trainer(self, device):
        from .train_bestrq import BestRQTrainer
        if self._trainer is None or self._trainer.device != torch.device(device):
            self._trainer = BestRQTrainer(_bestrq_cfg(self.config), device, dp_sync=False, seed=_dropout_seed())
            self._trainer_key = None
        return self._trainer

    def forward(self, input_values: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor] = None,
                mask_time_indices: Optional[torch.BoolTensor] = None, output_attentions=None, output_hidden_states=None, return_dict=None, **kw):
        if mask_time_indices is None:
            raise ValueError("BEST-RQ needs `mask_time_indices` (the reference reads its shape to stack the input frames, bestrq.py:127)")
        if output_attentions:
            raise NotImplementedError("attention probabilities are never materialised by the fused HIP attention kernel")
        if not input_values.is_cuda:
            raise RuntimeError("BestRQEBranchformerForPreTraining (HIP): inputs must be on the GPU; there is no CPU fallback")
        tr = self._get_trainer(input_values.device)
        fl = attention_mask.sum(-1).to(torch.int32) if attention_mask is not None else None
        training = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if training:
            from .autograd_bridge import run_training_forward

            def step(t):
                t.enc.store.zero_grad(); t.store.zero_grad()
                return t.forward_backward(input_values, fl, mask_time_indices)
            loss, out = run_training_forward(self, tr, step)
        else:
            from .autograd_bridge import bridge_generation
            key = (sum(p._version for p in self.parameters()), bridge_generation(self), tuple(p.data_ptr() for p in self.parameters()))
            if getattr(self, "_trainer_key", None) != key:
                tr.load_state_dict(dict(self.state_dict()))
                self._trainer_key = key
            out = tr.forward_backward(input_values, fl, mask_time_indices, backward=False)
            loss = out["loss"]
        hidden = out["last_hidden"]
        if return_dict is False:
            return (loss, hidden, None, None)
        return Wav2Vec2ForPreTrainingOutput(loss=loss, projected_states=hidden, codevector_perplexity=None, hidden_states=None, attentions=None,
                                            contrastive_loss=None, diversity_loss=None)


class BestRQEBranchformerForCTC(Wav2Vec2EBranchformerForCTC):
    """CTC fine-tuning head on a BEST-RQ pre-trained encoder (bestrq.py:192-322), incl. the recipes' options
    (`finetune_with_layer_mixing`: softmax(per_layer_weights)-weighted sum of all hidden states, :239-245; `finetune_with_additional_layer`: one
    more E-Branchformer layer `additional_layer.*` before the head, :247-274).  Both run inside the single-call forward engine (eval) and the
    HIP trainer (train() + loss.backward()); this class only adds the parameters."""
    config_class = BestRQEBranchformerForPreTrainingConfig

    def __init__(self, config):
        super().__init__(config)
        extra = bool(getattr(config, "finetune_with_additional_layer", False))
        mix = bool(getattr(config, "finetune_with_layer_mixing", False))
        if extra:
            from .modeling_ebranchformer import _Layer
            self.additional_layer = _Layer(config)
        if mix:
            w = torch.zeros(config.num_hidden_layers + 1)
            w[-1] = 1.0                                            # bestrq.py:203-205
            self.per_layer_weights = nn.Parameter(w)
        if extra or mix:
            self.post_init()
