"""Driven by tests/test_reference_route_cpu.py in a subprocess (it edits sys.modules and the HF Auto registry).

Imports the REFERENCE's own `utilities.model_utils` (third-party modules that are absent from this image — librosa, jiwer, wandb, torchaudio, … — are
stubbed: none of them is touched by model instantiation), installs the HIP classes with `huggingface_asr_amd.bind.install()`, then calls the reference's
`instantiate_aed_model` down its three branches (`model_utils.py:176-205`) and `instantiate_ctc_model` (`:117-155`) on locally saved configs / checkpoints
and prints one `OK <what>` line per check.  Nothing here runs a forward pass (CPU, no GPU)."""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from types import SimpleNamespace

REF_SRC = sys.argv[1]
WORK = sys.argv[2]
ORDER = sys.argv[3] if len(sys.argv) > 3 else "import_first"
STUBS = ("librosa", "jiwer", "wandb", "torchaudio", "evaluate", "kaldiio", "soundfile", "flashlight", "pyannote", "sclite")


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _Stub(self.__name__ + "." + name)
        setattr(self, name, v)
        return v

    def __call__(self, *a, **k):
        return self

    def __mro_entries__(self, bases):
        return (object,)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


sys.dont_write_bytecode = True
sys.path.insert(0, REF_SRC)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from huggingface_asr_amd import bind, shapes  # noqa: E402
from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig  # noqa: E402
from huggingface_asr_amd.modeling_ebranchformer import Wav2Vec2EBranchformerForCTC  # noqa: E402
from huggingface_asr_amd.modeling_joint import GPT2LMMultiHeadModel, GPT2MultiHeadConfig, JointCTCAttentionEncoderDecoder  # noqa: E402

# version skew (the reference pins transformers 4.39.3, this image has 5.x): names its glue modules import that no longer exist — harness side, like
# the shims of SURVEY.md §8c; none of them is reached by model instantiation
import transformers.generation.utils as _gu  # noqa: E402
for _n in ("BeamSearchOutput", "GreedySearchOutput", "SampleOutput", "BeamSampleOutput"):
    if not hasattr(_gu, _n):
        setattr(_gu, _n, getattr(_gu, "GenerateBeamOutput", object))

sys.meta_path.insert(0, _Finder())     # after transformers is loaded: it probes `librosa` & co. for its own optional features at import time

if ORDER == "import_first":            # what a trainer does: `from utilities.model_utils import ...` at module top, bind_all() later in main()
    import utilities.model_utils as MU
    ref_joint = MU.JointCTCAttentionEncoderDecoder
    assert ref_joint is not JointCTCAttentionEncoderDecoder
    bind.install()
else:                                  # what `python -m huggingface_asr_amd.launch` does: install() before the script is executed
    bind.install()
    import utilities.model_utils as MU
print("OK import+install", ORDER)

assert MU.JointCTCAttentionEncoderDecoder is JointCTCAttentionEncoderDecoder, MU.JointCTCAttentionEncoderDecoder
assert MU.GPT2LMMultiHeadModel is GPT2LMMultiHeadModel
assert MU.Wav2Vec2EBranchformerForCTC is Wav2Vec2EBranchformerForCTC
import utilities.bind as RB  # noqa: E402
assert RB.bind_all is bind.bind_all
RB.bind_all()                          # what the trainers call (train_enc_dec_asr.py:39)
from models.auto_wrappers import CustomModelForCausalLM  # noqa: E402
print("OK names rebound")

V = 50
tok = SimpleNamespace(pad_token_id=3, eos_token_id=2, bos_token_id=1, mask_token_id=4)
tok_len = V
tok = type("Tok", (), dict(vars(tok), __len__=lambda self: tok_len))()
enc_base = dict(shapes.TINY); enc_base.pop("num_fbanks")
enc_cfg = Wav2Vec2EBranchformerConfig(**enc_base)
dec_cfg = GPT2MultiHeadConfig(vocab_size=V, n_embd=32, n_layer=2, n_head=4, n_positions=64, head_locations=[0], head_weights=[0.6, 0.4],
                              add_cross_attention=True, is_decoder=True)
enc_dir, dec_dir = os.path.join(WORK, "enc"), os.path.join(WORK, "dec")
enc_cfg.save_pretrained(enc_dir); dec_cfg.save_pretrained(dec_dir)


def margs(**kw):
    base = dict(from_pretrained=None, from_encoder_decoder_config=False, base_encoder_model=enc_dir, base_decoder_model=dec_dir, config_overrides=None,
                ctc_weight=0.3, lsm_factor=0.1, shared_lm_head=False, decoder_pos_emb_fixed=True, average_checkpoints=False, finetune_mixing_mechanism=None)
    base.update(kw)
    return SimpleNamespace(**base)


def check_joint(model, what):
    assert type(model) is JointCTCAttentionEncoderDecoder, type(model)
    assert isinstance(model.encoder, Wav2Vec2EBranchformerForCTC) and type(model.encoder).__module__.startswith("huggingface_asr_amd"), type(model.encoder)
    assert isinstance(model.decoder, GPT2LMMultiHeadModel), type(model.decoder).__mro__
    assert model.config.ctc_weight == 0.3 and model.config.decoder.lsm_factor == 0.1 and model.config.decoder.pos_emb_fixed is True
    assert model.config.encoder.ctc_loss_reduction == "mean" and model.config.encoder.layerdrop == 0.0
    assert model.config.encoder.vocab_size == V and model.config.decoder.vocab_size == V and model.config.decoder_start_token_id == 1
    keys = set(model.state_dict())
    assert "decoder.transformer.wte.emb_layers.0.weight" in keys and "decoder.transformer.wpe.inv_freq" in keys, sorted(k for k in keys if "wte" in k or "wpe" in k)
    assert "decoder.additional_lm_heads.0.weight" in keys and "encoder.blank_projection.weight" in keys
    assert hasattr(model, "enc_to_dec_proj") == (model.config.encoder.hidden_size != model.config.decoder.hidden_size)
    dec_cls = [c for c in type(model.decoder).__mro__ if c.__name__ == "GPT2LMMultiHeadModel"][-1]
    print("OK", what, type(model).__module__ + "." + type(model).__name__, "decoder", dec_cls.__module__ + "." + dec_cls.__name__)


# branch 2 of instantiate_aed_model (model_utils.py:184-193): --from_encoder_decoder_config, what recipes_v0.0.1/librispeech_aed/*.sh use
m1 = MU.instantiate_aed_model(margs(from_encoder_decoder_config=True, config_overrides="encoder_hidden_dropout=0.05;decoder_resid_pdrop=0.2"), tok)
check_joint(m1, "instantiate_aed_model(from_encoder_decoder_config)")
assert m1.config.encoder.hidden_dropout == 0.05 and m1.config.decoder.resid_pdrop == 0.2

# branch 3 (:194-204): .from_encoder_decoder_pretrained on saved sub-models
enc_model = Wav2Vec2EBranchformerForCTC(Wav2Vec2EBranchformerConfig(**dict(enc_base, vocab_size=V)))
# (the decoder configs the recipes name carry `lsm_factor` / `pos_emb_fixed` already: from_encoder_decoder_pretrained can only override attributes that exist)
dec_model = GPT2LMMultiHeadModel(GPT2MultiHeadConfig(**dict(dec_cfg.to_dict(), pos_emb_fixed=True, lsm_factor=0.0)))
enc_model.save_pretrained(os.path.join(WORK, "enc_m")); dec_model.save_pretrained(os.path.join(WORK, "dec_m"))
m2 = MU.instantiate_aed_model(margs(base_encoder_model=os.path.join(WORK, "enc_m"), base_decoder_model=os.path.join(WORK, "dec_m")), tok)
check_joint(m2, "instantiate_aed_model(from_encoder_decoder_pretrained)")
for k, v in enc_model.state_dict().items():
    assert torch.equal(v, m2.state_dict()["encoder." + k]), k
for k, v in dec_model.state_dict().items():
    assert torch.equal(v, m2.state_dict()["decoder." + k]), k
print("OK sub-model weights carried over")

# branch 1 (:176-183): --from_pretrained of a saved joint model through AutoModelForSpeechSeq2Seq
m2.save_pretrained(os.path.join(WORK, "joint"))
m3 = MU.instantiate_aed_model(margs(from_pretrained=os.path.join(WORK, "joint")), tok)
check_joint(m3, "instantiate_aed_model(from_pretrained)")
for k, v in m2.state_dict().items():
    assert torch.equal(v, m3.state_dict()[k]), k
print("OK joint checkpoint round trip")

# instantiate_ctc_model (model_utils.py:117-155): AutoModelForCTC.from_config on the encoder config
# (its overrides are handed to config.update as STRINGS, model_utils.py:148-151; transformers 5.x validates field types, so only a string field can be overridden here)
mc = MU.instantiate_ctc_model(margs(config_overrides="position_embeddings_type=rotary"), tok, None)
assert type(mc) is Wav2Vec2EBranchformerForCTC and mc.config.vocab_size == V and mc.config.ctc_loss_reduction == "mean", type(mc)
assert mc.config.position_embeddings_type == "rotary" and "wav2vec2.encoder.embed_positions.inv_freq" in mc.state_dict()
print("OK instantiate_ctc_model", type(mc).__module__)

# the reference's own decoder registry hands out the HIP decoder as well (ctc_encoder_plus_autoregressive_decoder.py:93, bind.py:48-49)
d = CustomModelForCausalLM.from_config(m1.config.decoder)
assert isinstance(d, GPT2LMMultiHeadModel), type(d).__mro__
print("OK CustomModelForCausalLM.from_config ->", type(d).__mro__[1].__module__)

# the Whisper branch: model_utils.py:183 (`AutoModelForSpeechSeq2Seq.from_pretrained(model_path, config=config)`) on a Whisper checkpoint — what
# recipes_v0.0.1/decred/out_of_domain/decode_whisper_lumi.sh:60-66 (`--from_pretrained=openai/whisper-medium`) reaches — keeps HuggingFace's class (the trainer tests
# `isinstance(model, WhisperForConditionalGeneration)`, train_enc_dec_asr.py:82-83) and lands on the HIP encoder forward
from transformers import WhisperConfig, WhisperForConditionalGeneration  # noqa: E402
from transformers.models.whisper import modeling_whisper as MW  # noqa: E402
wcfg = WhisperConfig(d_model=128, encoder_layers=2, decoder_layers=1, encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=256, decoder_ffn_dim=256,
                     num_mel_bins=80, max_source_positions=50, max_target_positions=32, vocab_size=V, pad_token_id=3, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1)
wdir = os.path.join(WORK, "whisper")
WhisperForConditionalGeneration(wcfg).save_pretrained(wdir)
wconf = MU.AutoConfig.from_pretrained(wdir)                                   # :177, through the reference module's own names
wm = MU.AutoModelForSpeechSeq2Seq.from_pretrained(wdir, config=wconf)         # :183
assert type(wm) is WhisperForConditionalGeneration and isinstance(wm, MU.__dict__.get("WhisperForConditionalGeneration", WhisperForConditionalGeneration)), type(wm)
fwd = type(wm.model.encoder).forward
assert fwd.__module__ == "huggingface_asr_amd.whisper" and getattr(fwd, "_hfasr_hip", False), (fwd.__module__, fwd)
assert MW.WhisperEncoder._hfasr_reference_forward.__module__.startswith("transformers")
wm.eval()
os.environ["HFASR_WHISPER_STRICT"] = "1"                                       # strict: what the engine does not cover raises ...
try:
    wm.model.encoder(torch.zeros(1, 80, 100))
    raise SystemExit("HFASR_WHISPER_STRICT=1: the HIP Whisper encoder accepted CPU tensors")
except NotImplementedError:
    pass
del os.environ["HFASR_WHISPER_STRICT"]                                         # ... by default it runs transformers' own forward: a Whisper fine-tune (--do_train) keeps working
wm.train()
out = wm.model.encoder(torch.zeros(1, 80, 100))
assert tuple(out.last_hidden_state.shape) == (1, 50, 128)
print("OK whisper branch (model_utils.py:183) ->", type(wm).__module__.split(".")[0], "class,", fwd.__module__, "encoder forward")
# the reference's bind_all registers Wav2Vec2EBranchformerForPreTraining with AutoModelForPreTraining (bind.py:42); the HIP path does not build it and
# keeps the reference's own PyTorch class reachable there
from transformers import AutoModelForPreTraining  # noqa: E402
import models.encoders.e_branchformer as REB  # noqa: E402
pt = AutoModelForPreTraining._model_mapping[type(mc.config)]        # (resolved, not constructed: transformers 5.15's Wav2Vec2ForPreTraining reads config fields 4.39's conformer config had)
assert pt is REB.Wav2Vec2EBranchformerForPreTraining and pt.__module__ == "models.encoders.e_branchformer", pt
print("OK AutoModelForPreTraining(wav2vec2-ebranchformer) ->", pt.__module__)
print("ALL OK")
