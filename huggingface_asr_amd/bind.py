"""`bind_all()` / `install()` — the plug-in boundary (reference src/utilities/bind.py:36-58).

`bind_all()` registers OUR classes under the reference's model types with HuggingFace's Auto* registry; the reference's trainers
(`src/trainers/train_ctc_asr.py:30`, `train_enc_dec_asr.py:39`) call `bind_all()` first and then build models through `Auto*`.

The AED recipes do NOT only go through the registry, though: `src/utilities/model_utils.py:30-33` imports
`JointCTCAttentionEncoderDecoder{,Config}` BY NAME and constructs them directly (`:193` for `--from_encoder_decoder_config`, `:199` through
`.from_encoder_decoder_pretrained`), and the reference joint class builds its decoder with its own `CustomModelForCausalLM` registry
(`src/models/ctc_encoder_plus_autoregressive_decoder.py:93`, registered at `bind.py:48-49`).  `install()` therefore also
  * rebinds those names — in the defining modules (for importers that come later) and in every already-imported module of the reference tree
    that holds them (`utilities.model_utils`, `utilities.bind`, a trainer running as `__main__`, …) — to the HIP classes, and
  * registers the HIP decoder with the reference's `CustomModelForCausalLM`,
so the reference's scripts run unchanged: `python -m huggingface_asr_amd.launch src/trainers/train_enc_dec_asr.py <recipe flags>`.

The Whisper branch (`model_utils.py:183` on a Whisper checkpoint, `decode_whisper_lumi.sh:60-66`) keeps HuggingFace's `WhisperForConditionalGeneration` — the trainer tests
for that class (`train_enc_dec_asr.py:82-83`) — and gets the HIP encoder through `whisper.install_whisper()` (called by `bind_all()`): `WhisperEncoder.forward` is replaced,
decoder / `generate` / checkpoints stay transformers' own."""
import sys

from transformers import AutoConfig, AutoFeatureExtractor, AutoModelForCTC, AutoModelForPreTraining, AutoModelForSpeechSeq2Seq

from .configuration_ebranchformer import Wav2Vec2EBranchformerConfig
from .feature_extraction import CustomFeatureExtractor, CustomFeatureExtractorConfig
from .modeling_bestrq import BestRQEBranchformerForCTC, BestRQEBranchformerForPreTraining, BestRQEBranchformerForPreTrainingConfig
from .modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
from .modeling_joint import GPT2LMMultiHeadModel, GPT2MultiHeadConfig, JointCTCAttentionEncoderDecoder, JointCTCAttentionEncoderDecoderConfig, Seq2SeqLMOutputLosses

# reference module -> {name it exports: our class}.  (Classes the HIP path does not build — Wav2Vec2EBranchformerForPreTraining, the mixing / residual
# GPT-2 variants — keep their reference definitions)
REBIND = {
    "models.ctc_encoder_plus_autoregressive_decoder": {"JointCTCAttentionEncoderDecoder": JointCTCAttentionEncoderDecoder,
                                                      "JointCTCAttentionEncoderDecoderConfig": JointCTCAttentionEncoderDecoderConfig,
                                                      "Seq2SeqLMOutputLosses": Seq2SeqLMOutputLosses},
    "models.encoders.e_branchformer": {"Wav2Vec2EBranchformerConfig": Wav2Vec2EBranchformerConfig, "Wav2Vec2EBranchformerForCTC": Wav2Vec2EBranchformerForCTC},
    "models.decoders.multi_head_gpt2": {"GPT2MultiHeadConfig": GPT2MultiHeadConfig, "GPT2LMMultiHeadModel": GPT2LMMultiHeadModel},
    "models.bestrq": {"BestRQEBranchformerForCTC": BestRQEBranchformerForCTC, "BestRQEBranchformerForPreTraining": BestRQEBranchformerForPreTraining,
                      "BestRQEBranchformerForPreTrainingConfig": BestRQEBranchformerForPreTrainingConfig},
    "utilities.feature_extractors": {"CustomFeatureExtractor": CustomFeatureExtractor, "CustomFeatureExtractorConfig": CustomFeatureExtractorConfig},
}
_REF_PACKAGES = ("models", "utilities", "decoding", "trainers", "augmentations")


def bind_all():
    AutoConfig.register("wav2vec2-ebranchformer", Wav2Vec2EBranchformerConfig, exist_ok=True)
    AutoModelForCTC.register(Wav2Vec2EBranchformerConfig, Wav2Vec2EBranchformerForCTC, exist_ok=True)
    AutoConfig.register("bestrq-ebranchformer", BestRQEBranchformerForPreTrainingConfig, exist_ok=True)
    AutoModelForCTC.register(BestRQEBranchformerForPreTrainingConfig, BestRQEBranchformerForCTC, exist_ok=True)
    AutoModelForPreTraining.register(BestRQEBranchformerForPreTrainingConfig, BestRQEBranchformerForPreTraining, exist_ok=True)
    AutoConfig.register("gpt2-multi-head", GPT2MultiHeadConfig, exist_ok=True)
    AutoConfig.register("joint_aed_ctc_speech-encoder-decoder", JointCTCAttentionEncoderDecoderConfig, exist_ok=True)
    AutoModelForSpeechSeq2Seq.register(JointCTCAttentionEncoderDecoderConfig, JointCTCAttentionEncoderDecoder, exist_ok=True)
    AutoConfig.register("custom_feature_extractor", CustomFeatureExtractorConfig, exist_ok=True)
    AutoFeatureExtractor.register(CustomFeatureExtractorConfig, CustomFeatureExtractor, exist_ok=True)
    from .whisper import install_whisper                          # the Whisper branch (model_utils.py:183 on a Whisper checkpoint): HF's classes, our encoder forward
    install_whisper()
    ref_enc = sys.modules.get("models.encoders.e_branchformer")  # wav2vec2-style contrastive pre-training is not built on the HIP path: when the reference's
    ref_pt = getattr(ref_enc, "Wav2Vec2EBranchformerForPreTraining", None) if ref_enc is not None else None      # tree is loaded its own PyTorch class stays reachable
    if isinstance(ref_pt, type):                                 # through AutoModelForPreTraining, as its bind_all registers it (reference bind.py:42)
        ref_pt.config_class = Wav2Vec2EBranchformerConfig
        AutoModelForPreTraining.register(Wav2Vec2EBranchformerConfig, ref_pt, exist_ok=True)
    ref_auto = sys.modules.get("models.auto_wrappers")          # the reference's own decoder registry (bind.py:48-49), when its tree is loaded
    if ref_auto is not None:
        ref_auto.CustomModelForCausalLM.register(GPT2MultiHeadConfig, GPT2LMMultiHeadModel, exist_ok=True)


def _rebind_everywhere(replaced: dict):
    """every already-imported reference module that holds one of the replaced class objects under any name gets ours instead"""
    for name, mod in list(sys.modules.items()):
        if mod is None or not (name == "__main__" or name.split(".")[0] in _REF_PACKAGES):
            continue
        for attr, val in list(vars(mod).items()):
            if isinstance(val, type) and id(val) in replaced:
                setattr(mod, attr, replaced[id(val)])


def install():
    """Swap the reference's classes for the HIP ones (needs the reference's `src/` on sys.path; call before the trainer's main(), e.g.
    through `python -m huggingface_asr_amd.launch`).  Idempotent."""
    import importlib
    replaced = {}
    for modname, names in REBIND.items():
        try:
            mod = importlib.import_module(modname)
        except ImportError:          # a reference module whose third-party imports are absent on this box: nothing of it can be in use either
            continue
        for attr, ours in names.items():
            ref = getattr(mod, attr, None)
            if isinstance(ref, type) and ref is not ours:
                replaced[id(ref)] = ours
            setattr(mod, attr, ours)
    _rebind_everywhere(replaced)
    importlib.import_module("models.auto_wrappers")
    ref_bind = importlib.import_module("utilities.bind")
    ref_bind.bind_all = bind_all
    _rebind_everywhere({})          # no-op for classes; kept for symmetry when called twice
    for name, mod in list(sys.modules.items()):      # trainers do `from utilities.bind import bind_all`
        if mod is not None and (name == "__main__" or name.split(".")[0] in _REF_PACKAGES) and getattr(mod, "bind_all", None) is not None \
                and getattr(mod, "bind_all") is not bind_all and getattr(getattr(mod, "bind_all"), "__module__", "") == "utilities.bind":
            mod.bind_all = bind_all
    bind_all()
