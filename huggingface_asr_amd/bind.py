"""`bind_all()` — the plug-in boundary (reference src/utilities/bind.py:36-58).

Registers OUR classes under the reference's model types with HuggingFace's Auto* registry, so that the reference's
trainers (`src/trainers/train_ctc_asr.py:30`, `train_enc_dec_asr.py:39` call `bind_all()` first and then use only
`Auto*`) consume the HIP implementation unchanged.  `install()` monkey-patches `utilities.bind.bind_all` when the
reference tree is importable."""
from transformers import AutoConfig, AutoFeatureExtractor, AutoModelForCTC, AutoModelForPreTraining, AutoModelForSpeechSeq2Seq

from .configuration_ebranchformer import Wav2Vec2EBranchformerConfig
from .feature_extraction import CustomFeatureExtractor, CustomFeatureExtractorConfig
from .modeling_bestrq import BestRQEBranchformerForCTC, BestRQEBranchformerForPreTraining, BestRQEBranchformerForPreTrainingConfig
from .modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
from .modeling_joint import GPT2MultiHeadConfig, JointCTCAttentionEncoderDecoder, JointCTCAttentionEncoderDecoderConfig


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


def install():
    """Swap the reference's registration for ours (call before the reference trainer's own bind_all())."""
    import utilities.bind as ref_bind  # reference module, only present when its src/ is on sys.path
    ref_bind.bind_all = bind_all
    bind_all()
