"""Configuration mirror of the reference's `Wav2Vec2EBranchformerConfig`
(reference src/models/encoders/e_branchformer.py:37-61 + CustomFEConfig src/models/extractors.py:13-20).
Same model_type, same field names and defaults, so checkpoints/configs written by either side load in the other."""
from transformers.models.wav2vec2_conformer.configuration_wav2vec2_conformer import Wav2Vec2ConformerConfig


class Wav2Vec2EBranchformerConfig(Wav2Vec2ConformerConfig):
    model_type = "wav2vec2-ebranchformer"

    def __init__(
        self,
        ebranchformer_conv_dropout=0.1,
        csgu_activation="identity",
        csgu_kernel_size=31,
        csgu_use_linear_after_conv=False,
        merge_conv_kernel=31,
        use_macaron_ff=True,
        is_causal=False,
        conv_padding=(1, 1),
        num_fbanks=80,
        context_awareness_type=None,
        **kwargs,
    ):
        super().__init__(**kwargs)
        self.csgu_kernel_size = csgu_kernel_size
        self.csgu_activation = csgu_activation
        self.csgu_conv_dropout = ebranchformer_conv_dropout
        self.csgu_use_linear_after_conv = csgu_use_linear_after_conv
        self.merge_conv_kernel = merge_conv_kernel
        self.use_macaron_ff = use_macaron_ff
        self.is_causal = is_causal
        self.conv_padding = list(conv_padding)
        self.num_fbanks = num_fbanks
        self.context_awareness_type = context_awareness_type
