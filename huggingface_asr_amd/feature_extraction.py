"""Drop-in `CustomFeatureExtractor` (reference src/utilities/feature_extractors.py:14-61).

Same constructor/`__call__`/`pad` contract (it subclasses transformers' Speech2TextFeatureExtractor exactly as the
reference does), but `_extract_fbank_features` is a vectorised float64 numpy restatement of the per-frame Python loop
(fork-safe and GPU-free, because the reference runs the extractor inside dataloader workers), and
`extract_on_device` exposes the HIP log-mel + CMVN kernels as the GPU pre-stage of the model's forward."""
from __future__ import annotations

from typing import Union

import numpy as np
from transformers import BatchFeature, PretrainedConfig, Speech2TextFeatureExtractor

from .fbank import FbankTables, fbank_numpy


class CustomFeatureExtractorConfig(PretrainedConfig):
    model_type = "custom_feature_extractor"


class CustomFeatureExtractor(Speech2TextFeatureExtractor):
    def __init__(self, feature_size=80, norm_type="utterance", do_ceptral_normalize=True, update_norms=True,
                 global_means=None, global_stds=None, *args, **kwargs):
        if norm_type not in ["utterance", "global"]:
            raise ValueError(f"norm_type should be either 'utterance' or 'global'. Got {norm_type}")
        super().__init__(feature_size=feature_size, do_ceptral_normalize=do_ceptral_normalize and norm_type == "utterance",
                         *args, **kwargs)
        self.norm_type = norm_type
        self.update_norms = update_norms
        if self.norm_type == "global":
            import torch
            self.global_means = np.array(torch.load(global_means).tolist() if isinstance(global_means, str) else global_means)
            self.global_stds = np.array(torch.load(global_stds).tolist() if isinstance(global_stds, str) else global_stds)
        self._tables = None

    def _get_tables(self) -> FbankTables:
        if getattr(self, "_tables", None) is None:
            self._tables = FbankTables(self.num_mel_bins, self.sampling_rate)
        return self._tables

    def to_dict(self):
        d = super().to_dict()
        d.pop("_tables", None)
        for k in ("global_means", "global_stds"):
            if isinstance(d.get(k), np.ndarray):
                d[k] = d[k].tolist()
        return d

    def _extract_fbank_features(self, waveform: np.ndarray) -> np.ndarray:
        if self.dither != 0.0:
            raise NotImplementedError("dither != 0 is not supported by the vectorised extractor")
        return fbank_numpy(np.squeeze(waveform), self._get_tables())

    def global_normalize(self, input_features):
        return (input_features - self.global_means) / self.global_stds

    def __call__(self, *args, **kwargs) -> BatchFeature:
        batch = super().__call__(*args, **kwargs)
        if self.norm_type == "global":
            batch["input_features"] = self.global_normalize(batch.get("input_features"))
        return batch

    # ---- GPU pre-stage (not part of the reference surface)
    def extract_on_device(self, waveforms, num_samples=None, pad_to_multiple_of=None, default_transform=True, min_audio_length=8000, trim_to_longest=True):
        """waveforms (B,N) float32 CUDA tensor -> (input_features (B,T,80) fp32, attention_mask (B,T) int32): the batch the reference's dataloader side
        hands to the model.  `default_transform=True` applies what `DataPreprocessingManagerCallback.default_transform` does to every clip before the extractor
        (callbacks.py:108-118: strip leading / trailing zero samples, zero-pad to >= `min_audio_length` samples) on the device; `pad_to_multiple_of` is the
        collator's (collators.py:82-88), and like the collator the time axis ends at the longest clip of the batch (`trim_to_longest`, one host sync; pass False
        in a throughput loop and keep the buffer's own length).  Pass default_transform=False for clips that already went through it."""
        import torch

        from .fbank import fbank_gpu, strip_zeros_pad_gpu
        if default_transform:
            waveforms, num_samples = strip_zeros_pad_gpu(waveforms, num_samples, min_audio_length)
        kw = {}
        if self.norm_type == "global":
            kw = dict(normalize="global", global_means=torch.as_tensor(self.global_means, dtype=torch.float32, device=waveforms.device),
                      global_stds=torch.as_tensor(self.global_stds, dtype=torch.float32, device=waveforms.device))
        elif not self.do_ceptral_normalize:
            kw = dict(normalize=None)
        feats, frames = fbank_gpu(waveforms, self._get_tables(), num_samples, pad_to_multiple_of, normalize_means=self.normalize_means,
                                  normalize_vars=self.normalize_vars, padding_value=self.padding_value, **kw)
        if trim_to_longest and num_samples is not None:
            # the collator pads to the LONGEST clip of the batch (then up to the multiple); the device buffer may be longer.  One host sync for the length.
            t_max = int(frames.max().item())
            if pad_to_multiple_of:
                t_max = (t_max + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
            feats = feats[:, : max(t_max, 1)].contiguous() if t_max < feats.shape[1] else feats
        mask = (torch.arange(feats.shape[1], device=feats.device)[None, :] < frames[:, None]).to(torch.int32)
        return feats, mask
