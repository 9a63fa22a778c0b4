"""Device-side form of the reference's per-split training transform chain.

The reference describes its pre-processing as a list of steps per split (configs/default_data_preprocessing2d.json:3-58 —
`torchaudio.transforms.SpeedPerturbation` -> `feature_extractor` -> `augmentations.spec_aug.SpecAug`, each with `steps_before_activation`) and runs it PER
UTTERANCE inside dataloader workers: `DataPreprocessingManagerCallback.default_transform` (src/utilities/callbacks.py:108-118) strips leading / trailing zero
samples, zero-pads to >= 8000 samples and hands the array to `transformer` (:100-106), which calls the steps in order through `DelayedStartWrapper`
(:52-66: a step is the identity until the trainer's global step reaches its `steps_before_activation`, then stays on).

`DevicePreprocessing` applies the same chain to a whole (B, N) waveform batch that is already on the GPU:
  strip / pad (mi_trim_zeros_pad_f32) -> speed perturbation with ONE FACTOR PER UTTERANCE (mi_speed_resample_f32, one launch per factor group)
  -> Kaldi log-mel + CMVN (mi_fbank_f64 / cmvn) -> SpecAug with per-utterance parameters (mi_specaug_f32, one launch) -> padded (B, T, 80) + attention mask
(what `SpeechCollatorWithPadding`, collators.py:65-106, would have built from the per-utterance results).
The random parameters are drawn on the host with torch's CPU generator in the order a single dataloader worker draws them — for utterance 0 the speed factor, then its
SpecAug parameters, then utterance 1, ... — so `torch.manual_seed(k)` reproduces what the reference's chain does to the same batch in the same order.
Speed perturbation stays PARITY-UNPINNED (torchaudio is neither under /root/reference nor in this image: DESIGN.md 4b); the SpecAug leg is pinned to the
reference module through tests/golden/specaug.npz.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .augment import SpecAug, SpeedPerturbation


class _Step:
    def __init__(self, kind: str, module, start_at: int):
        self.kind, self.module, self.start_at, self.active = kind, module, int(start_at), False

    def new_step(self, step: int):            # DelayedStartWrapper.new_step (callbacks.py:58-61): switches on once, never off
        if step >= self.start_at:
            self.active = True


class DevicePreprocessing:
    def __init__(self, preprocessing_config: Dict[str, List[Dict]], feature_extractor, pad_to_multiple_of: Optional[int] = None, min_audio_length: int = 8000):
        """preprocessing_config: the reference's JSON (e.g. configs/default_data_preprocessing2d.json) as a dict; feature_extractor: huggingface_asr_amd's
        CustomFeatureExtractor (its `extract_on_device` is the `feature_extractor` step); pad_to_multiple_of: the collator's (recipes: 100)."""
        self.fe, self.pad_to, self.min_len = feature_extractor, pad_to_multiple_of, int(min_audio_length)
        self.chains: Dict[str, List[_Step]] = {}
        for split, steps in preprocessing_config.items():
            chain = []
            for cfg in steps:
                name, start = cfg["name"], cfg.get("steps_before_activation", 0)
                if name == "feature_extractor":
                    chain.append(_Step("fe", None, start))
                elif name.endswith("SpeedPerturbation"):
                    chain.append(_Step("speed", SpeedPerturbation(**cfg["params"]), start))
                elif name.endswith("SpecAug"):
                    chain.append(_Step("specaug", SpecAug(**cfg["params"]), start))
                else:
                    raise NotImplementedError(f"pre-processing step {name!r} has no device form (device forms: SpeedPerturbation, feature_extractor, SpecAug)")
            kinds = [s.kind for s in chain]
            if kinds.count("fe") != 1 or any(k == "speed" for k in kinds[kinds.index("fe"):]) or any(k == "specaug" for k in kinds[: kinds.index("fe")]):
                raise NotImplementedError(f"split {split!r}: the device chain is [SpeedPerturbation] -> feature_extractor -> [SpecAug] (got {kinds})")
            self.chains[split] = chain
        self.new_step(0)                        # on_init_end -> propagate_state_to_transforms (callbacks.py:125-133)

    def new_step(self, step: int):
        """DataPreprocessingManagerCallback.propagate_state_to_transforms (callbacks.py:120-123, called at every step begin and on resume)"""
        for chain in self.chains.values():
            for s in chain:
                s.new_step(int(step))

    @torch.no_grad()
    def __call__(self, waveforms: torch.Tensor, num_samples: Optional[torch.Tensor] = None, split: str = "train"):
        """waveforms (B, N) float32 CUDA, num_samples (B) valid samples per row (None: N) -> dict(input_features (B, T, F) fp32, attention_mask (B, T) int32,
        num_frames (B) int32, speed_factor_index (B) list | None)."""
        from .fbank import strip_zeros_pad_gpu
        if not waveforms.is_cuda:
            raise RuntimeError("DevicePreprocessing runs on the GPU (no CPU fallback); the reference's dataloader-side chain is the CPU form")
        chain = self.chains[split if split in self.chains else "default_preprocessing"]      # callbacks.py:127
        fe_step = next(s for s in chain if s.kind == "fe")
        if not fe_step.active:
            raise NotImplementedError("a delayed feature_extractor step would hand raw audio to the model; every reference config starts it at step 0")
        speed = next((s for s in chain if s.kind == "speed" and s.active), None)
        spec = next((s for s in chain if s.kind == "specaug" and s.active), None)
        wav, eff = strip_zeros_pad_gpu(waveforms, num_samples, self.min_len)                 # default_transform's array handling, on the device
        B = wav.shape[0]
        lens = [int(v) for v in eff.cpu()]                                                    # one host sync: the draws below need every utterance's length
        # ---- host draws, utterance by utterance, in the order one dataloader worker makes them
        idx, plist, out_lens = [], [], []
        for b in range(B):
            n = lens[b]
            if speed is not None:
                k = int(torch.randint(len(speed.module.speeders), ()))                       # torchaudio SpeedPerturbation.forward: one randint per call
                sp = speed.module.speeders[k]
                n = -(-n * sp.target // sp.source)                                            # ceil(n * target / source)
                idx.append(k)
            out_lens.append(n)
            if spec is not None:
                t_b = 1 + (n - 400) // 160                                                    # snip-edges frame count of the Kaldi fbank (25 ms / 10 ms at 16 kHz)
                plist.append(spec.module.draw_single(t_b, self.fe.feature_size))
        # ---- device: speed perturbation, one launch per factor group
        if speed is not None:
            n_max = max(out_lens)
            new = torch.zeros((B, n_max), dtype=torch.float32, device=wav.device)
            for k, sp in enumerate(speed.module.speeders):
                rows = [b for b in range(B) if idx[b] == k]
                if not rows:
                    continue
                sel = torch.tensor(rows, device=wav.device)
                y, _ = sp(wav.index_select(0, sel))
                w = min(y.shape[1], n_max)
                new[sel, :w] = y[:, :w]
            wav = new
            eff = torch.tensor(out_lens, dtype=torch.int32, device=wav.device)
        # ---- device: log-mel + CMVN on the batch (the clips already went through default_transform), trimmed to the longest clip like the collator
        feats, mask = self.fe.extract_on_device(wav, eff, pad_to_multiple_of=self.pad_to, default_transform=False)
        frames = mask.sum(-1).to(torch.int32)
        if spec is not None:
            P = torch.stack(plist, 0)
            feats = spec.module.apply_rows(feats, P)
        return dict(input_features=feats, attention_mask=mask, num_frames=frames, speed_factor_index=idx if speed is not None else None)
