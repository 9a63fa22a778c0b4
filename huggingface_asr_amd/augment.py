"""Drop-in `SpecAug` (reference src/augmentations/spec_aug.py — the ESPnet SpecAugment the recipes name in
configs/default_data_preprocessing2d.json:36-58) that augments a whole (B, T, F) feature batch on the GPU in one HIP kernel
(`mi_specaug_f32`, csrc/specaug.hip) instead of per utterance inside dataloader workers.

Same constructor arguments and `forward(x, x_lengths=None) -> (x, x_lengths)`.  The random parameters (warp centre / target, mask
positions and widths) are drawn on the HOST with torch's CPU generator in exactly the reference's order and shapes
(time_warp :122-123, mask_along_axis :63-72), so `torch.manual_seed(k)` reproduces the reference's augmentation; only the arithmetic
(bicubic warp + masking) runs on the device.  Masked values are replaced with zero (the reference's default and the recipes' setting).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Union

import torch

from . import _lib


class SpecAug(torch.nn.Module):
    def __init__(self, apply_time_warp: bool = True, time_warp_window: int = 5, time_warp_mode: str = "bicubic", apply_freq_mask: bool = True,
                 freq_mask_width_range: Union[int, Sequence[int]] = (0, 20), num_freq_mask: int = 2, apply_time_mask: bool = True,
                 time_mask_width_range: Optional[Union[int, Sequence[int]]] = None,
                 time_mask_width_ratio_range: Optional[Union[float, Sequence[float]]] = None, num_time_mask: int = 2):
        if not apply_time_warp and not apply_time_mask and not apply_freq_mask:
            raise ValueError("Either one of time_warp, time_mask, or freq_mask should be applied")
        if apply_time_mask and (time_mask_width_range is not None) and (time_mask_width_ratio_range is not None):
            raise ValueError('Either one of "time_mask_width_range" or "time_mask_width_ratio_range" can be used')
        if apply_time_warp and time_warp_mode != "bicubic":
            raise NotImplementedError("HIP SpecAug implements the bicubic time warp of the recipes")
        super().__init__()
        self.apply_time_warp, self.window = apply_time_warp, time_warp_window
        self.apply_freq_mask, self.apply_time_mask = apply_freq_mask, apply_time_mask
        rng = lambda r: (0, r) if isinstance(r, (int, float)) else tuple(r)
        self.freq_range, self.num_freq_mask = rng(freq_mask_width_range), num_freq_mask
        self.time_range = rng(time_mask_width_range) if time_mask_width_range is not None else None
        self.time_ratio = rng(time_mask_width_ratio_range) if time_mask_width_ratio_range is not None else None
        if apply_time_mask and self.time_range is None and self.time_ratio is None:
            raise ValueError('Either one of "time_mask_width_range" or "time_mask_width_ratio_range" should be used.')
        self.num_time_mask = num_time_mask

    # --- the reference's random draws, in its order --------------------------------------------------------------
    def _warp_params(self, t: int):
        if t - self.window <= self.window:
            return 0, 0, 0
        center = int(torch.randint(self.window, t - self.window, (1,))[0])
        warped = int(torch.randint(center - self.window, center + self.window, (1,))[0]) + 1
        return 1, center, warped

    @staticmethod
    def _mask_params(B, D, lo, hi, num):
        length = torch.randint(lo, hi, (B, num))
        pos = torch.randint(0, max(1, D - int(length.max())), (B, num))
        return pos, length

    def draw_single(self, t: int, fq: int) -> torch.Tensor:
        """The parameter row of ONE utterance of t frames augmented on its own — what the reference's dataloader-side chain does (callbacks.py:100-118: SpecAug is called
        per sample with a (T, F) tensor and no lengths): the draws `forward(x[None])` makes, in the same order.  -> int32 row [len, warp on, centre, warped, masks...]."""
        nf = self.num_freq_mask if self.apply_freq_mask else 0
        nt = self.num_time_mask if self.apply_time_mask else 0
        P = torch.zeros(4 + 2 * nf + 2 * nt, dtype=torch.int32)
        P[0] = t
        if self.apply_time_warp:
            P[1:4] = torch.tensor(self._warp_params(t), dtype=torch.int32)
        if nf:
            pos, ln = self._mask_params(1, fq, self.freq_range[0], self.freq_range[1], nf)
            P[4:4 + 2 * nf:2], P[5:5 + 2 * nf:2] = pos[0].int(), ln[0].int()
        if nt:
            if self.time_range is not None:
                lo, hi = self.time_range
            else:
                lo, hi = max(0, math.floor(t * self.time_ratio[0])), min(t, math.floor(t * self.time_ratio[1]))
            if hi > lo:
                pos, ln = self._mask_params(1, t, lo, hi, nt)
                P[4 + 2 * nf::2], P[5 + 2 * nf::2] = pos[0].int(), ln[0].int()
        return P

    def apply_rows(self, x: torch.Tensor, P: torch.Tensor) -> torch.Tensor:
        """x (B, T, F) fp32 CUDA (rows beyond an utterance's length are padding), P (B, 4 + 2 nf + 2 nt) int32 parameter rows (`draw_single`): every utterance warped /
        masked over ITS OWN length, padding left zero — one launch for the batch."""
        x = x.to(torch.float32).contiguous()
        B, Tn, Fq = x.shape
        nf = self.num_freq_mask if self.apply_freq_mask else 0
        nt = self.num_time_mask if self.apply_time_mask else 0
        Pd = P.to(device=x.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(x)
        _lib.check(_lib.lib().mi_specaug_f32(x.data_ptr(), out.data_ptr(), B, Tn, Fq, Pd.data_ptr(), nf, nt, 0.0, torch.cuda.current_stream().cuda_stream), "mi_specaug_f32")
        return out

    def forward(self, x: torch.Tensor, x_lengths: Optional[torch.Tensor] = None):
        if not x.is_cuda:
            raise RuntimeError("huggingface_asr_amd.augment.SpecAug runs on the GPU (no CPU fallback); keep the reference class for CPU workers")
        squeeze = x.dim() == 2
        if squeeze:
            x = x[None]
        x = x.to(torch.float32).contiguous()
        B, Tn, Fq = x.shape
        lens = [Tn] * B if x_lengths is None else [int(v) for v in x_lengths]
        nf = self.num_freq_mask if self.apply_freq_mask else 0
        nt = self.num_time_mask if self.apply_time_mask else 0
        P = torch.zeros((B, 4 + 2 * nf + 2 * nt), dtype=torch.int32)
        P[:, 0] = torch.tensor(lens, dtype=torch.int32)
        if self.apply_time_warp:
            if x_lengths is None or all(le == lens[0] for le in lens):
                P[:, 1:4] = torch.tensor(self._warp_params(Tn), dtype=torch.int32)          # one warp for the whole padded batch (:165-167)
            else:
                for b in range(B):                                                             # per utterance, sequential draws (:170-178)
                    P[b, 1:4] = torch.tensor(self._warp_params(lens[b]), dtype=torch.int32)
        pad_to_zero = self.apply_time_warp and x_lengths is not None and not all(le == lens[0] for le in lens)
        if nf:
            pos, ln = self._mask_params(B, Fq, self.freq_range[0], self.freq_range[1], nf)
            P[:, 4:4 + 2 * nf:2], P[:, 5:5 + 2 * nf:2] = pos.int(), ln.int()
        if nt:
            if self.time_range is not None:
                lo, hi = self.time_range
            else:
                lo = max(0, math.floor(Tn * self.time_ratio[0]))
                hi = min(Tn, math.floor(Tn * self.time_ratio[1]))
            if hi > lo:
                pos, ln = self._mask_params(B, Tn, lo, hi, nt)
                P[:, 4 + 2 * nf::2], P[:, 5 + 2 * nf::2] = pos.int(), ln.int()
        if not pad_to_zero:
            P[:, 0] = Tn            # equal lengths: the reference warps / masks the whole padded tensor and leaves the padding as it is
        Pd = P.to(x.device)
        out = torch.empty_like(x)
        _lib.check(_lib.lib().mi_specaug_f32(x.data_ptr(), out.data_ptr(), B, Tn, Fq, Pd.data_ptr(), nf, nt, 0.0, torch.cuda.current_stream().cuda_stream),
                   "mi_specaug_f32")
        return (out[0] if squeeze else out), x_lengths


# ============================================================================================ speed perturbation
def _sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio functional.py `_get_sinc_resample_kernel` (sinc_interp_hann): index arithmetic in float64, table cast to float32.
    -> (kernel (new, 2*width + orig) float32 CPU tensor, width)"""
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t *= base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base / orig
    k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    return (k * window * scale).to(torch.float32).contiguous(), width


class Speed(torch.nn.Module):
    """Drop-in for `torchaudio.transforms.Speed(orig_freq, factor)` on GPU waveform batches: `forward(waveform (..., N), lengths=None)
    -> (waveform', lengths')`, one HIP kernel (`mi_speed_resample_f32`, csrc/speed.hip)."""

    def __init__(self, orig_freq: int, factor: float):
        super().__init__()
        self.orig_freq, self.factor = int(orig_freq), float(factor)
        src, tgt = int(self.factor * self.orig_freq), self.orig_freq
        g = math.gcd(src, tgt)
        self.source, self.target = src // g, tgt // g
        self._table = {}
        if self.source != self.target:
            self._kernel, self._width = _sinc_resample_kernel(self.source, self.target)

    def forward(self, waveform: torch.Tensor, lengths: Optional[torch.Tensor] = None):
        out_len = None if lengths is None else torch.ceil(lengths * self.target / self.source).to(lengths.dtype)
        if self.source == self.target:
            return waveform, out_len
        if not waveform.is_cuda:
            raise RuntimeError("huggingface_asr_amd.augment.Speed runs on the GPU (no CPU fallback); keep torchaudio's class for CPU workers")
        shape = waveform.shape
        x = waveform.reshape(-1, shape[-1]).to(torch.float32).contiguous()
        B, N = x.shape
        n_out = -(-self.target * N // self.source)
        dev = x.device
        if dev not in self._table:
            self._table[dev] = self._kernel.to(dev)
        out = torch.empty((B, n_out), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().mi_speed_resample_f32(x.data_ptr(), x.stride(0), None, B, N, self.source, self.target, self._table[dev].data_ptr(), self._width,
                                                    out.data_ptr(), out.stride(0), n_out, None, torch.cuda.current_stream().cuda_stream), "mi_speed_resample_f32")
        return out.reshape(*shape[:-1], n_out), out_len


class SpeedPerturbation(torch.nn.Module):
    """Drop-in for `torchaudio.transforms.SpeedPerturbation(orig_freq, factors)` (configs/default_data_preprocessing2d.json:3-19): one
    `torch.randint(len(factors), ())` per call picks the factor for the whole batch, exactly as torchaudio draws it."""

    def __init__(self, orig_freq: int, factors: Sequence[float]):
        super().__init__()
        self.speeders = torch.nn.ModuleList([Speed(orig_freq=orig_freq, factor=f) for f in factors])

    def forward(self, waveform: torch.Tensor, lengths: Optional[torch.Tensor] = None):
        idx = int(torch.randint(len(self.speeders), ()))
        return self.speeders[idx](waveform, lengths)
