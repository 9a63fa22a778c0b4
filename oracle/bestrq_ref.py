"""ORACLE (test infrastructure, never shipped): CPU restatement of BEST-RQ pre-training.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates reference `src/models/bestrq.py`: RandomProjectionQuantizer.forward (:66-82), BestRQMask._mask_hidden_states (:84-97, with the
noise passed in instead of drawn from torch's RNG) and BestRQModel.forward (:107-152): cross-entropy with reduction="sum" over the masked
frames, divided by the number of codebooks.  The encoder is oracle/ebranchformer_ref.py with the masking hook at the reference's
`_mask_hidden_states` call site (tf wav2vec2_conformer :1164).  Pinned by tests/golden/bestrq_tiny.npz (made from the imported reference).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ebranchformer_ref as E


def rpq_targets(x: torch.Tensor, P: torch.Tensor, CB: torch.Tensor) -> torch.Tensor:
    """x (B, T', in_dim), P (books, in_dim, cd), CB (books, C, cd) -> (B, books, T') int64   (:80-82)"""
    h = F.normalize(x[:, None, ...] @ P)                                                    # (B, books, T', cd)
    return torch.linalg.vector_norm(CB[None, :, :, None, :] - h[:, :, None, :, :], dim=-1).argmin(dim=2)


def forward(sd: dict, cfg: dict, feats: torch.Tensor, attention_mask, mask_time_indices: torch.Tensor, noise: torch.Tensor, q=None):
    """-> dict(loss, last_hidden, targets (B, books, T'), logits (B, books, T', C)).  noise (B, T', d): values written at the masked frames."""
    B, T2 = mask_time_indices.shape
    targets = rpq_targets(feats.reshape(B, T2, -1), sd["rpq.P"], sd["rpq.CB"]).masked_fill(~mask_time_indices[:, None, :], -100)
    esd = {k: v for k, v in sd.items() if k.startswith("wav2vec2.")}

    def dm(x, layer, site):                     # reuse the dropout hook slot of the encoder input chain: site 0 runs right after the projection
        if layer == cfg["num_hidden_layers"] and site == 0:
            return torch.where(mask_time_indices[..., None], noise, x)
        return x
    hidden = E.encoder_forward(esd, cfg, feats, attention_mask, q, dm=dm)
    nb = sd["rpq.P"].shape[0]
    logits = torch.stack([F.linear(hidden, sd[f"classifiers.{k}.weight"], sd[f"classifiers.{k}.bias"]) for k in range(nb)], dim=1)
    loss = F.cross_entropy(logits.flatten(0, 1).transpose(1, 2), targets.flatten(0, 1), reduction="sum") / nb
    return dict(loss=loss, last_hidden=hidden, targets=targets, logits=logits)
