"""Drop-in HF logits processors for joint CTC/attention decoding (reference src/decoding/ctc_scorer.py:259-365).

`CTCRescorerLogitsProcessor` has the reference's constructor and `__call__(input_ids, scores)` contract, so
`JointCTCAttentionEncoderDecoder._get_logits_processor` (ctc_encoder_plus_autoregressive_decoder.py:360-404) or any
HF `generate()` call can append it unchanged; the prefix scores come from the HIP kernels in csrc/ctc_prefix.hip.
Reference quirk reproduced: the state is re-selected with token ids only, i.e. every beam continues from beam 0's
forward variables (:327-329 vs :191)."""
from __future__ import annotations

import torch
from transformers import GenerationConfig, LogitsProcessor

from . import _lib

LOGZERO = -10000000000.0


class GenerationConfigCustom(GenerationConfig):
    """reference src/decoding/config.py:4-23 — HF GenerationConfig + the joint-decoding knobs."""

    def __init__(self, ctc_weight=0.0, ctc_margin=0, lm_weight=0, lm_model=None, space_token_id=-1, eos_space_trick_weight=0,
                 apply_eos_space_trick=False, **kwargs):
        super().__init__(**kwargs)
        self.ctc_weight = ctc_weight
        self.ctc_margin = ctc_margin
        self.lm_weight = lm_weight
        self.lm_model = lm_model
        self.space_token_id = space_token_id
        self.eos_space_trick_weight = eos_space_trick_weight
        self.apply_eos_space_trick = apply_eos_space_trick


class CTCRescorerLogitsProcessor(LogitsProcessor):
    FULL_STATE_BYTES = 1 << 30      # keep every (hypothesis, token) chain between calls while that tensor stays below this (else: re-run the selected chains)

    def __init__(self, encoder_logits: torch.FloatTensor, encoder_output_lens: torch.LongTensor, pad_token_id: int,
                 eos_token_id: int, ctc_margin: int, ctc_weight: float, num_beams: int, space_token_id: int,
                 apply_eos_space_trick: bool, eos_space_trick_weight: float, debug: bool = False):
        super().__init__()
        if not encoder_logits.is_cuda:
            raise RuntimeError("CTCRescorerLogitsProcessor (HIP) needs device tensors; there is no CPU fallback")
        # ctc_margin > 0 is accepted and — exactly as in the reference — changes nothing: the windowing branch of CTCPrefixScoreTH.__call__ (ctc_scorer.py:127-132) needs
        # `att_w`, and the processor never passes it (`self.ctc_prefix_scorer(input_ids, self.ctc_states)`, ctc_scorer.py:330), so the scan always covers
        # [max(output_length, 1), input_length): the same scores for every margin.  tests/test_gpu_decoding.py pins that against the reference fixture.
        self.ctc_margin = int(ctc_margin or 0)
        self.pad_token_id, self.eos_token_id = pad_token_id, eos_token_id
        self.ctc_weight, self.num_beams = ctc_weight, num_beams
        self.space_token_id, self.apply_eos_space_trick = space_token_id, apply_eos_space_trick
        self.eos_space_trick_weight = eos_space_trick_weight
        self.blank = pad_token_id                  # the reference passes pad_token_id as the CTC blank (:278-284)
        self.logits = encoder_logits
        self.lens = encoder_output_lens.to(device=encoder_logits.device, dtype=torch.int32).contiguous()
        self.B, self.T, self.O = encoder_logits.shape
        self.x = None
        self.state = None                          # (r_prev, last_ids, out_len, psi) of the previous call

    def _prepare(self, W):
        dev = self.logits.device
        lg = self.logits if self.logits.stride(2) == 1 else self.logits.contiguous()
        self.x = torch.empty((self.B, self.T, self.O), dtype=torch.float32, device=dev)
        r0 = torch.empty((self.T, 2, self.B * W), dtype=torch.float32, device=dev)
        lse = torch.empty((self.B * self.T,), dtype=torch.float32, device=dev)
        rc = _lib.lib().mi_ctc_prefix_prepare(lg.data_ptr(), lg.stride(0), lg.stride(1), 0 if lg.dtype == torch.float32 else 1,
                                              self.lens.data_ptr(), self.B, self.T, self.O, self.blank, W, lse.data_ptr(),
                                              self.x.data_ptr(), r0.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "mi_ctc_prefix_prepare")
        return r0

    def ctc_scores(self, input_ids: torch.LongTensor) -> torch.Tensor:
        """CTC prefix scores (B*W, O) for extending every hypothesis by every token (= CTCPrefixScoreTH.__call__)."""
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        n_bh = input_ids.shape[0]
        W = n_bh // self.B
        dev = self.logits.device
        out_len = input_ids.shape[1] - 1
        last = input_ids[:, -1]
        psi = torch.empty((n_bh, self.O), dtype=torch.float32, device=dev)
        scores = torch.empty((n_bh, self.O), dtype=torch.float32, device=dev)
        full = self.T * 2 * n_bh * self.O * 4 <= self.FULL_STATE_BYTES
        if full:
            # the chains of EVERY (hypothesis, token) are kept (what the reference materialises, ctc_scorer.py:58-178): the state a hypothesis continues from — beam 0 of
            # its utterance (reference quirk), the token it ended on — is a column of the previous call's tensor, and each token costs ONE 250-frame scan instead of two
            if self.state is None:
                r_prev, psi_old, rp_full = self._prepare(W), None, 0
            else:
                r_prev, _, _, psi_old = self.state
                rp_full = 1
            r_all = torch.empty((self.T, 2, n_bh, self.O), dtype=torch.float32, device=dev)
            _lib.check(L.mi_ctc_prefix_score_full(self.x.data_ptr(), self.B, self.T, self.O, self.blank, W, r_prev.data_ptr(), rp_full,
                                                  psi_old.data_ptr() if psi_old is not None else None, last.data_ptr(), last.stride(0), out_len,
                                                  r_all.data_ptr(), psi.data_ptr(), scores.data_ptr(), st), "mi_ctc_prefix_score_full")
            self.state = (r_all, None, out_len, psi)
            return scores
        if self.state is None:
            r_prev = self._prepare(W)
            _lib.check(L.mi_ctc_prefix_score(self.x.data_ptr(), self.B, self.T, self.O, self.blank, W, r_prev.data_ptr(), last.data_ptr(),
                                             last.stride(0), out_len, 0, psi.data_ptr(), scores.data_ptr(), st), "mi_ctc_prefix_score")
        else:                                      # state re-selected along (beam 0 of the utterance — reference quirk —, the token every hypothesis ended on), then scored
            r_old, last_old, out_len_old, psi_old = self.state
            r_prev = torch.empty((self.T, 2, n_bh), dtype=torch.float32, device=dev)
            _lib.check(L.mi_ctc_prefix_advance(self.x.data_ptr(), self.B, self.T, self.O, self.blank, W, r_old.data_ptr(), last_old.data_ptr(), last_old.stride(0),
                                               out_len_old, psi_old.data_ptr(), last.data_ptr(), last.stride(0), out_len, r_prev.data_ptr(), psi.data_ptr(),
                                               scores.data_ptr(), st), "mi_ctc_prefix_advance")
        self.state = (r_prev, last.clone(), out_len, psi)
        return scores

    def __call__(self, input_ids: torch.LongTensor, scores: torch.FloatTensor) -> torch.FloatTensor:
        scores[:, self.pad_token_id] = LOGZERO
        ctc = self.ctc_scores(input_ids)
        next_token_scores = (1 - self.ctc_weight) * scores + self.ctc_weight * ctc
        if self.apply_eos_space_trick:            # ctc_scorer.py:333-349, verbatim semantics on small (B*W, O) tensors
            conflict = torch.logical_and(scores.argmax(dim=1) == self.eos_token_id, ctc.argmax(dim=1) == self.space_token_id)
            if conflict.any():
                on = torch.logical_and(
                    torch.logical_and(conflict, next_token_scores[:, self.eos_token_id] < next_token_scores[:, self.space_token_id]),
                    self.eos_space_trick_weight * next_token_scores[:, self.eos_token_id] > next_token_scores[:, self.space_token_id])
                if on.any():
                    next_token_scores[on, self.eos_token_id] = next_token_scores[on, self.eos_token_id] * self.eos_space_trick_weight
        return next_token_scores


class LogSoftmaxProcessor(LogitsProcessor):
    """ctc_scorer.py:357-365 — greedy decoding needs log-probabilities before the CTC mix."""

    def __call__(self, input_ids: torch.LongTensor, scores: torch.FloatTensor) -> torch.FloatTensor:
        from . import ops
        if not scores.is_cuda:
            raise RuntimeError("LogSoftmaxProcessor (HIP) needs device tensors")
        s = scores.float().contiguous()
        return s - ops.row_lse(s)[:, None]
