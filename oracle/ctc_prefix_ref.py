"""ORACLE (test infrastructure, never shipped): CPU restatement of the joint-decoding CTC prefix scorer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates reference `src/decoding/ctc_scorer.py`: `CTCPrefixScoreTH.__call__` (:58-178, the no-window /
full-vocabulary branch the logits processor uses), `index_select_state` (:180-207) and
`CTCRescorerLogitsProcessor.__call__` (:324-354) in a per-chain form: for every (hypothesis i, token c) the
forward variables r^n_t, r^b_t run over time, the prefix probability is their logsumexp over time.
Quirk kept: the state re-selection receives token ids, not beam*V+token indices (:327-329 vs :191), so every
beam inherits the forward variables of BEAM 0 of its utterance.
Pinned by tests/golden/ctc_prefix.npz (made from the imported reference by tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np

LOGZERO = np.float32(-10000000000.0)


def _lse(*xs):
    st = np.stack(xs).astype(np.float32)
    m = st.max(axis=0)
    return (m + np.log(np.exp(st - m).sum(axis=0, dtype=np.float32))).astype(np.float32)


def prepare_x(log_probs: np.ndarray, xlens, blank: int) -> np.ndarray:
    """:39-42 — frames beyond an utterance's length are (logzero, ..., blank = 0)."""
    x = log_probs.astype(np.float32).copy()
    for i, l in enumerate(xlens):
        if l < x.shape[1]:
            x[i, l:, :] = LOGZERO
            x[i, l:, blank] = 0
    return x


def initial_state(x: np.ndarray, blank: int, n_hyps: int):
    """:74-85 — r_prev (T, 2, B*W): r^n = logzero, r^b = cumsum of blank log-probs; s_prev = 0."""
    B, T, O = x.shape
    r = np.full((T, 2, B * n_hyps), LOGZERO, dtype=np.float32)
    cs = np.cumsum(x[:, :, blank].T, axis=0, dtype=np.float32)          # (T, B)
    r[:, 1] = np.repeat(cs, n_hyps, axis=1)
    return r


def recursion(x, blank, r_prev, last_ids, output_length, n_hyps, hyp, tok):
    """Chains (hyp[k], tok[k]), k < K: returns r (T, 2, K) and log_psi (K).  :104-167."""
    B, T, O = x.shape
    hyp, tok = np.asarray(hyp), np.asarray(tok)
    K = hyp.shape[0]
    b = hyp // n_hyps
    xc = x[b, :, tok].T.astype(np.float32)                               # (T, K)  x[t, b, c]
    xb = x[b, :, blank].T.astype(np.float32)                             # (T, K)  x[t, b, blank]
    rn, rb = r_prev[:, 0, hyp], r_prev[:, 1, hyp]                        # (T, K)
    same = (tok == np.asarray(last_ids)[hyp])[None, :]
    log_phi = np.where(same, rb, _lse(rn, rb))                           # :115-124
    r = np.full((T, 2, K), LOGZERO, dtype=np.float32)
    if output_length == 0:
        r[0, 0] = xc[0]
    start, end = max(output_length, 1), T
    for t in range(start, end):                                          # :148-151
        r[t, 0] = _lse(r[t - 1, 0], log_phi[t - 1]) + xc[t]
        r[t, 1] = _lse(r[t - 1, 0], r[t - 1, 1]) + xb[t]
    phi_x = np.concatenate([log_phi[:1], log_phi[:-1]], 0) + xc          # :154
    log_psi = _lse(*([phi_x[t] for t in range(start, end)] + [r[start - 1, 0]]))    # :164-167
    return r, log_psi


class PrefixScorer:
    """State machine of CTCRescorerLogitsProcessor (eos/space trick excluded, see processor below)."""

    def __init__(self, log_probs, xlens, blank, num_beams):
        self.blank, self.W = blank, num_beams
        self.x = prepare_x(log_probs, xlens, blank)
        self.B, self.T, self.O = self.x.shape
        self.state = None

    def step(self, input_ids: np.ndarray):
        """input_ids (B*W, len) incl. the start token -> ctc token scores (B*W, O)."""
        n_bh = input_ids.shape[0]
        W = n_bh // self.B
        out_len = input_ids.shape[1] - 1
        last = input_ids[:, -1]
        if self.state is None:
            r_prev = initial_state(self.x, self.blank, W)
            s_prev = np.zeros((n_bh, 1), np.float32)
        else:
            r_all, psi = self.state                                      # (T,2,n_bh,O), (n_bh,O)
            # :180-207 with best_ids = last token ids: hypothesis 0 of the utterance, column `token`
            src = (np.arange(n_bh) // W) * W
            r_prev = r_all[:, :, src, last]
            s_prev = psi[src, last][:, None]
        hyp = np.repeat(np.arange(n_bh), self.O)
        tok = np.tile(np.arange(self.O), n_bh)
        r, psi = recursion(self.x, self.blank, r_prev, last, out_len, W, hyp, tok)
        r = r.reshape(self.T, 2, n_bh, self.O)
        psi = psi.reshape(n_bh, self.O)
        psi[:, self.blank] = LOGZERO                                     # :173
        scores = psi - s_prev
        scores[scores == 0] = LOGZERO                                    # :176
        self.state = (r, psi)
        return scores.astype(np.float32)


def rescore(att_scores: np.ndarray, ctc_scores: np.ndarray, pad_id: int, ctc_weight: float) -> np.ndarray:
    """:325,332 (without the optional eos/space trick)."""
    s = att_scores.astype(np.float32).copy()
    s[:, pad_id] = LOGZERO
    return ((1 - ctc_weight) * s + ctc_weight * ctc_scores).astype(np.float32)
