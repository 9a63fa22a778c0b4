"""ORACLE (test infrastructure, never shipped): CPU restatement of the decoding loop behind the reference's `generate()`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference's `JointCTCAttentionEncoderDecoder.generate` (src/models/ctc_encoder_plus_autoregressive_decoder.py:450-482) hands the loop to transformers'
`GenerationMixin.generate` with its own logits processors (:360-404: `LogSoftmaxProcessor` when greedy, then `CTCRescorerLogitsProcessor`).  The loop itself is
third-party code that is NOT under /root/reference: module `transformers`, pinned `==4.39.3` by the reference's requirements.txt:17, 5.15.0 installed in this image.
This file restates the INSTALLED loop — `tf:` = transformers/generation/utils.py of 5.15.0 — because that is the code the reference's generate() runs when it is
imported here, and therefore what the fixtures `tests/golden/gen_*.npz` (written by tests/golden/make_golden.py `gen` from the reference's own generate()) pin:
  * `beam_search`  = tf:3208-3560 `_beam_search` with its helpers `_get_top_k_continuations` (tf:3077-3130), `_get_running_beams_for_next_iteration` (tf:3132-3151),
    `_update_finished_beams` (tf:3153-3204), `_check_early_stop_heuristic` (tf:3008-3052), `_beam_search_has_unfinished_sequences` (tf:3055-3075); stopping criteria =
    `MaxLengthCriteria` + `EosTokenCriteria` (the two a `max_length` / `eos_token_id` configuration creates);
  * `greedy`       = tf:2783-2960 `_sample` with `do_sample=False`.
Known differences from the pinned 4.39.3 `BeamSearchScorer` (read, not runnable here — recorded in DESIGN.md §2): the early-stop heuristic there takes the best of ALL 2W
candidates of a step (an end-of-sequence candidate included) where this loop takes the best RUNNING beam, and at `max_length` it closes the W running beams (which
may include candidates ranked below W) where this loop closes the step's top W candidates.  Both divide by the number of generated tokens raised to `length_penalty`.

`score_fn(ids (B*W, cur_len) int64) -> (B*W, V) float32` returns what the loop's `logits_processor(...)` call returns: processed next-token scores of every running row."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

NEG = np.float32(-1.0e9)


def _topk(v: np.ndarray, k: int):
    """(values, indices) of the k largest per row, best first; equal values in index order (torch.topk leaves that order unspecified)."""
    idx = np.argsort(-v, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(v, idx, 1), idx


def _gather(t: np.ndarray, idx: np.ndarray) -> np.ndarray:      # tf `_gather_beams`
    return t[np.arange(t.shape[0])[:, None], idx]


def beam_search(score_fn: Callable[[np.ndarray], np.ndarray], B: int, W: int, V: int, *, max_length: int, eos: int, pad: int, start: int, length_penalty: float = 1.0,
                early_stopping=False, num_return: Optional[int] = None, trace: Optional[dict] = None, cand_fn=None):
    """-> (sequences (B * num_return, L) int64 padded with `pad`, sequences_scores (B * num_return,) float32), best first per utterance.
    `trace` (a dict) receives per step: "running" the (B, W, cur_len) prefixes, "open" which utterances could still change, "cands" the (values, indices) of the
    top 2W candidates, "margin" / "stop_gap" the smallest decision gaps.  `cand_fn(step, running, open) -> (values, indices)` replaces the candidates of a step
    (tests replay the candidates a device loop walked through these rules)."""
    num_return = W if num_return is None else num_return
    K, prompt = 2 * W, 1                                                     # beams_to_keep = max(2, 1 + n_eos) * num_beams with one EOS id (tf:3285-3286)
    top_mask = np.arange(K) < W
    running = np.full((B, W, max_length), pad, np.int64)                     # tf:3317-3324
    running[:, :, 0] = start
    sequences = running.copy()
    run_scores = np.zeros((B, W), np.float32)                                # tf:3329-3331
    run_scores[:, 1:] = NEG
    beam_scores = np.full((B, W), NEG, np.float32)
    finished = np.zeros((B, W), bool)                                        # is_sent_finished
    gen_len = np.zeros((B, W), np.int64)                                     # generated tokens of every kept hypothesis (tf keeps `beam_indices` and counts its filled entries)
    unsat = np.ones((B, 1), bool)                                            # is_early_stop_heuristic_unsatisfied
    cur_len = prompt
    while True:
        open_ = unsat[:, 0] & ~(finished.all(1) & (early_stopping is True))                                       # utterances whose kept hypotheses can still change
        if trace is not None:
            trace.setdefault("running", []).append(running[:, :, :cur_len].copy())
            trace.setdefault("open", []).append(open_.copy())
        if score_fn is not None:
            logp = np.asarray(score_fn(running[:, :, :cur_len].reshape(B * W, cur_len)), np.float32)              # tf:3398-3399 (log_softmax + processors live in score_fn)
            acc = (logp.reshape(B, W, V) + run_scores[:, :, None]).reshape(B, W * V)                                # tf:3431-3433
            topv, topi = _topk(acc, K)                                                                             # tf:3119
            if trace is not None:
                trace.setdefault("acc", []).append(acc)
        if cand_fn is not None:
            topv, topi = cand_fn(len(trace["running"]) - 1 if trace is not None else None, running[:, :, :cur_len], open_)
            topv, topi = np.asarray(topv, np.float32), np.asarray(topi, np.int64)
        if trace is not None:
            trace.setdefault("cands", []).append((topv.copy(), topi.copy()))
        beam, tok = topi // V, topi % V
        cand = _gather(running, beam)
        cand[:, :, cur_len] = tok
        hit = (tok == eos) | (cur_len + 1 >= max_length)                                                           # EosTokenCriteria | MaxLengthCriteria on the new length
        if trace is not None:
            trace.setdefault("margin", []).append((topv[:, :W] - topv[:, 1:W + 1]).min(1))
        # e. the W best candidates that did not stop run on (tf:3132-3151)
        rl = topv + hit.astype(np.float32) * NEG
        _, nidx = _topk(rl, W)
        running, run_scores = _gather(cand, nidx), _gather(rl, nidx)
        # f. candidates among the step's first W that stopped compete with the kept hypotheses (tf:3153-3204)
        s = topv / np.float32((cur_len + 1 - prompt) ** length_penalty)
        s = s + (finished.all(1, keepdims=True) & (early_stopping is True)).astype(np.float32) * NEG
        s = s + (~unsat).astype(np.float32) * NEG
        just = hit & top_mask[None]
        s = s + (~just).astype(np.float32) * NEG
        m_s, m_idx = _topk(np.concatenate([beam_scores, s], 1), W)
        sequences = _gather(np.concatenate([sequences, cand], 1), m_idx)
        finished = _gather(np.concatenate([finished, just], 1), m_idx)
        gen_len = _gather(np.concatenate([gen_len, np.full((B, K), cur_len + 1 - prompt)], 1), m_idx)
        beam_scores = m_s
        cur_len += 1
        # g. can a running beam still beat the worst kept hypothesis? (tf:3008-3052)
        hyp_len = (max_length if (early_stopping == "never" and length_penalty > 0.0) else cur_len) - prompt
        best = run_scores[:, :1] / np.float32(hyp_len ** length_penalty)
        worst = np.where(finished, beam_scores.min(1, keepdims=True), NEG)
        if trace is not None:
            trace.setdefault("stop_gap", []).append(np.where(finished.all(1) & unsat[:, 0], np.abs(best[:, 0] - beam_scores.min(1)), np.inf))
        unsat = unsat & (best > worst).any(1, keepdims=True)
        if not (unsat.any() and not (finished.all() and early_stopping is True) and not hit.all()):              # tf:3055-3075
            break
    seqs = sequences[:, :num_return].reshape(B * num_return, max_length)                                           # tf:3508-3520
    L = prompt + int(gen_len[:, :num_return].max())
    return seqs[:, :L], beam_scores[:, :num_return].reshape(-1)


def greedy(score_fn: Callable[[np.ndarray], np.ndarray], B: int, *, max_length: int, eos: int, pad: int, start: int, trace: Optional[dict] = None):
    """-> sequences (B, L) int64: argmax of the processed scores; a closed row takes `pad`; the loop ends when every row has met EOS or `max_length` (tf:2875-2945)."""
    ids = np.full((B, 1), start, np.int64)
    unfinished = np.ones((B,), bool)
    while unfinished.any():
        sc = np.asarray(score_fn(ids), np.float32)
        nxt = sc.argmax(1)
        if trace is not None:
            top2 = np.sort(sc, 1)[:, -2:]
            trace.setdefault("margin", []).append(np.where(unfinished, top2[:, 1] - top2[:, 0], np.inf))
        nxt = np.where(unfinished, nxt, pad)
        ids = np.concatenate([ids, nxt[:, None]], 1)
        unfinished = unfinished & ~((nxt == eos) | (ids.shape[1] >= max_length))
    return ids


def joint_score_fn(sd: dict, enc_cfg: dict, dec_cfg: dict, jcfg: dict, x, am, W: int, ctc_weight: float, q=None, eos_space=None):
    """The processed scores of the reference's joint decoding for the oracle models (oracle/aed_ref.py, oracle/ctc_prefix_ref.py): log_softmax of the decoder's
    last-position logits (teacher-forced recomputation of the whole prefix every step — no cache to get wrong), then CTCRescorerLogitsProcessor.__call__
    (ctc_scorer.py:324-354: pad masked, (1 - w) att + w ctc, optional eos/space trick `eos_space = (eos_id, space_id, weight)`), only when `ctc_weight > 0`
    (ctc_encoder_plus_autoregressive_decoder.py:382).  Rows are utterance-major, W rows per utterance."""
    import torch

    from . import aed_ref as A
    from . import ctc_prefix_ref as P
    q = q or (lambda t: t)
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    with torch.no_grad():
        hidden = A.E.encoder_forward(esd, enc_cfg, x, am, q)
        enc_logits = A.E.ctc_head(esd, hidden, q)
        outer = A.E.conv_out_lengths_outer(am.sum(-1), enc_cfg).long()
        enc_h = torch.nn.functional.linear(q(hidden), q(sd["enc_to_dec_proj.weight"]), sd["enc_to_dec_proj.bias"]) if "enc_to_dec_proj.weight" in sd else hidden
    B, T2 = hidden.shape[:2]
    mask = torch.arange(T2)[None] < outer[:, None]
    pad = jcfg["pad_token_id"]
    sc = P.PrefixScorer(torch.log_softmax(enc_logits, -1).numpy(), outer.numpy(), pad, W) if ctc_weight > 0 else None
    enc_rep, mask_rep = enc_h.repeat_interleave(W, 0), mask.repeat_interleave(W, 0)

    def fn(ids: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            _, logits = A.decoder_forward(sd, "decoder.", dec_cfg, torch.from_numpy(ids), enc_rep, mask_rep, None, q)
        last = logits[:, -1].float()
        if sc is None:
            return (torch.log_softmax(last, -1) if W > 1 else last).numpy()          # no processor at all: beam search normalises itself (tf:3398), greedy takes raw logits
        att = torch.log_softmax(last, -1).numpy()
        ctc = sc.step(ids)
        out = P.rescore(att, ctc, pad, ctc_weight)
        if eos_space is not None:                                                     # ctc_scorer.py:333-349
            e, s, w = eos_space
            att_m = att.copy()
            att_m[:, pad] = P.LOGZERO
            conflict = (att_m.argmax(1) == e) & (ctc.argmax(1) == s)
            on = conflict & (out[:, e] < out[:, s]) & (w * out[:, e] > out[:, s])
            out[on, e] = out[on, e] * np.float32(w)
        return out
    return fn, B
