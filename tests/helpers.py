"""Shared helpers for the parity tests: rebuild the seeded inputs the golden fixtures were made from."""
import os

import numpy as np
import torch

from huggingface_asr_amd import shapes, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def seeded_state_dict(cfg, seed):
    sd = synth.state_dict_numpy(shapes.param_shapes(cfg), seed)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def synth_feats(seed, B, T, lengths):
    x = synth.normal(seed, "feats", (B, T, 80), 1.0)
    am = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lengths):
        am[b, :n] = 1
        x[b, n:] = 0.0
    return torch.from_numpy(x), torch.from_numpy(am)


def synth_labels(seed, B, U, vocab, tgt_lens):
    lab = synth.labels(seed, B, U, vocab, lo=0)
    for b, n in enumerate(tgt_lens):
        lab[b, n:] = -100
    return torch.from_numpy(lab)


def case_inputs(g, cfg):
    """Rebuild (state_dict, feats, attention_mask, labels) for a golden encoder case `g`."""
    seed = int(g["seed"])
    B, T, U = [int(v) for v in g["shape"]]
    sd = seeded_state_dict(cfg, seed)
    wsum = float(sum(v.double().sum() for v in sd.values()))
    assert abs(wsum - float(g["weight_sum"])) < 1e-6 * max(1.0, abs(wsum)), "seeded weights drifted from the fixture"
    x, am = synth_feats(seed, B, T, [int(v) for v in g["lengths"]])
    lab = synth_labels(seed, B, U, cfg["vocab_size"], [int(v) for v in g["tgt_lens"]])
    return sd, x, am, lab


TINY_DEC = dict(vocab_size=51, n_embd=128, n_layer=3, n_head=2, n_positions=64, head_locations=[1], head_weights=[0.4, 0.6],
                lsm_factor=0.1, layer_norm_epsilon=1e-5)
AED_JCFG = dict(ctc_weight=0.3, pad_token_id=50, decoder_start_token_id=2)


def aed_case_inputs(g):
    """(state_dict, feats, attention_mask, labels) for a golden AED case: weights re-seeded from the stored names/shapes."""
    import ast
    seed = int(g["seed"])
    sd = {str(n): torch.from_numpy(synth.init_param(seed, str(n), ast.literal_eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])}
    wsum = float(sum(v.double().sum() for v in sd.values()))
    assert abs(wsum - float(g["weight_sum"])) < 1e-6 * max(1.0, abs(wsum))
    B, T, U = [int(v) for v in g["shape"]]
    x, am = synth_feats(seed, B, T, [int(v) for v in g["lengths"]])
    return sd, x, am, torch.from_numpy(g["labels"])


BESTRQ_CFG = dict(best_rq_codebook_size=96, best_rq_codebook_dim=8, best_rq_in_dim=320, best_rq_num_books=2)


def bestrq_case_inputs(g):
    """(state_dict incl. the frozen rpq buffers, feats, attention_mask, mask_time_indices) of the BEST-RQ golden case."""
    import ast
    seed = int(g["seed"])
    sd = {str(n): torch.from_numpy(synth.init_param(seed, str(n), ast.literal_eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])}
    wsum = float(sum(v.double().sum() for v in sd.values()))
    assert abs(wsum - float(g["weight_sum"])) < 1e-6 * max(1.0, abs(wsum))
    sd["rpq.P"], sd["rpq.CB"] = torch.from_numpy(g["rpq_P"]), torch.from_numpy(g["rpq_CB"])
    B, T, T2 = [int(v) for v in g["shape"]]
    x, am = synth_feats(seed, B, T, [int(v) for v in g["lengths"]])
    return sd, x, am, torch.from_numpy(g["mask"])


def oracle_generate(sd, enc_cfg, dec_cfg, jcfg, x, am, W, max_length, ctc_weight, eos=1, margins=None, length_penalty=1.0, early_stopping=False, q="bf16", stop_gaps=None):
    """Joint decoding on the CPU oracle: the loop of oracle/generate_ref.py (pinned by tests/golden/gen_*.npz against the reference's own generate()) over the oracle models
    with the kernels' bf16 storage model (`q="bf16"`; `q=None`: plain fp32, what the fixtures were made with).  Returns per utterance (score, tokens, kept hypotheses
    [(score, tokens)] best first); W = 1 runs the beam loop with one beam (the same tokens as transformers' greedy loop, plus a score).
    margins / stop_gaps: lists that receive, per step, the smallest gap among the top W + 1 candidates of every utterance / the early-stop rule's gap."""
    from oracle import aed_ref as A
    from oracle import generate_ref as G
    fn, B = G.joint_score_fn(sd, enc_cfg, dec_cfg, jcfg, x, am, W, ctc_weight, q=A.E.bf16_round if q == "bf16" else q)
    V, pad, start = dec_cfg["vocab_size"], jcfg["pad_token_id"], jcfg["decoder_start_token_id"]
    tr = {}
    seq, sc = G.beam_search(fn, B, W, V, max_length=max_length, eos=eos, pad=pad, start=start, length_penalty=length_penalty, early_stopping=early_stopping, trace=tr)
    if margins is not None:
        margins.extend([[float(v) for v in row] for row in tr["margin"]])
    if stop_gaps is not None:
        stop_gaps.extend([[float(v) for v in row] for row in tr["stop_gap"]])
    out = []
    for b in range(B):
        hyps = []
        for k in range(W):
            t = seq[b * W + k].tolist()
            n = len(t)
            while n > 1 and t[n - 1] == pad:
                n -= 1
            hyps.append((float(sc[b * W + k]), t[:n]))
        out.append((hyps[0][0], hyps[0][1], hyps))
    return out


def gen_case_inputs(name):
    """(fixture, state_dict, feats, attention_mask, decoder config) of a `gen_*` fixture: seeded weights + the structured overrides of tests/gen_model.py."""
    import ast

    import gen_model as GM
    g = load_golden(name)
    seed, fixed, lengths = GM.CASES[name]
    assert seed == int(g["seed"]) and fixed == bool(int(g["fixed_pos"])) and lengths == [int(v) for v in g["lengths"]]
    sd = {str(n): torch.from_numpy(synth.init_param(seed, str(n), ast.literal_eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])}
    wsum = float(sum(v.double().sum() for v in sd.values()))
    assert abs(wsum - float(g["weight_sum"])) < 1e-6 * max(1.0, abs(wsum))
    ov = GM.overrides(seed, fixed)
    osum = float(sum(float(v.double().sum()) for v in ov.values()))
    assert abs(osum - float(g["override_sum"])) < 1e-6 * max(1.0, abs(osum)), "the structured weights drifted from the fixture"
    sd.update(ov)
    B, T = [int(v) for v in g["shape"]]
    x, am = synth_feats(seed, B, T, lengths)
    return g, sd, x, am, dict(TINY_DEC, pos_emb_fixed=fixed)
