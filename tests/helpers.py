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
