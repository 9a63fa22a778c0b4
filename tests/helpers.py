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


def oracle_generate(sd, enc_cfg, dec_cfg, jcfg, x, am, W, max_length, ctc_weight, eos=1, margins=None):
    """The same joint decoding loop on the CPU oracle (teacher-forced recomputation each step, oracle prefix scorer).
    margins: a list that receives, per step and utterance, the smallest gap between consecutive candidates among the top W + 1 (the decisions beam search takes)."""
    from oracle import aed_ref as A
    from oracle import ctc_prefix_ref as P
    q = A.E.bf16_round
    esd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    with torch.no_grad():
        hidden = A.E.encoder_forward(esd, enc_cfg, x, am, q)
        enc_logits = A.E.ctc_head(esd, hidden, q)
        outer = A.E.conv_out_lengths_outer(am.sum(-1), enc_cfg).long()
        enc_h = torch.nn.functional.linear(q(hidden), q(sd["enc_to_dec_proj.weight"]), sd["enc_to_dec_proj.bias"]) if "enc_to_dec_proj.weight" in sd else hidden
    B, T2 = hidden.shape[:2]
    mask = torch.arange(T2)[None] < outer[:, None]
    pad, start, V = jcfg["pad_token_id"], jcfg["decoder_start_token_id"], dec_cfg["vocab_size"]
    sc = P.PrefixScorer(torch.log_softmax(enc_logits, -1).numpy(), outer.numpy(), pad, W)
    ids = torch.full((B * W, 1), start, dtype=torch.long)
    beam_scores = torch.zeros(B, W); beam_scores[:, 1:] = -1e9; beam_scores = beam_scores.view(-1)
    finished, done = [[] for _ in range(B)], [False] * B
    enc_rep, mask_rep = enc_h.repeat_interleave(W, 0), mask.repeat_interleave(W, 0)
    while ids.shape[1] < max_length and not all(done):
        with torch.no_grad():
            _, logits = A.decoder_forward(sd, "decoder.", dec_cfg, ids, enc_rep, mask_rep, None, q)
        att = torch.log_softmax(logits[:, -1].float(), -1).numpy()
        ctc = sc.step(ids.numpy())
        scores = torch.from_numpy(P.rescore(att, ctc, pad, ctc_weight))
        cand = (scores + beam_scores[:, None]).view(B, W * V)
        top_s, top_i = cand.topk(2 * W, dim=1)
        if margins is not None:
            margins.append([float((top_s[b, :W] - top_s[b, 1:W + 1]).min()) for b in range(B)])
        cur_len = ids.shape[1]
        nb = []
        for b in range(B):
            row = []
            if done[b]:
                nb.append([(0.0, pad, b * W)] * W); continue
            for rank in range(2 * W):
                s, idx = float(top_s[b, rank]), int(top_i[b, rank])
                beam, tok = idx // V, idx % V
                if tok == eos:
                    if rank < W:
                        finished[b].append((s / cur_len, ids[b * W + beam].tolist() + [tok]))
                else:
                    row.append((s, tok, b * W + beam))
                if len(row) == W:
                    break
            nb.append(row)
            if len(finished[b]) >= W and float(top_s[b].max()) / cur_len <= sorted(finished[b], key=lambda t: -t[0])[W - 1][0]:
                done[b] = True
        beam_idx = torch.tensor([r[2] for row in nb for r in row])
        new_tok = torch.tensor([r[1] for row in nb for r in row])[:, None]
        beam_scores = torch.tensor([r[0] for row in nb for r in row])
        ids = torch.cat([ids[beam_idx], new_tok], 1)
    out = []
    bs = beam_scores.view(B, W)
    for b in range(B):
        if not done[b]:
            for k in range(W):
                finished[b].append((float(bs[b, k]) / ids.shape[1], ids[b * W + k].tolist()))
        out.append(max(finished[b], key=lambda t: t[0]) + (sorted(finished[b], key=lambda t: -t[0]),))      # (score, tokens, every kept hypothesis best first)
    return out


