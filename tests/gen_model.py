"""Structured weights for the `gen_*` fixtures (tests/golden/make_golden.py `gen`): the tiny joint CTC / attention model of the `aed_tiny*` fixtures with a decoder whose
next-token logits are well separated BY CONSTRUCTION, so that beam search takes its decisions with margins far above bf16 noise and the HIP path can be held to the
reference's `generate()` token for token (same idea as tests/config5_model.py at the DeCRED_base size).

Every active token has NSUCC designated successors at distinct logit levels; the end-of-sequence token is the best successor of some tokens and the third-best of others, so
greedy and beam search close hypotheses at different depths and some beams only close at `max_length`.  The encoder's blank projection gets a large bias: with a blank-dominated
CTC posterior a short prefix is a plausible complete labelling, i.e. the CTC prefix scorer does not veto the end-of-sequence token (with random posteriors its score is ~ -150).
Everything else (the transformer blocks, cross-attention, the encoder) keeps its seeded random weights, so the KV cache, its beam re-ordering and the CTC prefix scores still
shape the candidates' values.  Only constants and huggingface_asr_amd.synth are used: the generator (which loads these tensors into the REFERENCE model) and the tests (which
load them into the HIP model and the oracle) build the same numbers."""
import numpy as np
import torch

from huggingface_asr_amd import synth

D, V = 128, 51
EOS, START, PAD = 1, 2, 50
ACTIVE, NACT, NSUCC = 3, 40, 5          # tokens [3, 43) are "active": every token's successors lie among them (or are EOS)
GAMMA = 22.0
STEP, SPREAD = 0.2, 1.0      # level of the k-th successor of token t: 1 - STEP * k * (1 + SPREAD * u_t), u_t in [0, 1) per token
BLANK_BIAS = 6.0


def successors(t: int):
    s = [ACTIVE + (7 * (t % NACT) + 11 * k + 3) % NACT for k in range(NSUCC)]
    if t % 7 == 3:
        s[0] = EOS              # the best continuation closes the hypothesis
    elif t % 3 == 0:
        s[2] = EOS              # a beam of rank 2 closes while better ones run on
    return s


def overrides(seed: int, fixed_pos: bool) -> dict:
    """name -> tensor for the state-dict entries the structure replaces (keys of the reference's JointCTCAttentionEncoderDecoder)."""
    # token directions: mutually orthogonal, orthogonal to the all-ones vector (LayerNorm's mean removal) and — fixed positions — to the position vectors of the first 32 steps
    fixed = [torch.ones(1, D)]
    if fixed_pos:
        inv = 1 / (10000 ** (torch.arange(0.0, D, 2.0) / D))
        ang = torch.outer(torch.arange(32.0), inv)
        fixed.append(torch.cat([ang.sin(), ang.cos()], -1))
    fixed = torch.cat(fixed, 0)
    rnd = torch.from_numpy(synth.normal(seed, "gen/emb", (NACT + 1, D), 1.0))
    qm, _ = torch.linalg.qr(torch.cat([fixed, rnd], 0).double().t())
    dirs = (qm[:, fixed.shape[0]:fixed.shape[0] + NACT + 1].t() * (D ** 0.5)).float()              # (NACT + 1, D), norm sqrt(D)
    emb = torch.from_numpy(synth.normal(seed, "gen/emb_rest", (V, D), 1.0))
    toks = list(range(ACTIVE, ACTIVE + NACT)) + [START]
    emb[toks] = dirs
    u = torch.from_numpy(synth.uniform(seed, "gen/u", (V,), 0.0, 1.0))
    head = torch.zeros(V, D)
    for t in toks:
        for k, v in enumerate(successors(t)):
            head[v] += GAMMA * (1.0 - STEP * k * (1.0 + SPREAD * float(u[t]))) / D * emb[t]
    out = {"decoder.lm_head.weight": head, "decoder.transformer.ln_f.weight": torch.ones(D), "decoder.transformer.ln_f.bias": torch.zeros(D),
           "encoder.blank_projection.bias": torch.full((1,), BLANK_BIAS)}
    if fixed_pos:           # AdaptiveEmbedding multiplies by sqrt(d) (reference src/models/embeddings.py:60)
        out["decoder.transformer.wte.emb_layers.0.weight"] = emb / (D ** 0.5)
    else:
        out["decoder.transformer.wte.weight"] = emb
    return out


# name -> (seed, fixed positions, frame lengths of the utterances).  Seeds: the best of 120 by the smallest decision margin over all settings (a beam search of 13
# steps x 5 beams takes a few hundred decisions between real-valued sums: the best seeds reach ~0.013, bf16 noise on these sums is ~0.01-0.05 — which is why the GPU test
# compares token for token only up to a decision the fixture itself certifies as a near tie, tests/test_gpu_generate.py); both have hypotheses closed by EOS at several
# depths and hypotheses closed by max_length among the five kept.
CASES = {
    "gen_tiny": (11, False, [198, 131]),
    "gen_tiny_fixedpos": (8, True, [198, 131]),
}
# generation settings every case is decoded with: (num_beams, length_penalty, early_stopping, max_length)
SETTINGS = [(1, 1.0, False, 14), (3, 1.0, False, 14), (5, 1.0, False, 14), (5, 0.6, False, 14), (5, 1.6, False, 14), (3, 1.0, True, 14), (5, 1.0, True, 14),
            (3, 1.0, "never", 14), (5, 1.6, "never", 14), (5, 1.0, False, 6), (1, 1.0, False, 5)]


def setting_key(W, lp, es, ml):
    return f"W{W}_lp{lp}_es{es}_ml{ml}"
