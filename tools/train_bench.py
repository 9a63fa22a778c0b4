#!/usr/bin/env python
"""Training-step timing of the encoder+CTC trainer (BASELINE config 3's encoder side): fwd + bwd + AdamW on one GPU, or DP over
the ranks torchrun starts (gradient all-reduce over RCCL).  Not the headline bench (bench.py): a secondary measurement.

    python tools/train_bench.py [--size base|small] [--batch 32] [--steps 10] [--warmup 3] [--pos relative]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from huggingface_asr_amd import shapes, synth, parallel as PL
from huggingface_asr_amd.train import EncoderCTCTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="base"); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--pos", default="relative"); ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--no-dw-overwrite", action="store_true", help="the grouped weight-gradient launches always add into their targets (the form before), for a same-box A/B")
ap.add_argument("--no-ctc-from-bwd", action="store_true", help="the CTC loss by its forward kernel (the form before), for a same-box A/B")
ap.add_argument("--no-walk-qb", action="store_true", help="q + pos_bias_u / q + pos_bias_v of the attention backward by a pass of their own (the form before), for a same-box A/B")
ap.add_argument("--no-dual-ln", action="store_true", help="the layer's two branch LayerNorm backward passes as two launches (the form before round 4's last change), for a same-box A/B")
ap.add_argument("--specaug", action="store_true", help="in-model SpecAugment as in the recipes (mask_time_prob 0.05, length 10, min 2 masks)")
ap.add_argument("--dropout", type=float, default=0.0, help="the recipes train with 0.1 at every dropout site (hidden, activation, attention, CSGU, final)")
ap.add_argument("--finetune", action="store_true", help="the frozen fine-tuning recipes' setting (recipes/librispeech/ssl/*/lumi/finetune_frozen*.sh): layer mixing + "
                                                        "additional layer before the CTC head, encoder layers frozen")
ap.add_argument("--model", default="ctc", choices=["ctc", "aed"], help="aed = BASELINE config 3: small encoder + 6x256 GPT-2 decoder, ctc_weight 0.3, lsm 0.1, "
                                                                         "fixed positions, per-GPU batch 96, lengths uniform 1-20 s sorted into the batch")
a = ap.parse_args()
world, rank, local = PL.env_world()
dev = torch.device("cuda", local); torch.cuda.set_device(dev)
PL.init("nccl", dev)
base = {"base": shapes.BASE, "small": shapes.SMALL, "tiny": shapes.TINY}[a.size]
cfg = dict(base, position_embeddings_type=a.pos, ctc_zero_infinity=True, ctc_loss_reduction="mean", hidden_dropout=a.dropout, activation_dropout=a.dropout,
           attention_dropout=a.dropout, final_dropout=a.dropout, feat_proj_dropout=0.0, csgu_conv_dropout=a.dropout, layerdrop=0.0, apply_spec_augment=a.specaug,
           mask_time_prob=0.05, mask_time_length=10, mask_time_min_masks=2,
           finetune_with_additional_layer=a.finetune, finetune_with_layer_mixing=a.finetune)
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
B, T = a.batch, a.frames
if a.model == "aed":
    from huggingface_asr_amd.train_aed import JointAEDTrainer, _dec_map, decoder_specs
    dcfg = dict(vocab_size=5000, n_embd=256, n_layer=6, n_head=4, n_positions=1024, head_locations=[], head_weights=[1.0], lsm_factor=0.1,
                layer_norm_epsilon=1e-5, pos_emb_fixed=True, tie_word_embeddings=False)
    jcfg = dict(ctc_weight=0.3, pad_token_id=3, decoder_start_token_id=1)
    tr = JointAEDTrainer(cfg, dcfg, jcfg, dev, lr=2e-3, weight_decay=1e-6)
    tr.enc.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    for s_ in tr.store.specs.values():                      # seeded decoder weights straight into the packed store
        t = tr.store.p(s_.name)
        t.copy_((torch.ones(s_.shape) if s_.name.endswith("_g") else torch.randn(s_.shape, generator=g) * (0.0 if s_.name.endswith(("_b", "bqkv", "bq", "bkv", "bo", "bco", "bfc", "bpr")) else 0.02)).to(dev))
    tr.store.refresh_mirrors(cast=True)
    rng = np.random.default_rng(rank)
    fl = np.sort(rng.integers(100, 2001, size=B))[::-1].copy()           # 1-20 s, longest first (length-grouped batch)
    T = int((fl.max() + 99) // 100 * 100)
    U = 60
else:
    tr = EncoderCTCTrainer(cfg, dev, lr=2e-3, weight_decay=1e-6)
    tr.load_state_dict(sd)
    if a.finetune:
        tr.set_frozen({k for k in sd if k.startswith("wav2vec2.encoder.")})          # freeze_encoder(): train_ctc_asr.py:51-52
    fl = np.full((B,), T - 2)
    U = 40
feats = torch.from_numpy(synth.normal(100 + rank, "feats", (B, T, 80), 1.0)).to(dev)
lens = torch.from_numpy(fl.astype(np.int32)).to(dev)
labels = torch.from_numpy(synth.labels(rank, B, U, cfg["vocab_size"], lo=5)).to(dev)
if a.model == "aed":                                        # label lengths follow the audio lengths (about 3 tokens / s)
    for b in range(B):
        labels[b, max(2, int(fl[b] / 100 * 3)):] = -100
if a.no_dual_ln:
    (tr.enc if a.model == "aed" else tr).dual_ln = False
if a.no_dw_overwrite:
    (tr.enc if a.model == "aed" else tr).dw_overwrite = False
if a.no_ctc_from_bwd:
    (tr.enc if a.model == "aed" else tr).ctc_from_bwd = False
if a.no_walk_qb:
    (tr.enc if a.model == "aed" else tr).walk_qb = False
state = {}
def step():
    if a.fwd_only:
        state["o"] = tr.forward_backward(feats, lens, labels, backward=False)
    else:
        state["o"] = tr.train_step(feats, lens, labels)
for _ in range(a.warmup):
    step()
dt = PL.timed(step, a.steps, sync=torch.cuda.synchronize, device=dev)
if rank == 0:
    sec = world * float(fl.sum()) / 100.0 * a.steps
    print(json.dumps({"metric": "audio-seconds/sec, encoder+CTC TRAIN step (fwd+bwd+AdamW)" if not a.fwd_only else "train-mode forward only",
                      "value": round(sec / dt, 1), "ms_per_step": round(dt / a.steps * 1e3, 2), "n_gpus": world, "size": a.size, "model": a.model, "per_gpu_batch": B,
                      "frames": T, "loss": round(float(state["o"]["loss"]), 4), "n_params": tr.store.n + (tr.enc.store.n if a.model == "aed" else 0),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
