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
a = ap.parse_args()
world, rank, local = PL.env_world()
dev = torch.device("cuda", local); torch.cuda.set_device(dev)
PL.init("nccl", dev)
base = {"base": shapes.BASE, "small": shapes.SMALL, "tiny": shapes.TINY}[a.size]
cfg = dict(base, position_embeddings_type=a.pos, ctc_zero_infinity=True, ctc_loss_reduction="mean", hidden_dropout=0.0, activation_dropout=0.0,
           attention_dropout=0.0, final_dropout=0.0, feat_proj_dropout=0.0, csgu_conv_dropout=0.0, layerdrop=0.0, apply_spec_augment=False)
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
tr = EncoderCTCTrainer(cfg, dev, lr=2e-3, weight_decay=1e-6)
tr.load_state_dict(sd)
B, T = a.batch, a.frames
feats = torch.from_numpy(synth.normal(100 + rank, "feats", (B, T, 80), 1.0)).to(dev)
lens = torch.full((B,), T - 2, dtype=torch.int32, device=dev)
labels = torch.from_numpy(synth.labels(rank, B, 40, cfg["vocab_size"])).to(dev)
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
    sec = world * B * T / 100.0 * a.steps
    print(json.dumps({"metric": "audio-seconds/sec, encoder+CTC TRAIN step (fwd+bwd+AdamW)" if not a.fwd_only else "train-mode forward only",
                      "value": round(sec / dt, 1), "ms_per_step": round(dt / a.steps * 1e3, 2), "n_gpus": world, "size": a.size, "per_gpu_batch": B,
                      "frames": T, "loss": round(float(state["o"]["loss"]), 4), "n_params": tr.store.n,
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
