"""A/B: the bench step (one batch of 32 x 10 s per step) on ONE stream vs K steps pipelined over k streams (k engines with their own workspaces, the same weights):
independent batches in flight together — every step still is one pass over one batch of 32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import fbank as FB, ops, shapes, synth
from huggingface_asr_amd.engine import EBranchformerEngine

dev = torch.device("cuda:0")
cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean")
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
B = 32
tables = FB.FbankTables(80); tables.device(dev)


def run(k, steps=60, warm=10):
    engs, waves, labs = [], [], []
    for i in range(k):
        e = EBranchformerEngine(cfg, dev); e.load_state_dict(sd); engs.append(e)
        waves.append(torch.from_numpy(synth.waveforms(100 + i, B, 160000)).to(dev))
        labs.append(torch.from_numpy(synth.labels(i, B, 40, cfg["vocab_size"])).to(dev))
    streams = [torch.cuda.Stream() for _ in range(k)]
    losses = [None] * k

    def step(i):
        with torch.cuda.stream(streams[i]):
            feats, frames = FB.fbank_gpu(waves[i], tables, pad_frames_to=100)
            out = engs[i].forward(feats, frames, want_hidden=False)
            losses[i], _, _ = ops.ctc_loss(out["logits"], labs[i], out["outer_len"], reduction="mean", zero_infinity=True)
    for j in range(warm):
        step(j % k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(steps):
        step(j % k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3, [float(x) for x in losses]


for k in (1, 2, 3, 1, 2):
    ms, ls = run(k)
    print(f"streams {k}: {ms:.3f} ms per step of 32 x 10 s = {B * 10 / ms * 1e3:.0f} audio-s/s   losses {ls}", flush=True)
