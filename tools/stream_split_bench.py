"""A/B: the bench step (32 x 10 s) as ONE forward of 32 utterances vs k concurrent forwards of 32/k utterances on k streams (engine workspace slots)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import fbank as FB, ops, shapes, synth
from huggingface_asr_amd.engine import EBranchformerEngine

dev = torch.device("cuda:0")
cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean")
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
B = 32
wave = torch.from_numpy(synth.waveforms(100, B, 160000)).to(dev)
labels = torch.from_numpy(synth.labels(0, B, 40, cfg["vocab_size"])).to(dev)
tables = FB.FbankTables(80); tables.device(dev)


def run(k, steps=60, warm=10):
    engs = []
    for i in range(k):
        e = EBranchformerEngine(cfg, dev); e.load_state_dict(sd); engs.append(e)
    streams = [torch.cuda.Stream() for _ in range(k)] if k > 1 else [torch.cuda.current_stream()]
    n = B // k
    losses = [None] * k

    def step():
        if k == 1:
            feats, frames = FB.fbank_gpu(wave, tables, pad_frames_to=100)
            out = engs[0].forward(feats, frames, want_hidden=False)
            losses[0], _, _ = ops.ctc_loss(out["logits"], labels, out["outer_len"], reduction="mean", zero_infinity=True)
            return
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                feats, frames = FB.fbank_gpu(wave[i * n:(i + 1) * n], tables, pad_frames_to=100)
                out = engs[i].forward(feats, frames, want_hidden=False)
                losses[i], _, _ = ops.ctc_loss(out["logits"], labels[i * n:(i + 1) * n], out["outer_len"], reduction="mean", zero_infinity=True)
        for s in streams:
            cur.wait_stream(s)
    for _ in range(warm):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    loss = float(sum(float(l) for l in losses) / k)
    print(f"streams={k}: {dt * 1e3:.3f} ms/step  {B * 10 / dt:.0f} audio-s/s  loss {loss:.4f}", flush=True)


for k in (1, 2, 4, 1, 2):
    run(k)
