"""The bench step with the waveforms arriving from pinned HOST memory every step (never the reported `value`)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import fbank as FB, ops, shapes, synth
from huggingface_asr_amd.engine import EBranchformerEngine
dev = torch.device("cuda", 0)
cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean")
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
eng = EBranchformerEngine(cfg, dev); eng.load_state_dict(sd)
B = 32
host = torch.from_numpy(synth.waveforms(100, B, 160000)).pin_memory()
labels = torch.from_numpy(synth.labels(0, B, 40, cfg["vocab_size"])).to(dev)
tables = FB.FbankTables(80); tables.device(dev)
resident = host.to(dev)
def step(w):
    feats, frames = FB.fbank_gpu(w, tables, pad_frames_to=100)
    out = eng.forward(feats, frames, want_hidden=False)
    return ops.ctc_loss(out["logits"], labels, out["outer_len"], reduction="mean", zero_infinity=True)[0]
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
t_res = timed(lambda: step(resident))
t_h2d = timed(lambda: step(host.to(dev, non_blocking=True)))
copy = torch.cuda.Stream(); bufs = [torch.empty_like(resident) for _ in range(2)]; evs = [torch.cuda.Event() for _ in range(2)]
def pipelined():
    # double-buffered: the copy of step i+1 runs on its own stream (a DMA engine, no kernels) while step i computes
    i = pipelined.i = getattr(pipelined, "i", 0) + 1
    with torch.cuda.stream(copy):
        bufs[i & 1].copy_(host, non_blocking=True); evs[i & 1].record(copy)
    torch.cuda.current_stream().wait_event(evs[(i - 1) & 1]) if i > 1 else None
    step(bufs[(i - 1) & 1] if i > 1 else resident)
t_pipe = timed(pipelined)
print(f"resident {B * 10 / t_res:.0f} audio-s/s ({t_res * 1e3:.2f} ms/step) | H2D in line {B * 10 / t_h2d:.0f} ({t_h2d * 1e3:.2f} ms) | H2D double-buffered on a copy stream {B * 10 / t_pipe:.0f} ({t_pipe * 1e3:.2f} ms); {host.numel() * 4 / 1e6:.1f} MB per step")
