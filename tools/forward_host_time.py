"""host time to enqueue one forward bench step (fbank + mi_ebf_forward + CTC loss) against its GPU time: is one Python thread per GPU enough for 4 steps in flight?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import fbank as FB, ops, shapes, synth
from huggingface_asr_amd.pipeline import ForwardPipeline, reserve_hw_queues
reserve_hw_queues(4)
dev = torch.device("cuda:0")
cfg = dict(shapes.BASE, position_embeddings_type="relative", ctc_zero_infinity=True, ctc_loss_reduction="mean")
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
pipe = ForwardPipeline(cfg, dev, sd, lanes=4, wide_tiles=True)
tables = FB.FbankTables(80); tables.device(dev)
batches = [(torch.from_numpy(synth.waveforms(100 + i, 32, 160000)).to(dev), torch.from_numpy(synth.labels(i, 32, 40, cfg["vocab_size"])).to(dev)) for i in range(4)]
def step(e, lane):
    w, lab = batches[lane]
    feats, frames = FB.fbank_gpu(w, tables, pad_frames_to=100)
    out = e.forward(feats, frames, want_hidden=False)
    return ops.ctc_loss(out["logits"], lab, out["outer_len"], reduction="mean", zero_infinity=True)[0]
for _ in range(8): pipe.submit(step)
torch.cuda.synchronize()
for K in (8, 16, 32):
    t0 = time.perf_counter()
    for _ in range(K): pipe.submit(step)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{K} steps: host enqueue {(t1 - t0) / K * 1e3:.2f} ms per step, GPU {(t2 - t0) / K * 1e3:.2f} ms per step")
