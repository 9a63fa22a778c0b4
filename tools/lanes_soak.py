"""Soak of the throughput mode: N bench-size steps over `lanes` streams (wide tiles), every step's logits compared on the device with what its engine gave alone.
usage: python tools/lanes_soak.py [N=3000] [lanes=4]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.pipeline import ForwardPipeline, reserve_hw_queues
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hwq = reserve_hw_queues(lanes)
dev = torch.device("cuda:0")
cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean")
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
pipe = ForwardPipeline(cfg, dev, sd, lanes=lanes, wide_tiles=lanes >= 3)
lens = torch.full((32,), 998, dtype=torch.int32, device=dev)
feats = [torch.from_numpy(synth.normal(11 + i, "feats", (32, 1000, 80), 1.0)).to(dev) for i in range(lanes)]
refs = [pipe.engines[i].forward(feats[i], lens)["logits"].clone() for i in range(lanes)]
torch.cuda.synchronize()
bad = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(lanes)]          # per lane, updated on the lane's own stream


def one(e, lane):
    lg = e.forward(feats[lane], lens)["logits"]
    bad[lane] += (lg != refs[lane]).any().to(torch.int64)


t0 = time.perf_counter()
for _ in range(N):
    pipe.submit(one)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{N} steps over {lanes} lanes (wide tiles {lanes >= 3}, {hwq} hardware queues) in {dt:.1f} s = {dt / N * 1e3:.2f} ms per step incl. the comparison: "
      f"{int(sum(int(b) for b in bad))} steps differ from their engine's single-stream logits")
