#!/usr/bin/env python
"""hipGraph replay of the single-call forward engine against eager launches: the ~300 kernels of one `mi_ebf_forward` are captured once
(torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream the C ABI is given) and replayed.  Interesting at small batch, where the
kernels are shorter than the launch path:   python tools/graph_latency.py [--batch 1 8 32] [--frames 1000] [--iters 50]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.engine import EBranchformerEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[1, 8, 32]); ap.add_argument("--frames", type=int, default=1000); ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dict(shapes.BASE)
eng = EBranchformerEngine(cfg, dev)
eng.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()})


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in a.batch:
    feats = torch.from_numpy(synth.normal(7, "feats", (B, a.frames, 80), 1.0)).to(dev)
    lens = torch.full((B,), a.frames - 3, dtype=torch.int32, device=dev)
    out = {}
    def eager():
        out["o"] = eng.forward(feats, lens, want_hidden=False)
    for _ in range(3):
        eager()                                            # fills the position-projection cache: nothing but launches is left to capture
    ref = out["o"]["logits"].clone()
    t_eager = timed(eager, a.iters)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eager()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        eager()
    cap = out["o"]["logits"]
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(cap, ref))
    t_graph = timed(g.replay, a.iters)
    print(json.dumps({"batch": B, "frames": a.frames, "eager_ms": round(t_eager, 3), "graph_replay_ms": round(t_graph, 3), "bit_identical": same}), flush=True)
