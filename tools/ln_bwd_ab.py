"""Timing of mi_layernorm_bwd at the training step's shapes (graph replay, so that the host does not pace the launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops_train as T
dev = "cuda:0"
torch.manual_seed(0)
M = 8000
for d, xdt, dydt, dxdt, acc in [(512, torch.float32, torch.bfloat16, torch.float32, True), (512, torch.float32, torch.bfloat16, torch.float32, False),
                                (1024, torch.bfloat16, torch.bfloat16, torch.bfloat16, False)]:
    x = torch.randn(M, d, device=dev).to(xdt); dy = torch.randn(M, d, device=dev).to(dydt); dx = torch.zeros(M, d, device=dev, dtype=dxdt)
    g = torch.randn(d, device=dev); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
    f = lambda: T.layernorm_bwd(x, g, dy, dx, accumulate=acc, dgamma=dg, dbeta=db)
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    byts = M * d * (x.element_size() + dy.element_size() + dx.element_size() * (2 if acc else 1))
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"d={d} x {xdt} dy {dydt} dx {dxdt} accumulate={acc}: {us:6.2f} us per call (kernel + partial reduce), {byts / us / 1e6:.2f} TB/s", flush=True)
