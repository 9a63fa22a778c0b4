"""Timing of the LayerNorm chain (mi_layernorm_chain) in the three forms the encoder uses, at the bench shape (8000 x 512 fp32 rows)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
M, d = 8000, int(sys.argv[1]) if len(sys.argv) > 1 else 512
x = torch.randn(M, d, device=dev)
g = [torch.randn(d, device=dev) for _ in range(6)]
oa, ob = torch.empty(M, d, device=dev, dtype=torch.bfloat16), torch.empty(M, d, device=dev, dtype=torch.bfloat16)
y = torch.empty(M, d, device=dev)
big = torch.randn(64 << 20, device=dev)           # 256 MB: evicts x from the Infinity Cache between launches when touched
forms = {
    "single LN -> bf16": lambda: ops.layernorm_chain(x, lna=(g[0], g[1]), outa=oa),
    "one LN, two affines -> 2 x bf16": lambda: ops.layernorm_chain(x, lna=(g[0], g[1]), outa=oa, lnb=(g[2], g[3]), outb=ob),
    "final LN -> fp32 x, next LN -> bf16": lambda: ops.layernorm_chain(x, ln1=(g[0], g[1]), store_y=y, lna=(g[2], g[3]), outa=oa),
}
for name, f in forms.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # launched from Python: one call costs ~10 us of host time, so time a graph replay of 20 calls instead
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"d={d} {name:40s} {e0.elapsed_time(e1) * 1000 / 20:6.2f} us per launch", flush=True)
