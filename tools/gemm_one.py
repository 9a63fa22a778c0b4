"""Run a few launches of one GEMM shape (for rocprofv3 --pmc): python tools/gemm_one.py M N K kind [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
m, n, k = (int(v) for v in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "bf16"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = "cuda:0"
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
kw = {}
if kind == "resid":
    out = torch.randn(m, n, device=dev); kw = dict(resid=out, alpha=0.5)
elif kind == "gelu":
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16); kw = dict(act="gelu")
else:
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    big.zero_()                       # flush L2 / Infinity Cache between launches (cold inputs)
    ops.gemm(a, w, bias, out=out, **kw)
torch.cuda.synchronize()
