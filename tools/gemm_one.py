"""Run a few launches of one GEMM shape (for rocprofv3 --pmc): python tools/gemm_one.py M N K [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
m, n, k = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = "cuda:0"
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm(a, w, bias, out=out)
torch.cuda.synchronize()
