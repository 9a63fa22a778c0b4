"""GEMM duration vs K at fixed M x N (run under rocprofv3 --kernel-trace; kernel durations are read from the trace in launch order)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from huggingface_asr_amd import ops
dev = "cuda:0"
REP = 12
for (M, N) in ((8000, 2048), (8000, 512)):
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        for _ in range(REP): ops.gemm(a, w, b, out=out)
        torch.cuda.synchronize()
print("done")
