"""Does a chip shared by several streams reward EFFICIENT tiles over chip-filling ones?  N = 512, K = 2048 GEMMs (the FFN-out shape, bf16 out):
(a) one stream, the product's 128 x 128 kernel (252 blocks per launch); (b) k streams, each launch forced onto the 256 x 256 kernel (64 blocks per launch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops

dev = "cuda:0"
M, N, K = 8000, 512, 2048
n = 8
A = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(n)]
W = [torch.randn(N, K, device=dev).to(torch.bfloat16) * 0.02 for _ in range(n)]
O = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(n)]


def run(k, variant, reps=40):
    streams = [torch.cuda.Stream() for _ in range(k)]

    def once():
        for i in range(n):
            with torch.cuda.stream(streams[i % k]):
                ops.gemm(A[i], W[i], None, out=O[i], variant=variant)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6


for k, v, name in ((1, 0, "1 stream, product kernel (128x128, 252 blocks)"), (1, 40, "1 stream, 256x256 (64 blocks)"), (2, 0, "2 streams, product kernel"),
                   (2, 40, "2 streams, 256x256"), (4, 40, "4 streams, 256x256"), (4, 0, "4 streams, product kernel")):
    us = run(k, v)
    print(f"{name:48s} {us:7.2f} us per GEMM  = {2 * M * N * K / us / 1e6:7.1f} TFLOP/s aggregate", flush=True)
