"""which tile serves the GEMMs of small and mid-size batches: default dispatch against the 32 x 64 small-M tile (variant 32), chained launches on one stream (as in a forward)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
def t(fn, n=50, chain=20):
    """a graph of `chain` back-to-back launches replayed n times: the host is out of the picture"""
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(chain): fn()
    g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (n * chain) * 1e6
for M in (250, 500, 1000, 2000, 4000):
    for N, K in ((512, 512), (512, 2048), (2048, 512), (1536, 512)):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        r = {}
        for v in (0, 32, 41):
            try:
                r[v] = t(lambda: ops.gemm(a, w, bias, out=out, variant=v))
            except Exception as e:
                r[v] = float("nan")
        print(f"M={M:5d} N={N:5d} K={K:5d}: default {r[0]:6.1f} us   32x64 tiles {r[32]:6.1f} us   LDS-DMA kernels only (41) {r[41]:6.1f} us")
