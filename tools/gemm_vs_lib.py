"""Calibration only (never the product path): our bf16 GEMM kernels vs the vendor library torch dispatches to (hipBLASLt / rocBLAS) on the E-Branchformer-base shapes and
two large squares, same random operands, bias epilogue on both sides.  Each figure is a hipGraph replay of 20 launches, best of 5 (a launch from Python costs ~10 us of host
time — more than several of these kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
M = 8000
shapes = [("FFN in / cgMLP in  8000x2048x512", M, 2048, 512), ("FFN out           8000x512x2048", M, 512, 2048), ("QKV               8000x1536x512", M, 1536, 512),
          ("attention out     8000x512x512", M, 512, 512), ("cgMLP out / merge 8000x512x1024", M, 512, 1024), ("front-end out     8000x512x5120", M, 512, 5120),
          ("CTC head          8000x5008x512", M, 5008, 512), ("Whisper FFN in    24000x3072x768", 24000, 3072, 768), ("Whisper FFN out   24000x768x3072", 24000, 768, 3072),
          ("square            4096^3", 4096, 4096, 4096), ("square            8192^3", 8192, 8192, 8192)]


def bench(f, N=20):
    f(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(N): f()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            e0.record(st); g.replay(); e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / N)
    return best


print("shape | ours us | ours TF | vendor us | vendor TF | vendor / ours time")
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b32 = torch.randn(n, device=dev); b16 = b32.to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    t_ours = bench(lambda: ops.gemm(a, w, b32, out=out))
    t_lib = bench(lambda: torch.nn.functional.linear(a, w, b16))
    fl = 2.0 * m * n * k
    print(f"{name:34s} | {t_ours:8.1f} | {fl / t_ours / 1e6:7.1f} | {t_lib:8.1f} | {fl / t_lib / 1e6:7.1f} | {t_lib / t_ours:5.2f}", flush=True)
