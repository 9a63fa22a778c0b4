"""Calibration only (not part of the product path): our bf16 GEMM kernel vs the vendor library torch dispatches to (hipBLASLt / rocBLAS)
on the E-Branchformer-base shapes, same random operands, bias epilogue on both sides."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
M = 8000
shapes = [("ffn_in 8000x2048x512", M, 2048, 512), ("ffn_out 8000x512x2048", M, 512, 2048), ("qkv 8000x1536x512", M, 1536, 512), ("wo 8000x512x512", M, 512, 512),
          ("cp2 8000x512x1024", M, 512, 1024), ("head 8000x5008x512", M, 5008, 512), ("big 8192^3", 8192, 8192, 8192)]
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, m, n, k in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b32 = torch.randn(n, device=dev); b16 = b32.to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    t_ours = bench(lambda: ops.gemm(a, w, b32, out=out))
    t_lib = bench(lambda: torch.nn.functional.linear(a, w, b16))
    fl = 2.0 * m * n * k
    print(f"{name:26s} ours {t_ours:8.1f} us {fl/t_ours/1e6:7.1f} TF | vendor {t_lib:8.1f} us {fl/t_lib/1e6:7.1f} TF", flush=True)
