"""Micro-benchmark of mi_gemm_bf16 on the E-Branchformer-base shapes (B=32 -> M=8000). Random data (guide rule 25)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops

dev = "cuda:0"
M = 8000
shapes = [  # (name, M, N, K, kind)
    ("ffn_in   8000x2048x512", M, 2048, 512, "gelu_bf16"),
    ("ffn_in noact bf16", M, 2048, 512, "bf16"),
    ("ffn_in f32 out", M, 2048, 512, "f32"),
    ("ffn_out  8000x512x2048", M, 512, 2048, "resid_f32"),
    ("ffn_out bf16 noresid", M, 512, 2048, "bf16"),
    ("qk       8000x1024x512", M, 1024, 512, "bf16"),
    ("wo       8000x512x512", M, 512, 512, "bf16"),
    ("cp2      8000x512x1024", M, 512, 1024, "bf16"),
    ("merge    8000x512x1024", M, 512, 1024, "resid_f32"),
    ("vT       512x8000x512", 512, M, 512, "vt"),
    ("feout    8000x512x5120", M, 512, 5120, "f32"),
    ("head     8000x5001x512", M, 5001, 512, "f32"),
    ("big      8192x8192x8192", 8192, 8192, 8192, "bf16"),
]
iters = int(os.environ.get("ITERS", "30"))
for name, m, n, k, kind in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn(m if kind == "vt" else n, device=dev)
    kw = {}
    if kind == "gelu_bf16":
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16); kw = dict(act="gelu")
    elif kind == "resid_f32":
        out = torch.randn(m, n, device=dev); kw = dict(resid=out, alpha=0.5)
    elif kind == "f32":
        out = torch.empty(m, n, device=dev)
    elif kind == "vt":
        out = torch.zeros(m, 32 * 256, device=dev, dtype=torch.bfloat16); kw = dict(bias_per_row=True, col_remap=(250, 256))
    else:
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, bias, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(a, w, bias, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:28s} {us:8.1f} us  {2.0*m*n*k/us/1e6:8.1f} TFLOP/s", flush=True)
