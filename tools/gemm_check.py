"""correctness + speed of a GEMM kernel variant (HFASR_GEMM_VARIANT) on wide-N shapes incl. every epilogue"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
def bench(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (m, n, k) in [(8000, 2048, 512), (8000, 1536, 512), (8000, 512, 2048), (1000, 256, 64), (300, 768, 96), (8192, 8192, 8192), (8000, 2048, 32), (257, 512, 160)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev)
    ref = a.float() @ w.float().t() + b
    errs = []
    o1 = ops.gemm(a, w, b, out_dtype=torch.float32); errs.append(float((o1 - ref).abs().max()))
    o2 = ops.gemm(a, w, b, act="gelu"); errs.append(float((o2.float() - torch.nn.functional.gelu(ref)).abs().max()))
    o3 = ops.gemm(a, w, b, out_dtype=torch.float32, resid=r, alpha=0.5); errs.append(float((o3 - (r + 0.5 * ref)).abs().max()))
    o4 = ops.gemm(a, w, None); errs.append(float((o4.float() - (ref - b)).abs().max()))
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: ops.gemm(a, w, b, out=out, act="gelu"))
    print(f"{m}x{n}x{k}: max err f32 {errs[0]:.4f} gelu-bf16 {errs[1]:.4f} resid {errs[2]:.4f} nobias-bf16 {errs[3]:.4f} | gelu epilogue {t:8.1f} us {2.0*m*n*k/t/1e6:7.1f} TF", flush=True)
