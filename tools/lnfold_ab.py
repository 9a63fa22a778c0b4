"""A/B of the LayerNorm-fold GEMM epilogues against the plain kernels (M = 8000): producer (fp32 + residual, optional bf16 copy / statistics) and consumer (folded LN vs plain)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import _lib, ops

dev = "cuda:0"
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
M = 8000


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K in (2048, 1024):
    N = 512
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16); b = torch.randn(N, device=dev)
    x = torch.randn(M, N, device=dev); c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); sb = torch.zeros(M, 32, device=dev)
    p = lambda t: t.data_ptr() if t is not None else None
    def prod(C2, S):
        return lambda: L.mi_gemm_resid_stats_f32(p(a), K, p(w), K, p(b), p(x), N, p(x), N, 0.5, p(C2), N, p(S), M, N, K, st)
    plain = lambda: L.mi_gemm_bf16(p(a), K, p(w), K, p(b), 1, p(x), N, 1, p(x), N, 0.5, 0, M, N, K, 0, 0, st)
    print(f"producer K={K}: plain {timeit(plain):.2f} us | entry, no extras {timeit(prod(None, None)):.2f} | +bf16 copy {timeit(prod(c2, None)):.2f} | +stats {timeit(prod(None, sb)):.2f} | both {timeit(prod(c2, sb)):.2f}")
for N, act in ((2048, 1), (1536, 0)):
    K = 512
    xb = torch.randn(M, K, device=dev).to(torch.bfloat16); wf = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    cs = wf.float().sum(-1).contiguous(); cb = torch.randn(N, device=dev); sb = torch.rand(M, 32, device=dev) * 50 + 600; out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    p = lambda t: t.data_ptr()
    fold = lambda np_: (lambda: L.mi_gemm_lnfold_bf16(p(xb), K, p(wf), K, p(cs), p(cb), p(sb), np_, 1e-5, p(out), N, act, M, N, K, st))
    plain = lambda: L.mi_gemm_bf16(p(xb), K, p(wf), K, p(cb), 1, p(out), N, 0, None, 0, 1.0, act, M, N, K, 0, 0, st)
    print(f"consumer N={N} act={act}: plain {timeit(plain):.2f} us | folded npart=16 {timeit(fold(16)):.2f} | npart=1 {timeit(fold(1)):.2f}")
