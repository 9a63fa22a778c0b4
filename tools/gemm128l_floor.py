"""loader / consumer 128 x 128 GEMM (variant 43) on whatever build HFASR_HIP_LIB names: launch time over K (fp32 + residual form)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
res = []
for k in (512, 1024, 2048, 5120):
    m, n = 8000, 512
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev); out = torch.empty((m, n), device=dev)
    for v in (42, 43):
        ops.gemm(a, w, b, out=out, resid=r, alpha=0.5, variant=v); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, w, b, out=out, resid=r, alpha=0.5, variant=v)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        res.append((k, v, sorted(ts)[3]))
name = os.environ.get("HFASR_HIP_LIB", "product").split("/")[-1]
for v in (42, 43):
    t = {k: x for k, vv, x in res if vv == v}
    print(f"{name} v{v}: " + " ".join(f"K={k}: {t[k]:.2f}" for k in t) + f"   slope {(t[5120] - t[2048]) / 48:.3f} us per K tile")
