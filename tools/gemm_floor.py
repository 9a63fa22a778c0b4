"""Times the step's GEMM shapes on whatever build HFASR_HIP_LIB names (tools/gemm_floor_ab.sh runs it once per build: product, GEMM_FLOOR = 1 — staging only —, GEMM_FLOOR = 2 —
fragment reads + MFMAs only): per shape the launch time as a 20-launch hipGraph replay would see it (back-to-back launches, events around the batch) and the slope over K.

    python tools/gemm_floor.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
shapes = [(8000, 2048, 512, "gelu"), (8000, 2048, 1024, "gelu"), (8000, 2048, 2048, "gelu"), (8000, 1536, 512, "none"),
          (8000, 512, 512, "none"), (8000, 512, 1024, "none"), (8000, 512, 2048, "none"), (8000, 512, 4096, "none"),
          (8000, 512, 1024, "resid"), (8000, 512, 2048, "resid")]
print(os.environ.get("HFASR_HIP_LIB", "product build"))
for (m, n, k, kind) in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if kind == "resid" else None
    out = torch.empty((m, n), device=dev, dtype=torch.float32 if kind == "resid" else torch.bfloat16)

    def run():
        if kind == "resid":
            return ops.gemm(a, w, b, out=out, resid=r, alpha=0.5)
        return ops.gemm(a, w, b, out=out, act="gelu" if kind == "gelu" else "none")
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    t = sorted(ts)[len(ts) // 2]
    print(f"{m}x{n}x{k} {kind:5s} {t:7.2f} us  ({2.0 * m * n * k / t / 1e6:7.1f} TF if it were the product)", flush=True)
