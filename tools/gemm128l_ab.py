"""A/B of the 128 x 128 phase kernel's two forms on the N = 512 GEMMs of the step: 42 = register-pipelined (eight symmetric waves), 43 = loader / consumer (four MFMA waves +
four LDS-DMA waves).  Bit equality of every output (fp32 + residual, bf16, the LayerNorm-fold producer's bf16 copy and partial statistics, ragged M), then interleaved timing.

    python tools/gemm128l_ab.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
ok = True
for (m, n, k) in [(8000, 512, 2048), (8000, 512, 1024), (8000, 512, 512), (8000, 512, 5120), (777, 256, 384), (130, 128, 768), (48000, 256, 1024)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev)
    outs = {}
    for v in (42, 43):
        o32 = torch.full((m, n), float("nan"), device=dev)
        ops.gemm(a, w, b, out=o32, resid=r, alpha=0.5, variant=v)
        o16 = torch.full((m, n), float("nan"), device=dev, dtype=torch.bfloat16)
        ops.gemm(a, w, b, out=o16, variant=v)
        st = ops.gemm_resid_stats(a, w, b, r, alpha=0.5, variant=v) if n <= 512 and k % 128 == 0 else ()
        outs[v] = (o32, o16) + tuple(st)
    torch.cuda.synchronize()
    ref = r + 0.5 * (a.float() @ w.float().t() + b)
    eq = [bool(torch.equal(x, y)) for x, y in zip(outs[42], outs[43])]
    err = float((outs[43][0] - ref).abs().max())
    ok &= all(eq) and err < 0.05
    line = f"{m}x{n}x{k}: 42 == 43 {eq}  max err vs fp32 {err:.4f}"
    for kind in ("resid", "bf16"):
        best = {42: [], 43: []}
        out = torch.empty((m, n), device=dev, dtype=torch.float32 if kind == "resid" else torch.bfloat16)
        for rnd in range(7):
            for v in (42, 43):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    if kind == "resid":
                        ops.gemm(a, w, b, out=out, resid=r, alpha=0.5, variant=v)
                    else:
                        ops.gemm(a, w, b, out=out, variant=v)
                e1.record(); torch.cuda.synchronize()
                best[v].append(e0.elapsed_time(e1) * 1e3 / 20)
        t = {v: sorted(best[v])[3] for v in best}
        line += f" | {kind}: {t[42]:6.2f} -> {t[43]:6.2f} us ({2.0 * m * n * k / t[43] / 1e6:6.1f} TF)"
    print(line, flush=True)
print("ALL EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
