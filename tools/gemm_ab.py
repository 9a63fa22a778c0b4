"""A/B of GEMM kernel variants in ONE process (guide rule 24): correctness of each variant against an fp32 reference, then interleaved timing rounds.

    python tools/gemm_ab.py [variants, default "41,0"]  [--cold]

Variants are the per-call kernel selection of mi_gemm_bf16_v (0 default dispatch, 40 = phase kernels wherever supported, 41 = never,
42 / 47 = 128x128 phase kernel pipelined / two-segment, 30 = older 128x128 tiles).  --cold writes 512 MB to another buffer between launches (the state a real step leaves the memory system in).
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops, _lib

dev = "cuda:0"
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "41,0").split(",")]
cold = "--cold" in sys.argv
torch.manual_seed(0)

shapes = [  # M, N, K, kind
    (8000, 2048, 512, "gelu"), (8000, 2048, 512, "none"), (8000, 1536, 512, "none"), (8000, 2048, 128, "gelu"), (8000, 2048, 192, "none"),
    (8000, 512, 2048, "resid"), (8000, 512, 1024, "resid"), (8000, 512, 512, "none"), (8000, 512, 1024, "none"), (8000, 512, 5120, "f32"), (8000, 5001, 512, "f32"), (300, 517, 192, "f32"),
    (777, 512, 320, "gelu"), (777, 384, 320, "resid"), (256, 256, 128, "none"), (4096, 4096, 4096, "none"), (8192, 8192, 8192, "none"),
]
scratch = torch.empty(128 * 1024 * 1024, device=dev) if cold else None


def run(a, w, b, out, kind, r, v=0):
    if kind == "resid":
        return ops.gemm(a, w, b, out=out, resid=r, alpha=0.5, variant=v)
    if kind == "f32":
        return ops.gemm(a, w, b, out=out, variant=v)
    return ops.gemm(a, w, b, out=out, act="gelu" if kind == "gelu" else "none", variant=v)


for (m, n, k, kind) in shapes:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    ref = a.float() @ w.float().t() + b
    r0 = torch.randn(m, n, device=dev) if kind == "resid" else None
    if kind == "gelu":
        ref = torch.nn.functional.gelu(ref)
    elif kind == "resid":
        ref = r0 + 0.5 * ref
    line = f"{m}x{n}x{k} {kind:5s}"
    outs = {}
    for v in variants:
        ldp = (n + 7) // 8 * 8                       # rows padded to a multiple of 8 elements, as the engine's logits buffer (16-B stores)
        out = torch.full((m, ldp), float("nan"), device=dev, dtype=torch.float32 if kind in ("resid", "f32") else torch.bfloat16)[:, :n]
        r = r0.clone() if r0 is not None else None
        run(a, w, b, out, kind, r, v)
        torch.cuda.synchronize()
        err = float((out.float() - ref).abs().max())
        outs[v] = out
        line += f" | v{v} err {err:.4f}"
    if len(variants) > 1:
        line += f" | v{variants[0]}==v{variants[-1]}: {bool(torch.equal(outs[variants[0]], outs[variants[-1]]))} maxdiff {float((outs[variants[0]].float() - outs[variants[-1]].float()).abs().max()):.5f}"
    # interleaved timing rounds
    iters = 20 if m * n * k < 1e11 else 5
    out = torch.empty((m, (n + 7) // 8 * 8), device=dev, dtype=torch.float32 if kind in ("resid", "f32") else torch.bfloat16)[:, :n]
    r = r0
    best = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            run(a, w, b, out, kind, r, v)
            torch.cuda.synchronize()
            if cold:
                ts = []
                for _ in range(iters):
                    scratch.fill_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(a, w, b, out, kind, r, v); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                best[v].append(sorted(ts)[len(ts) // 2])
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run(a, w, b, out, kind, r, v)
                e1.record(); torch.cuda.synchronize()
                best[v].append(e0.elapsed_time(e1) * 1e3 / iters)
    for v in variants:
        t = sorted(best[v])[len(best[v]) // 2]
        line += f" | v{v} {t:7.1f} us {2.0 * m * n * k / t / 1e6:7.1f} TF (min {min(best[v]):.1f})"
    print(line, flush=True)
