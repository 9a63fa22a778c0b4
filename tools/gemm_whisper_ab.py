"""Which tile the Whisper-small GEMMs (M = 24000 = 16 x 1500 frames, d = 768, FFN 3072) should run on: default dispatch (256 x 256 tiles once there are >= 128 of them) against the
128 x 128 phase kernel (variant 42) — N = 768 gives 282 tiles of 256^2 on 256 CUs (1.1 rounds) or 1128 of 128^2 (4.4 rounds).  hipGraph replay of 20 launches, best of 5."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)


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


for (m, n, k, kind) in [(24000, 768, 768, "resid"), (24000, 768, 3072, "resid"), (24000, 2304, 768, "none"), (24000, 3072, 768, "gelu"), (8000, 512, 2048, "resid"), (12000, 768, 768, "resid"), (6000, 768, 3072, "resid")]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    line = f"{m}x{n}x{k} {kind:5s}"
    for v in (0, 42):
        if kind == "resid":
            x = torch.randn(m, n, device=dev)
            f = lambda: ops.gemm(a, w, b, out=x, resid=x, alpha=1.0, variant=v)
        else:
            out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            f = lambda: ops.gemm(a, w, b, out=out, act="gelu" if kind == "gelu" else "none", variant=v)
        try:
            t = bench(f)
            line += f" | v{v}: {t:7.1f} us {2.0 * m * n * k / t / 1e6:6.0f} TF"
        except Exception as e:  # noqa: BLE001
            line += f" | v{v}: {type(e).__name__}"
    print(line, flush=True)
