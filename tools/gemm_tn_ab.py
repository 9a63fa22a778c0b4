"""A/B of the weight-gradient GEMM (mi_gemm_tn_bf16: dW += dY^T X, contraction over the M = 8000 rows) between kernel variants, interleaved in one process.
    python tools/gemm_tn_ab.py [variants, default "1,0"]      0 = 128 x 128 output tiles (product), 1 = 128 x 64"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops_train as T
dev = "cuda:0"
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,0").split(",")]
M = 8000
tot = {v: 0.0 for v in variants}
for name, N, K, cnt in [("ffn_w1 / mlp_w1", 2048, 512, 3), ("ffn_w2", 512, 2048, 2), ("qkv", 1536, 512, 1), ("wo", 512, 512, 1), ("mlp_w2 / merge", 512, 1024, 2),
                        ("head", 5008, 512, 0), ("feout", 512, 5120, 0)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    ref = dy.float().t() @ x.float()
    line = f"{name:16s} dW {N}x{K}"
    for v in variants:
        dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        T.gemm_tn_(dw, dy, x, db=db, variant=v); torch.cuda.synchronize()
        line += f" | v{v} err {float((dw - ref).abs().max() / ref.abs().max()):.1e}"
    best = {v: [] for v in variants}
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    for rnd in range(5):
        for v in variants:
            T.gemm_tn_(dw, dy, x, db=db, variant=v); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): T.gemm_tn_(dw, dy, x, db=db, variant=v)
            e1.record(); torch.cuda.synchronize()
            best[v].append(e0.elapsed_time(e1) * 50)
    for v in variants:
        t = sorted(best[v])[2]
        tot[v] += cnt * t
        line += f" | v{v} {t:6.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF"
    print(line, flush=True)
print("per encoder layer (9 dW GEMMs incl. slab reduce):", {f"v{v}": round(t, 1) for v, t in tot.items()}, "us")
