"""A/B of the implicit-GEMM conv (mi_conv2d_cl_bf16) between kernel variants: equality of outputs, error vs torch conv2d, timing (bench shape)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops, _lib
dev = "cuda:0"
torch.manual_seed(0)
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "41,40").split(",")]
for (B, T, Fd, Cin, Cout, causal) in [(2, 61, 40, 64, 256, False), (3, 100, 40, 128, 256, True), (32, 500, 40, 256, 256, False)]:
    x = torch.randn(B, T, Fd, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev) * 0.1
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    xin = x.float().permute(0, 3, 1, 2)
    ref = F.gelu(F.conv2d(F.pad(xin, (2, 0, 2, 0)) if causal else xin, w.float(), b, stride=2, padding=0 if causal else 1)).permute(0, 2, 3, 1)
    outs = {}
    line = f"B{B} T{T} F{Fd} Cin{Cin} causal={causal}"
    for v in variants:
        o = ops.conv2d_cl(x, wp, b, causal=causal, variant=v)
        torch.cuda.synchronize()
        outs[v] = o
        line += f" | v{v} err {float((o.float() - ref).abs().max()):.4f}"
    line += f" | equal {bool(torch.equal(outs[variants[0]], outs[variants[-1]]))}"
    for rnd in range(2):
        for v in variants:
            for _ in range(2): ops.conv2d_cl(x, wp, b, causal=causal, variant=v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.conv2d_cl(x, wp, b, causal=causal, variant=v)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            M = o.numel() // Cout
            if rnd: line += f" | v{v} {us:.1f} us {2.0 * M * Cout * 9 * Cin / us / 1e6:.0f} TF"
    print(line, flush=True)
