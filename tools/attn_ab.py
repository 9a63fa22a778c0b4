"""Timing of the LDS-staged attention (mi_attention_qkv_bf16) at the bench shape and around it: fixed cost vs per-key-step cost, relative-position share."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)


def run(B, T, H, hd, rel, causal=False, reps=50):
    d = H * hd
    qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
    pos = (torch.randn(2 * T - 1, d, device=dev) * 0.5).to(torch.bfloat16) if rel else None
    u = torch.randn(d, device=dev) * 0.1 if rel else None
    v = torch.randn(d, device=dev) * 0.1 if rel else None
    for _ in range(3):
        ops.attention_qkv(qkv, B, T, H, pos=pos, bias_u=u, bias_v=v, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attention_qkv(qkv, B, T, H, pos=pos, bias_u=u, bias_v=v, causal=causal)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    fl = 2.0 * B * H * T * T * hd * (2 + (2 if rel else 0))
    return us, fl / us / 1e6


shapes = [(32, 250, 4, 128)] if len(sys.argv) > 1 else [(32, 250, 4, 128), (32, 128, 4, 128), (64, 128, 4, 128), (32, 256, 4, 128), (32, 500, 4, 128), (32, 250, 8, 64), (16, 1500, 12, 64)]
for (B, T, H, hd) in shapes:
    for rel in (True, False):
        us, tf = run(B, T, H, hd, rel)
        print(f"B{B} T{T} H{H} hd{hd} rel={int(rel)}: {us:7.1f} us  {tf:6.0f} TF", flush=True)
