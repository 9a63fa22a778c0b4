"""Time the fused relative-position attention op at the bench shape (B=32, T'=250, 4 heads x 128), events over back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
B, H, d = 32, 4, 512
T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
qkv = torch.randn(B * T, 3 * d, device=dev).to(torch.bfloat16)
pos = torch.randn(2 * T - 1, d, device=dev).to(torch.bfloat16)
u, v = torch.randn(d, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
f = lambda: ops.attention_qkv(qkv, B, T, H, pos=pos, bias_u=u, bias_v=v, lengths=lens)
ref = f().float()
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
print(f"attention (rel-pos, B={B}, T={T}, {H}x{d // H}): {best:.1f} us per launch; checksum {float(ref.abs().sum()):.3f}")
