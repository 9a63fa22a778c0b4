"""A/B of the LDS-staged attention forward (mi_attention_qkv_bf16_v): variant 2 = the eight-wave kernel (round 4), 1 = the four-wave kernel of rounds 1-3
(variant 0, what the product calls, picks the eight-wave kernel except with relative positions at head size 64).
A launch from Python costs ~10 us of host time, more than the kernel at the bench shape: each figure is a hipGraph replay of 20 launches, best of 5.
Also prints max |new - old| (the two differ by the summation order of the keys only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)


def run(B, T, H, hd, rel, variant, causal=False, N=20):
    d = H * hd
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = (torch.randn(B * T, 3 * d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    pos = (torch.randn(2 * T - 1, d, device=dev, generator=g) * 0.5).to(torch.bfloat16) if rel else None
    u = torch.randn(d, device=dev, generator=g) * 0.1 if rel else None
    v = torch.randn(d, device=dev, generator=g) * 0.1 if rel else None
    f = lambda: ops.attention_qkv(qkv, B, T, H, pos=pos, bias_u=u, bias_v=v, causal=causal, variant=variant)
    out = f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(N): f()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            e0.record(st); gr.replay(); e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1000 / N)
    fl = 2.0 * B * H * T * T * hd * (2 + (1 if rel else 0))          # algorithmic: QK^T, PV and the T x T band of the position term
    return best, fl / best / 1e6, out.float()


shapes = [(32, 250, 4, 128), (32, 500, 4, 128), (96, 500, 4, 64), (32, 250, 8, 64), (16, 1500, 12, 64), (4, 1500, 12, 64)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, T, H, hd) in shapes:
    for rel in (True, False):
        u0, t0, o0 = run(B, T, H, hd, rel, 2)
        u1, t1, o1 = run(B, T, H, hd, rel, 1)
        print(f"B{B} T{T} H{H} hd{hd} rel={int(rel)}: eight-wave {u0:7.1f} us {t0:5.0f} TF | four-wave {u1:7.1f} us {t1:5.0f} TF | x{u1 / u0:.2f} | max diff {float((o0 - o1).abs().max()):.4f}", flush=True)
