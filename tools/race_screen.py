"""Race screen: run an LDS-DMA kernel (attention / GEMM / csgu) on one stream while another stream hammers memory, and compare with the
quiet result bit for bit.  A kernel whose LDS reads are not ordered behind its LDS-DMA writes passes quiet runs and fails here."""
import os, sys, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
B, T, H, d, I = 32, 250, 4, 512, 2048
M = B * T
qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
pos = torch.randn(2 * T - 1, d, device=dev).to(torch.bfloat16)
u, v = torch.randn(d, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
a = torch.randn(M, d, device=dev).to(torch.bfloat16); w = (torch.randn(I, d, device=dev) / d ** 0.5).to(torch.bfloat16); bias = torch.randn(I, device=dev)
h = torch.randn(M, I, device=dev).to(torch.bfloat16)
g, be = torch.ones(I // 2, device=dev), torch.zeros(I // 2, device=dev)
cw, cb = torch.randn(I // 2, 31, device=dev) * 0.2, torch.zeros(I // 2, device=dev)
cases = {
    "attention": lambda: ops.attention_qkv(qkv, B, T, H, pos=pos, bias_u=u, bias_v=v, lengths=lens),
    "gemm": lambda: ops.gemm(a, w, bias, act="gelu"),
    "csgu": lambda: ops.csgu(h, g, be, cw, cb, B, T),
    "dwconv": lambda: ops.dwconv_residual(h[:, :1024].contiguous(), cw, cb, B, T),
}
noise_src = torch.randn(64 * 1024 * 1024, device=dev)
noise_dst = torch.empty_like(noise_src)
side = torch.cuda.Stream()
for name, f in cases.items():
    ref = f().clone(); torch.cuda.synchronize()
    bad = 0
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(3): noise_dst.copy_(noise_src)
        out = f()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
    print(f"{name}: {bad}/40 runs differ under memory load", flush=True)

# pairwise: kernel X on the main stream while kernel Y runs on a side stream (co-residency on the CUs: LDS / register isolation)
print("pairwise co-residency:")
names = list(cases)
refs = {n: cases[n]().clone() for n in names}
torch.cuda.synchronize()
for x in names:
    for y in names:
        bad = 0
        for it in range(20):
            with torch.cuda.stream(side):
                for _ in range(2): oy = cases[y]()
            ox = cases[x]()
            torch.cuda.synchronize()
            if not torch.equal(ox, refs[x]) or not torch.equal(oy, refs[y]):
                bad += 1
        if bad: print(f"  {x} (main) || {y} (side): {bad}/20 runs differ", flush=True)
print("done")
