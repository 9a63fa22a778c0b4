"""Instrumented build of the loader / consumer 128 x 128 GEMM (-DGEMM128_STAMPS, tools/bin/libhfasr_stamps.so via HFASR_HIP_LIB): per block the loader wave's cycles in
its K loop — issuing pieces / waiting for them to land (vmcnt) / waiting at the barrier — and the consumer wave's cycles waiting for its LDS reads / at the barrier.

    HFASR_HIP_LIB=tools/bin/libhfasr_stamps.so python tools/gemm128l_stamps.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
for (m, n, k) in [(8000, 512, 2048), (8000, 512, 5120)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev)
    buf = torch.zeros((m + 8, n), device=dev)
    for _ in range(3):
        ops.gemm(a, w, b, out=buf[:m], resid=r, alpha=0.5, variant=43)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, w, b, out=buf[:m], resid=r, alpha=0.5, variant=43)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nb = (m + 127) // 128 * (n // 128)
    st = buf[m:].reshape(-1)[: nb * 8].reshape(nb, 8).cpu()
    nkt = k // 64
    med = st.median(0).values
    print(f"{os.environ.get('HFASR_HIP_LIB', '').split('/')[-1]} {m}x{n}x{k}: {nkt} K tiles, per K tile (median over {nb} blocks, shader cycles): loader issue {med[0] / (nkt - 4):.0f} | vmcnt wait {med[1] / (nkt - 4):.0f} | barrier wait {med[2] / (nkt - 4):.0f} | "
          f"loop {med[3] / (nkt - 4):.0f}   consumer: LDS-read wait {med[4] / nkt:.0f} | barrier wait {med[5] / nkt:.0f}   kernel start -> end of K loop {med[6]:.0f} counts, launch {us:.2f} us "
          f"(counter >= {med[6] / us / 1e3:.2f} GHz)")
