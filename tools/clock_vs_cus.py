"""The shader clock the chip holds under the loader / consumer GEMM's K loop (LDS-DMA staging + fragment reads + MFMAs on every active CU) as a function of how many CUs are
active: the instrumented build's cycle count per K tile (tools/gemm128l_stamps.py) against the wall time per K tile (slope over K), for M = 1024 ... 8000 rows (32 ... 252 blocks).

    HFASR_HIP_LIB=tools/bin/libhfasr_stamps.so python tools/clock_vs_cus.py          (and ..._stamps_nostage.so: the same loop without the staging)
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
name = os.environ.get("HFASR_HIP_LIB", "").split("/")[-1]
n = 512
for m in (1024, 2048, 4096, 8000):
    res = {}
    for k in (2048, 5120):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
        b = torch.randn(n, device=dev); r = torch.randn(m, n, device=dev)
        buf = torch.zeros((m + 8, n), device=dev)
        for _ in range(3):
            ops.gemm(a, w, b, out=buf[:m], resid=r, alpha=0.5, variant=43)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, w, b, out=buf[:m], resid=r, alpha=0.5, variant=43)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        nb = (m + 127) // 128 * (n // 128)
        st = buf[m:].reshape(-1)[: nb * 8].reshape(nb, 8).cpu().median(0).values
        res[k] = (sorted(ts)[2], float(st[3]) / (k // 64 - 4))
    us_per_tile = (res[5120][0] - res[2048][0]) / 48
    cyc = res[5120][1]
    print(f"{name} M = {m:5d} ({(m + 127) // 128 * 4:3d} blocks): {us_per_tile:.3f} us and {cyc:.0f} cycles per K tile -> {cyc / us_per_tile / 1e3:.2f} GHz", flush=True)
