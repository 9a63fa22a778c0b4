"""A/B of the depthwise-conv kernels (CSGU and merge) at the bench shape: one-tile-per-block form vs the persistent prefetching form (DWP = its grid size), graph replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
B, T, C = 32, 250, 1024
M = B * T
u = torch.randn(M, 2 * C, device=dev).to(torch.bfloat16)
g, be = torch.randn(C, device=dev), torch.randn(C, device=dev)
w, bias = torch.randn(C, 31, device=dev) * 0.2, torch.randn(C, device=dev)
m = torch.randn(M, C, device=dev).to(torch.bfloat16)


def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50


ref = None
for grid in [0] + [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,768,1024").split(",")]:
    os.environ["DWP"] = str(grid)
    st = ops.row_stats(u[:, C:])
    a = ops.csgu(u, g, be, w, bias, B, T)
    b = ops.dwconv_residual(m, w, bias, B, T)
    torch.cuda.synchronize()
    if ref is None: ref = (a.clone(), b.clone())
    same = bool(torch.equal(a, ref[0]) and torch.equal(b, ref[1]))
    t1 = timeit(lambda: ops.csgu(u, g, be, w, bias, B, T))        # includes row_stats (~5 us)
    t2 = timeit(lambda: ops.dwconv_residual(m, w, bias, B, T))
    print(f"grid {grid:5d}: csgu(+row_stats) {t1:6.2f} us  merge {t2:6.2f} us  bit-identical {same}", flush=True)
