"""A/B: weight-gradient GEMMs one by one (split-M + slab reduce) vs grouped (no split) for the shapes of one base encoder layer."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops_train as T

dev = "cuda:0"
M = 8000
shapes = [(2048, 512), (512, 2048), (1536, 512), (512, 512), (2048, 512), (512, 1024), (512, 1024), (2048, 512), (512, 2048)]      # (N, K): ff1 w1 w2, qkv, wo, mlp w1 w2, mrg, ff2 w1 w2
probs = []
for N, K in shapes:
    probs.append((torch.zeros(N, K, device=dev), torch.randn(M, N, device=dev).to(torch.bfloat16), torch.randn(M, K, device=dev).to(torch.bfloat16)))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def one_by_one(sel):
    for dw, dy, x in sel:
        T.gemm_tn_(dw, dy, x)


def grouped(sel):
    b = T.TnBatch(); b.MIN_TILES = 0
    for dw, dy, x in sel:
        b.add(dw, dy, x, dw.shape[0], None)
    b.flush()


print(f"all 9: one by one {timeit(lambda: one_by_one(probs)):.1f} us | grouped {timeit(lambda: grouped(probs)):.1f} us")
for k in (0, 1, 3):
    N, K = shapes[k]
    print(f"single {N}x{K}: split-M {timeit(lambda: one_by_one(probs[k:k+1])):.1f} us | grouped alone ({-(-N//256) * -(-K//128)} blocks) {timeit(lambda: grouped(probs[k:k+1])):.1f} us")
sel = [probs[0], probs[4], probs[7]]
print(f"three 2048x512: one by one {timeit(lambda: one_by_one(sel)):.1f} | grouped {timeit(lambda: grouped(sel)):.1f}")
