"""A/B of the grouped weight-gradient launch (mi_gemm_tn_group_bf16) in situ — a chip-filling launch, not a lone block: tile / ring-depth forms on
(a) the small encoder's problems at BASELINE config 3's size (d = 256, M = 96 x 500 rows, 4 layers per launch) and (b) two base-size layers (d = 512, M = 8000).
usage: python tools/tn_group_depth.py [--set=0|1|2] [tile_k ...]      (tile_k as mi_gemm_tn_group_bf16 takes it; default 128 256 1128)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops_train as T

dev = "cuda:0"
SMALL = [(1024, 256), (256, 1024), (768, 256), (256, 256), (256, 256), (1024, 256), (256, 512), (256, 512), (1024, 256), (256, 1024)]
BASE = [(2048, 512), (512, 2048), (1536, 512), (512, 512), (512, 512), (2048, 512), (512, 1024), (512, 1024), (2048, 512), (512, 2048)]


def build(shapes, M, layers):
    out = []
    for _ in range(layers):
        for N, K in shapes:
            out.append((torch.zeros(N, K, device=dev), torch.randn(M, N, device=dev).to(torch.bfloat16), torch.randn(M, K, device=dev).to(torch.bfloat16), torch.zeros(N, device=dev)))
    return out


def run(probs, tk):
    b = T.TnBatch(); b.MIN_TILES = 0
    for dw, dy, x, db in probs:
        b.add(dw, dy, x, dw.shape[0], db)
    b.flush(tile_k=tk)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


args = [a for a in sys.argv[1:] if not a.startswith("--set=")]
only = [int(a[6:]) for a in sys.argv[1:] if a.startswith("--set=")]
tks = [int(a) for a in args] or [128, 256]
for si, (name, shapes, M, layers) in enumerate((("small x4 layers, M=48000", SMALL, 48000, 4), ("base x2 layers, M=8000", BASE, 8000, 2), ("base x1 layer, M=8000", BASE, 8000, 1))):
    if only and si not in only:
        continue
    probs = build(shapes, M, layers)
    gf = sum(2.0 * M * N * K for N, K in shapes) * layers / 1e9
    ref = None
    for tk in tks:
        for dw, _, _, db in probs:
            dw.zero_(); db.zero_()
        run(probs, tk)
        got = torch.cat([dw.flatten() for dw, _, _, _ in probs] + [db for _, _, _, db in probs]).clone()
        if ref is None:
            ref = got
        same = bool(torch.equal(got, ref))
        us = timeit(lambda: run(probs, tk))
        w = tk % 1000
        tiles = sum(-(-N // 256) * -(-K // w) for N, K in shapes) * layers
        print(f"{name}: tile_k {tk:5d}  {tiles:4d} tiles  {us:8.1f} us  {gf / us * 1e3:7.1f} TF  same bits as first form: {same}", flush=True)
