"""GEMM duration by operand cache state: warm (same A / W re-read), cold (512 MB written elsewhere between launches), cold but A touched, cold but W
touched (= what a weight prefetch one op ahead would give), evicted by a 512 MB READ instead of a write.  Run under rocprofv3 --kernel-trace; durations are read from the trace."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from huggingface_asr_amd import ops
dev = "cuda:0"
REP = 10
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
fl32 = flush.view(torch.int32)
tiny = torch.zeros(64, device=dev)
mid = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
shapes = [(8000, 2048, 512), (8000, 512, 2048), (8000, 1536, 512), (8000, 512, 512), (8000, 512, 1024)]
for mode in (0, 1, 2, 3, 4, 5, 6):
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        af, wf = a.view(torch.int16), w.view(torch.int16)
        torch.cuda.synchronize()
        for _ in range(REP):
            if mode in (1, 2, 3): flush.zero_()
            if mode == 5: tiny.zero_()                    # a different (tiny) kernel in between, nothing evicted
            if mode == 6: mid.zero_()                     # 32 MB written in between (what ONE neighbouring kernel of the step does)
            if mode == 4: fl32.sum()                       # evict by READING 512 MB: no dirty lines left behind
            if mode == 2: af.sum()
            if mode == 3: wf.sum()
            ops.gemm(a, w, b, out=out)
        torch.cuda.synchronize()
print("done")
