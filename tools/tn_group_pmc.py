"""Two base-size layers' weight-gradient GEMMs as one grouped launch (256 x 256 tiles), a few times: the target of `rocprofv3 --pmc ... -- python3 tools/tn_group_pmc.py`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import ops_train as T

dev = "cuda:0"
M = 8000
shapes = [(2048, 512), (512, 2048), (1536, 512), (512, 512), (2048, 512), (512, 1024), (512, 1024), (2048, 512), (512, 2048)] * 2
probs = [(torch.zeros(N, K, device=dev), torch.randn(M, N, device=dev).to(torch.bfloat16), torch.randn(M, K, device=dev).to(torch.bfloat16)) for N, K in shapes]
tile_k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for _ in range(5):
    b = T.TnBatch()
    for dw, dy, x in probs:
        b.add(dw, dy, x, dw.shape[0], None)
    b.flush(tile_k=tile_k)
torch.cuda.synchronize()
