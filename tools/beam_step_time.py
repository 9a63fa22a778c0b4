"""mi_beam_step alone at the config-5 shape (B = 1, W = 5 and 1, V = 5001, step 20 of 40) on whatever build HFASR_HIP_LIB names: microseconds per launch, 50 back-to-back launches
behind a kernel that rewrites the logits (the state the token loop leaves them in).  tools/beam_step_phases.sh runs it over the BEAM_STOP builds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import _lib, ops
dev = "cuda:0"
L = _lib.lib()
name = os.environ.get("HFASR_HIP_LIB", "product").split("/")[-1]
for W in (5, 1):
    B, V, cur, max_length = 1, 5001, 20, 40
    Lmax = max_length + 1
    n = B * W
    g = torch.Generator().manual_seed(W)
    logits = (torch.randn(n, 5008, generator=g) * 2).to(dev)[:, :V]
    ctc = (torch.randn(n, V, generator=g) * 3 - 5).to(dev)
    ids = torch.full((n, Lmax), 5000, dtype=torch.long, device=dev); ids[:, 0] = 2
    ts = []
    for rep in range(30):
        bs = torch.zeros(n, device=dev); bs[1:] = -1.0 * torch.arange(1, n, device=dev)
        done, nfin = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        fs, fl = torch.zeros(B, W, device=dev), torch.zeros(B, W, dtype=torch.int32, device=dev)
        ft = torch.full((B, W, Lmax), 5000, dtype=torch.long, device=dev)
        nt, bi = torch.empty(n, dtype=torch.long, device=dev), torch.empty(n, dtype=torch.long, device=dev)
        logits.mul_(1.0001)                                     # another kernel wrote them last
        lse = ops.row_lse(logits)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.mi_beam_step(logits.data_ptr(), logits.stride(0), lse.data_ptr(), ctc.data_ptr(), 0.7, 0.3, 1, 5000, 1, B, W, V, cur, max_length, Lmax, float(cur), float(cur), 0,
                                  ids.data_ptr(), bs.data_ptr(), nt.data_ptr(), bi.data_ptr(), done.data_ptr(), nfin.data_ptr(), fs.data_ptr(), fl.data_ptr(), ft.data_ptr(), None, None, None,
                                  torch.cuda.current_stream().cuda_stream), "mi_beam_step")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{name} W={W}: {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})")
