"""soak of the fused token step (csrc/decoder_fused.hip) against the launch-per-op step: DeCRED_base-size decoder, W = 1 / 3 / 5 / 8 rows, 120 steps each with a beam re-ordering
every 7th step and ragged encoder lengths; every step's logits finite and within fp32-sum noise of the other form's, no drift over the decode."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import config5_model as M
from huggingface_asr_amd.decoder import JointAEDEngine
dev = "cuda:0"
sd = M.state_dict(0, structured=False)
eng = JointAEDEngine(M.ENC_CFG, M.DEC_CFG, M.JCFG, dev)
eng.load_state_dict(sd)
d, T2 = 512, 250
g = torch.Generator().manual_seed(0)
worst = 0.0
for W in (1, 3, 5, 8):
    enc_bf = (torch.randn(W * T2, d, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    key_len = torch.tensor([T2 - 17 * r for r in range(W)], dtype=torch.int32, device=dev)
    kvs = eng.dec.cross_kv(enc_bf)
    ca, cb = eng.dec.init_cache(W, 128), eng.dec.init_cache(W, 128)
    ids = torch.randint(5, 4990, (W, 121), generator=g).to(dev)
    for u in range(120):
        eng.dec._gcfg.step_form = 0
        a = eng.dec.step(ids[:, u:u + 1], ca, kvs, T2, key_len)
        eng.dec._gcfg.step_form = 1
        b = eng.dec.step(ids[:, u:u + 1], cb, kvs, T2, key_len)
        eng.dec._gcfg.step_form = 0
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), (W, u)
        err = float((a - b).abs().max())
        worst = max(worst, err)
        assert err < 3e-2, (W, u, err)
        if u % 7 == 6 and W > 1:
            perm = torch.randint(0, W, (W,), generator=g).to(dev)
            eng.dec.reorder_cache(ca, perm); eng.dec.reorder_cache(cb, perm)
    print(f"W = {W}: 120 steps, max |dlogit| fused vs launch-per-op {worst:.3e}", flush=True)
print("decode soak ok")
