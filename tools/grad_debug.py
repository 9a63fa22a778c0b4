"""debug: per-tensor gradient error of the HIP trainer vs a golden / oracle-autograd reference"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch, torch.nn.functional as F
from helpers import case_inputs, load_golden
from huggingface_asr_amd import shapes
from huggingface_asr_amd.train import EncoderCTCTrainer
name = sys.argv[1] if len(sys.argv) > 1 else "grads_tiny_rel"
extra = {"position_embeddings_type": "rotary"} if "rotary" in name else {}
g = load_golden(name)
cfg = dict(shapes.TINY, ctc_zero_infinity=True, ctc_loss_reduction="mean", **extra)
sd, x, am, lab = case_inputs(g, cfg)
ND = dict(hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, final_dropout=0.0, feat_proj_dropout=0.0, csgu_conv_dropout=0.0, apply_spec_augment=False, layerdrop=0.0)
tr = EncoderCTCTrainer(dict(cfg, **ND), "cuda:0"); tr.load_state_dict(sd)
tr.store.zero_grad()
out = tr.forward_backward(x.cuda(), am.sum(-1).cuda(), lab.cuda())
print("loss", float(out["loss"]), float(g["loss"]))
grads = tr.grad_dict()
for k in g.files:
    if not k.startswith("grad:"): continue
    want = torch.from_numpy(g[k]).reshape(-1); got = grads[k[5:]].float().cpu().reshape(-1)
    nw = float(want.norm()); err = float((got - want).norm())
    print(f"{err/max(nw,1e-12):10.4f} cos {float(F.cosine_similarity(got, want, dim=0)):8.5f} |ref| {nw:10.4g}  {k[5:]}")
