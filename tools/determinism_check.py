"""Bit-reproducibility of the whole base forward (B=16 x 10 s) over N runs; HFASR_BRANCH_OVERLAP=1 puts the two branches of a layer on two streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.engine import EBranchformerEngine
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean")
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
eng = EBranchformerEngine(cfg, dev); eng.load_state_dict(sd)
feats = torch.from_numpy(synth.normal(1, "feats", (32, 1000, 80), 1.0)).to(dev)
lens = torch.full((32,), 998, dtype=torch.int32, device=dev)
eng.branch_overlap, want = False, eng.branch_overlap
ref = eng.forward(feats, lens)["logits"].clone()            # single-stream result
eng.branch_overlap = want
bad, worst = 0, 0.0
for i in range(N):
    lg = eng.forward(feats, lens)["logits"]
    if not torch.equal(lg, ref):
        bad += 1; worst = max(worst, float((lg - ref).abs().max()))
torch.cuda.synchronize()
print(f"branch_overlap={eng.branch_overlap}: {bad}/{N} forwards differ from the single-stream result (max |diff| {worst:.4f})")
