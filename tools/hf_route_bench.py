#!/usr/bin/env python
"""Cost of the HF-Trainer route against the native trainer: the same base encoder + CTC training step, 32 x 10 s,
  native:  EncoderCTCTrainer.train_step                       (flat store, fused AdamW + clip on the device)
  hf:      what GradAwareTrainer / HF Trainer run per step     (training_utils.py:93-115): model.train(); loss = model(**batch).loss; loss.backward();
           clip_grad_norm_(model.parameters(), 1.0); torch.optim.AdamW.step(); zero_grad()
    python tools/hf_route_bench.py [--steps 10] [--size base]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.configuration_ebranchformer import Wav2Vec2EBranchformerConfig
from huggingface_asr_amd.modeling_ebranchformer import Wav2Vec2EBranchformerForCTC
from huggingface_asr_amd.train import EncoderCTCTrainer

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--size", default="base")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--fused", action="store_true", help="torch.optim.AdamW(fused=True) (HF: --optim adamw_torch_fused) instead of the foreach implementation")
ap.add_argument("--store", action="store_true", help="huggingface_asr_amd.optim.StoreAdamW (Trainer(optimizers=(opt, None)), --max_grad_norm 0): the native fused clip + AdamW step on the flat store")
a = ap.parse_args()
dev = "cuda:0"
base = dict({"base": shapes.BASE, "small": shapes.SMALL, "tiny": shapes.TINY}[a.size])
nodrop = dict(hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, final_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, apply_spec_augment=False)
cfg = dict(base, ctc_zero_infinity=True, ctc_loss_reduction="mean", csgu_conv_dropout=0.0, **nodrop)
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
B, T, U = a.batch, 1000, 40
feats = torch.from_numpy(synth.normal(100, "feats", (B, T, 80), 1.0)).to(dev)
am = torch.zeros(B, T, dtype=torch.long, device=dev); am[:, :998] = 1
lens = am.sum(-1).to(torch.int32)
labels = torch.from_numpy(synth.labels(0, B, U, cfg["vocab_size"], lo=5)).to(dev)


def timed(fn):
    for _ in range(a.warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


tr = EncoderCTCTrainer(cfg, dev, lr=1e-4, weight_decay=1e-6)
tr.load_state_dict(sd)
native = timed(lambda: tr.train_step(feats, lens, labels))
del tr; torch.cuda.empty_cache()

hb = dict(base); hb.pop("num_fbanks", None)
model = Wav2Vec2EBranchformerForCTC(Wav2Vec2EBranchformerConfig(**hb, ctc_zero_infinity=True, ctc_loss_reduction="mean", ebranchformer_conv_dropout=0.0, **nodrop))
model.load_state_dict(sd, strict=False)
model.to(dev).train()
if a.store:
    from huggingface_asr_amd.optim import StoreAdamW
    opt = StoreAdamW(model, lr=1e-4, weight_decay=1e-6, max_grad_norm=1.0)
else:
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-6, fused=True if a.fused else None)   # None: torch picks its foreach implementation
parts = {}


def hf_step():
    t0 = time.perf_counter()
    out = model(input_values=feats, attention_mask=am, labels=labels)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    if not a.store:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step(); opt.zero_grad()
    t3 = time.perf_counter()
    for k, v in (("host_forward_ms", t1 - t0), ("host_backward_ms", t2 - t1), ("host_clip_adamw_ms", t3 - t2)):
        parts[k] = parts.get(k, 0.0) + v * 1e3


hf = timed(hf_step)
n = a.steps + a.warmup
print(json.dumps({"native_ms_per_step": round(native, 2), "hf_route_ms_per_step": round(hf, 2), "overhead_pct": round(100 * (hf / native - 1), 1),
                  "host_side_ms": {k: round(v / n, 2) for k, v in parts.items()}, "size": a.size, "batch": B, "optimizer": "StoreAdamW" if a.store else "torch fused" if a.fused else "torch foreach"}))
