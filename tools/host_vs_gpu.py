"""How far ahead of the GPU the Python driver of the training step runs: host time to ENQUEUE a step (no synchronisation) against the synchronised step time."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.train import EncoderCTCTrainer
dev = "cuda:0"
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
cfg = dict(shapes.BASE, ctc_zero_infinity=True, ctc_loss_reduction="mean", hidden_dropout=p, activation_dropout=p, attention_dropout=p, final_dropout=p,
           feat_proj_dropout=0.0, csgu_conv_dropout=p, layerdrop=0.0, apply_spec_augment=False)
sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
tr = EncoderCTCTrainer(cfg, dev, lr=2e-3, weight_decay=1e-6); tr.load_state_dict(sd)
B, T = 32, 1000
feats = torch.from_numpy(synth.normal(100, "feats", (B, T, 80), 1.0)).to(dev)
lens = torch.full((B,), T - 2, dtype=torch.int32, device=dev)
labels = torch.from_numpy(synth.labels(0, B, 40, cfg["vocab_size"], lo=5)).to(dev)
for _ in range(3):
    tr.train_step(feats, lens, labels)
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(feats, lens, labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print(json.dumps({"dropout": p, "host_enqueue_ms": round(sorted(host)[len(host) // 2], 2), "step_from_idle_ms": round(sorted(total)[len(total) // 2], 2)}))
