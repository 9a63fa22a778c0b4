"""Run-to-run reproducibility of the training step: 20 optimizer steps twice from the same state; reports the number of gradient / parameter elements whose BITS differ.
Since round 4 the answer is 0: every parameter-gradient reduction is a fixed-order sum (rounds 1-3 ended the bias / LayerNorm / depthwise-conv / embedding reductions in
float atomics: <= 3e-7 on the first gradient, amplified by AdamW).  With and without dropout (the masks are a pure function of (seed, step, site, element))."""
import sys, os, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from huggingface_asr_amd import shapes, synth
from huggingface_asr_amd.train import EncoderCTCTrainer
dev = "cuda:0"
for p in (0.0, 0.1):
    cfg = dict(shapes.SMALL, ctc_zero_infinity=True, ctc_loss_reduction="mean", hidden_dropout=p, activation_dropout=p, attention_dropout=p, final_dropout=p,
               feat_proj_dropout=0.0, csgu_conv_dropout=p, layerdrop=0.0, apply_spec_augment=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    feats = torch.from_numpy(synth.normal(5, "feats", (8, 400, 80), 1.0)).to(dev); lens = torch.full((8,), 400, dtype=torch.int32, device=dev)
    labels = torch.from_numpy(synth.labels(5, 8, 12, cfg["vocab_size"])).to(dev)
    runs = []
    for r in range(2):
        tr = EncoderCTCTrainer(cfg, dev, lr=1e-3, seed=3); tr.load_state_dict(sd)
        g1 = None
        ls = []
        for i in range(20):
            tr.store.zero_grad()
            o = tr.forward_backward(feats, lens, labels)
            if i == 0:
                g1 = tr.store.flat_g.clone()
            tr.optimizer_step()
            ls.append(float(o["loss"]))
        runs.append((ls, tr.store.flat_p.clone(), g1))
    dl = max(abs(a - b) for a, b in zip(runs[0][0], runs[1][0]))
    dg = float((runs[0][2] - runs[1][2]).abs().max()); gm = float(runs[0][2].abs().max())
    print(json.dumps({"dropout": p, "steps": 20, "first_loss_equal": runs[0][0][0] == runs[1][0][0], "max_loss_diff": dl, "first_step_grad_max_diff": dg, "grad_max": gm,
                      "first_step_grad_elements_differing": int((runs[0][2] != runs[1][2]).sum()), "param_elements_differing_after_20_steps": int((runs[0][1] != runs[1][1]).sum()),
                      "param_max_diff": float((runs[0][1] - runs[1][1]).abs().max())}))
