"""torch.autograd bridge of the HIP training step, so that the reference's trainers (HF `Trainer` / `GradAwareTrainer`,
src/utilities/training_utils.py:93-115) can drive our drop-in models unchanged:  `loss = model(**batch).loss; loss.backward();
optimizer.step()`.

PyTorch is plumbing here: `HipStep.apply` runs forward AND backward of the whole model on the HIP trainer in its forward() — the
analytic backward needs no autograd graph — and hands the finished parameter gradients to autograd in backward(), scaled by the
incoming d(loss) (gradient accumulation / loss scaling).  torch's optimizer and DDP hooks then see ordinary `.grad`s.
The trainer's own all-reduce and AdamW are not used on this route (HF Trainer owns them); the native route is
`huggingface_asr_amd.train.EncoderCTCTrainer.train_step` / `train_aed.JointAEDTrainer.train_step`.
"""
from __future__ import annotations

import torch


class HipStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, names, *params):
        """runner(state_dict) -> (dict of output tensors, grad_dict in reference names).  Returns the loss (graph-connected to `params`)."""
        sd = {n: p.detach() for n, p in zip(names, params)}
        outs, grads = runner(sd)
        ctx.grads = [grads.get(n) for n in names]
        ctx.mark_non_differentiable(*[v for k, v in outs.items() if k != "loss" and torch.is_tensor(v)])
        runner.outputs = outs
        return outs["loss"].clone()

    @staticmethod
    def backward(ctx, gloss):
        out = [None, None]
        for i, g in enumerate(ctx.grads):                   # frozen parameters (requires_grad False: freeze_encoder etc.) get no gradient
            out.append(None if (g is None or not ctx.needs_input_grad[2 + i]) else g * gloss)
        return tuple(out)


def run_training_forward(model, trainer, step_fn):
    """model: nn.Module whose named_parameters() carry the reference names; trainer: object with load_state_dict / grad_dict / zero grads;
    step_fn(trainer) -> dict of outputs incl. 'loss' (runs forward + backward on the HIP path).  Returns (loss with grad_fn, outputs)."""
    named = [(n, p) for n, p in model.named_parameters()]
    names = [n for n, _ in named]
    extra = {k: v for k, v in model.state_dict().items() if k not in set(names)}      # buffers (e.g. rotary inv_freq)

    class _Runner:
        outputs = None

        def __call__(self, sd):
            trainer.load_state_dict({**extra, **sd})
            outs = step_fn(trainer)
            grads = trainer.grad_dict()
            return outs, grads

    if hasattr(trainer, "set_frozen"):                  # frozen sub-modules: their weight-gradient GEMMs are skipped, not computed and dropped
        trainer.set_frozen({n for n, p in named if not p.requires_grad})
    runner = _Runner()
    loss = HipStep.apply(runner, names, *[p for _, p in named])
    return loss, runner.outputs
