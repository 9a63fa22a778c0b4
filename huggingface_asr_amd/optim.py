"""torch.optim.Optimizer facade over the HIP trainers' fused optimizer step, for the HF-Trainer route.

On that route (autograd_bridge.py) the model's nn.Parameters are views of the trainer's flat fp32 master store and their `.grad`s are views of its flat
gradient store.  torch's own AdamW then walks ~560 tensors (multi-tensor launches), `clip_grad_norm_` makes two more passes over the gradients, and the bf16
mirrors the GEMMs read are re-cast in a pass of their own.  `StoreAdamW.step()` instead runs what the native route runs (train.py `optimizer_step`): one
sum-of-squares pass, the clip coefficient on the device, ONE AdamW kernel over the flat store that also writes the bf16 mirrors, one launch for the K-major
weight transposes.  Same update rule as torch.optim.AdamW (decoupled weight decay on matrices / conv taps only: the reference trainers' parameter grouping,
`tf:trainer.py get_decay_parameter_names`), fp32 moments.

Use with HF `Trainer` (the reference's `GradAwareTrainer` / `CustomSeq2SeqTrainer` take the same argument):

    opt = StoreAdamW(model, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2), eps=args.adam_epsilon,
                     weight_decay=args.weight_decay, max_grad_norm=args.max_grad_norm)
    trainer = Trainer(model=model, args=args, ..., optimizers=(opt, None))      # with --max_grad_norm 0 in `args`, so that HF does not clip a second time

There is no fallback: a model whose parameters are not (or no longer) adopted by the bridge raises.
"""
from __future__ import annotations

import torch


class StoreAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm))
        self.last_grad_norm = None
        self._pending_state = None          # moments loaded before the parameters were adopted (accelerate's prepare / a resume): applied by the first step()

    # ------------------------------------------------------------------ helpers
    def _bridge(self):
        bridge = getattr(self.model, "_hip_bridge", None)
        if bridge is None or not getattr(bridge, "zero_copy", False) or not bridge._still_adopted():
            raise RuntimeError("StoreAdamW: the model's parameters are not views of the HIP trainer's flat store (run a training forward first; "
                               "`model.to(...)` / `.float()` after it re-allocates them) — there is no fallback to a torch optimizer")
        return bridge

    def _adopted_bridge(self):
        bridge = getattr(self.model, "_hip_bridge", None)
        return bridge if (bridge is not None and getattr(bridge, "zero_copy", False) and bridge._still_adopted()) else None

    def _apply_pending(self, tr):
        if self._pending_state:
            stores = tr.stores()
            if len(stores) != len(self._pending_state) or any(st.flat_m.numel() != rec["m"].numel() for st, rec in zip(stores, self._pending_state)):
                raise RuntimeError("StoreAdamW: the loaded optimizer state does not match this model's parameter stores")
            for st, rec in zip(stores, self._pending_state):
                st.flat_m.copy_(rec["m"]); st.flat_v.copy_(rec["v"]); st.step_count = int(rec["step"])
        self._pending_state = None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("StoreAdamW does not re-evaluate the model: closures are not supported")
        bridge = self._bridge()
        tr = bridge.trainer
        self._apply_pending(tr)
        g = self.param_groups[0]
        named = dict(bridge.named)
        # pieces whose reference layout is not a view of the packed layout (the front end's `out` Linear): their `.grad` is a copy autograd owns — the trainer's
        # own slot still holds the same (scaled) gradient unless gradients were accumulated or edited in place, so it is refreshed from `.grad`
        for n in bridge.copy_names:
            p = named[n]
            if p.grad is not None:
                tr.import_grad_piece(n, p.grad)
        hp = tr.hp
        saved = {k: hp.get(k) for k in ("betas", "eps", "weight_decay", "max_grad_norm")}
        hp.update(betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"], max_grad_norm=g["max_grad_norm"] or 0.0)
        try:
            self.last_grad_norm = tr.optimizer_step(lr=g["lr"])                  # clip + AdamW + bf16 mirrors + transposes on the flat store(s)
        finally:
            hp.update(saved)
        for n in bridge.copy_names:                                             # masters -> the separately stored parameters (imported again by the next forward)
            named[n].copy_(tr.export_piece(n))
        bridge.mirrors_fresh = True                                             # the next training forward need not re-cast the masters
        bridge.generation += 1                                                  # the masters changed through raw pointers: no tensor version moved (eval-engine cache key)
        return None

    # optimizer state = the flat moment buffers (per store) + step counts: what a Trainer checkpoint saves and `--restart_from` restores.
    # Both work BEFORE the first training forward (the bridge adopts the parameters there): `accelerator.prepare(optimizer)` round-trips the state at construction and
    # HF Trainer's resume loads it before any step — an un-adopted optimizer has no moments yet (empty list), and a loaded state waits for the first step().
    def state_dict(self):
        bridge = self._adopted_bridge()
        if bridge is None:
            stores = [dict(m=r["m"].clone(), v=r["v"].clone(), step=r["step"]) for r in (self._pending_state or [])]
        else:
            self._apply_pending(bridge.trainer)
            stores = [dict(m=st.flat_m.clone(), v=st.flat_v.clone(), step=st.step_count) for st in bridge.trainer.stores()]
        return {"state": {"stores": stores},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} | {"params": list(range(len(g["params"])))} for g in self.param_groups]}

    def load_state_dict(self, sd):
        recs = list(sd.get("state", {}).get("stores", []))
        bridge = self._adopted_bridge()
        self._pending_state = [dict(m=r["m"].detach().clone(), v=r["v"].detach().clone(), step=int(r["step"])) for r in recs] or None
        if bridge is not None:
            self._apply_pending(bridge.trainer)
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})
