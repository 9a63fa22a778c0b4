"""torch.autograd bridge of the HIP training step, so that the reference's trainers (HF `Trainer` / `GradAwareTrainer`,
src/utilities/training_utils.py:93-115) can drive our drop-in models unchanged:  `loss = model(**batch).loss; loss.backward();
optimizer.step()`.

PyTorch is plumbing here: `HipStep.apply` runs forward AND backward of the whole model on the HIP trainer in its forward() — the
analytic backward needs no autograd graph — and hands the finished parameter gradients to autograd in backward(), scaled by the
incoming d(loss) (gradient accumulation / loss scaling).  torch's optimizer and DDP hooks then see ordinary `.grad`s.
The trainer's own all-reduce and AdamW are not used on this route (HF Trainer owns them); the native route is
`huggingface_asr_amd.train.EncoderCTCTrainer.train_step` / `train_aed.JointAEDTrainer.train_step`.

Zero-copy parameters (round 2).  On the first training forward the bridge ADOPTS the model's nn.Parameters: `p.data` becomes a view into the
trainer's flat fp32 master store (`alias_views("p")`: reshapes, row slices of the packed [Wq;Wk;Wv] / lm_head ⊕ blank matrices, the
permuted channels-last conv weight, the transposed GPT-2 Conv1D weights), so torch's optimizer updates the masters in place and a step costs one
bf16-mirror refresh instead of a state-dict import; the gradients handed to autograd are the matching views of the flat gradient store
(`alias_views("g")`), which autograd installs as `.grad` without a copy when `.grad` is None (`zero_grad(set_to_none=True)`, torch's and HF Trainer's
default).  Pieces whose reference layout is not a view of the packed one (the front end's `out` Linear) are copied in / out.  Gradient accumulation
(`.grad` still set when the next forward starts) takes a slower path that leaves the accumulated `.grad`s intact.  Trainers without `alias_views`
(BEST-RQ) keep the import / export path.
"""
from __future__ import annotations

import torch


class HipStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bridge, step_fn, *params):
        outs = bridge._forward(step_fn)
        ctx.bridge = bridge
        ctx.generation = bridge.generation      # the gradients live in the trainer's flat store, not in this node: only the LATEST forward's backward may read them
        ctx.mark_non_differentiable(*[v for k, v in outs.items() if k != "loss" and torch.is_tensor(v)])
        bridge.outputs = outs
        return outs["loss"].clone()

    @staticmethod
    def backward(ctx, gloss):
        b = ctx.bridge
        if ctx.generation != b.generation:
            raise RuntimeError("HIP training step: backward() of a forward that is no longer the latest one — the gradients of a step live in the trainer's flat store and "
                               "the next training forward overwrote them (run forward -> backward in order; two forwards before a backward are not supported)")
        if b.grads_taken:
            raise RuntimeError("HIP training step: backward() called twice on the same forward (the flat gradient store was already scaled and handed to autograd)")
        b.grads_taken = True
        grads = b._grads(gloss)
        return (None, None) + tuple(g if ctx.needs_input_grad[2 + i] else None for i, g in enumerate(grads))


class _Bridge:
    def __init__(self, model, trainer):
        self.model, self.trainer = model, trainer
        self.named = list(model.named_parameters())
        self.names = [n for n, _ in self.named]
        pset = set(self.names)
        self.extra = {k: v for k, v in model.state_dict().items() if k not in pset}      # buffers (e.g. rotary inv_freq)
        self.zero_copy = hasattr(trainer, "alias_views")
        self.adopted_ptrs = None
        self.outputs = None
        self._saved = None
        self.mirrors_fresh = False                          # set by optim.StoreAdamW: its step already wrote the bf16 mirrors and the transposes
        self.generation = 0                                 # +1 per training forward and per StoreAdamW step: the drop-in models key their eval-engine cache on it
        self.grads_taken = False                            # (an optimizer writing the masters through raw pointers bumps no tensor version)

    # ------------------------------------------------------------------ zero-copy parameters
    def _adopt(self):
        """make every aliasable nn.Parameter a view into the flat master store (values carried over); remember the rest for the copy path"""
        tr = self.trainer
        tr.load_state_dict({**self.extra, **{n: p.detach() for n, p in self.named}})       # one full import: masters, mirrors, non-parameter state
        views = tr.alias_views("p")
        self.copy_names = []
        with torch.no_grad():
            for n, p in self.named:
                v = views.get(n)
                if v is None or tuple(v.shape) != tuple(p.shape) or v.dtype != p.dtype or v.device != p.device:
                    self.copy_names.append(n)
                    continue
                p.data = v                                  # same values (just imported); from here on optimizer updates land in the flat store
        self.adopted_ptrs = [p.data_ptr() for _, p in self.named]
        self._versions = None

    def _still_adopted(self):
        return self.adopted_ptrs is not None and all(p.data_ptr() == q for (_, p), q in zip(self.named, self.adopted_ptrs))

    def _forward(self, step_fn):
        tr = self.trainer
        self.generation += 1
        self.grads_taken = False
        if hasattr(tr, "set_frozen"):                       # frozen sub-modules: their weight-gradient GEMMs are skipped, not computed and dropped
            tr.set_frozen({n for n, p in self.named if not p.requires_grad})
        if not self.zero_copy:
            tr.load_state_dict({**self.extra, **{n: p.detach() for n, p in self.named}})
            return step_fn(tr)
        if not self._still_adopted():                       # first call, or the parameters were re-allocated (model.to(...), .float(), ...)
            self._adopt()
        else:
            pd = dict(self.named)
            for n in self.copy_names:
                tr.import_piece(n, pd[n])
            if not self.mirrors_fresh:
                for st in tr.stores():                      # torch's optimizer wrote the fp32 masters in place: bf16 mirrors + K-major transposes follow
                    st.refresh_mirrors(cast=True)
            self.mirrors_fresh = False
        # gradient accumulation in progress?  (.grad of an adopted parameter still aliases the flat gradient store the step is about to overwrite)
        self._saved = None
        bases = {st.flat_g.untyped_storage().data_ptr() for st in tr.stores()}
        if any(p.grad is not None and p.grad.untyped_storage().data_ptr() in bases for _, p in self.named):
            self._saved = [st.flat_g.clone() for st in tr.stores()]
        return step_fn(tr)

    def _grads(self, gloss):
        tr = self.trainer
        if not self.zero_copy:
            gd = tr.grad_dict()
            return [None if gd.get(n) is None else gd[n] * gloss for n in self.names]
        stores = tr.stores()
        from . import ops_train as T
        for st in stores:
            T.scale_by_device_scalar_(st.flat_g, gloss)     # one kernel per store, a no-op launch for the usual d(loss) = 1 (no 2 x 516 MB pass, no host sync)
        copies = {n: tr.export_grad_piece(n) for n in self.copy_names}
        if self._saved is not None:                         # accumulation: hand out fresh tensors, put the accumulated sums back under the live .grad views
            fresh = [st.flat_g.clone() for st in stores]
            for st, old in zip(stores, self._saved):
                st.flat_g.copy_(old)
            self._saved = None
            keep = [st.flat_g for st in stores]
            try:
                for st, f in zip(stores, fresh):
                    st.flat_g = f
                views = tr.alias_views("g")
            finally:
                for st, k in zip(stores, keep):
                    st.flat_g = k
        else:
            views = tr.alias_views("g")
        return [copies[n] if n in copies else views.get(n) for n in self.names]


class LabelRangeCheck:
    """The reference raises when a label is >= vocab_size (`labels.max() >= vocab_size`, e_branchformer.py:461-462): a device -> host read per forward, i.e. a
    full synchronisation per training step — on this route the host then never runs ahead of the GPU and the card idles ~4 ms of every 29 ms step while the
    next step is being enqueued.  Training forwards therefore check asynchronously: the comparison is enqueued, its one-byte result copied to pinned host memory
    behind an event, and the flag is looked at when the NEXT forward starts (if its event has completed by then; otherwise at the one after).  A bad batch
    still raises ValueError naming the step, one forward later than the reference would; `flush()` (called by eval forwards) checks synchronously."""

    def __init__(self, limit: int, what: str):
        self.limit, self.what = int(limit), what
        self.host = None
        self.pending = []                                   # (event, slot, step)
        self.step = 0

    def _raise(self, step):
        raise ValueError(f"Label values must be <= {self.what}: {self.limit} (training forward #{step}; reported asynchronously)")

    def poll(self, block=False):
        keep = []
        for ev, slot, step in self.pending:
            if block:
                ev.synchronize()
            if ev.query():
                if bool(self.host[slot]):
                    self.pending = []
                    self._raise(step)
            else:
                keep.append((ev, slot, step))
        self.pending = keep

    def submit(self, labels: torch.Tensor):
        self.poll()
        if self.host is None:
            self.host = torch.zeros(8, dtype=torch.bool).pin_memory()
        if len(self.pending) >= 8:                          # never more than the ring holds in flight: wait for the oldest
            self.poll(block=True)
        slot = self.step % 8
        self.host[slot:slot + 1].copy_((labels.max() >= self.limit).reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, slot, self.step))
        self.step += 1

    def flush(self):
        self.poll(block=True)


def bridge_generation(model) -> int:
    """changes whenever a training forward or a StoreAdamW step may have changed the model's weights behind torch's version counters"""
    b = getattr(model, "_hip_bridge", None)
    return 0 if b is None else b.generation


def detach_state_dict_views(module, state_dict, prefix, local_metadata):
    """state-dict hook of the drop-in models.  After the bridge adopted the parameters they are views of ONE flat storage, some non-contiguous (conv2's permuted
    weight, the GPT-2 Conv1D `.t()` views): `save_pretrained` then sees every weight as a shared tensor and transformers' `_find_disjoint` / `_end_ptr` calls
    `.view(-1)` on the non-contiguous ones and raises.  Entries that are views of a larger storage (or non-contiguous) leave as private contiguous copies —
    what `state_dict()` of an un-adopted model hands out, values identical."""
    for k, t in list(state_dict.items()):
        if torch.is_tensor(t) and (not t.is_contiguous() or t.untyped_storage().nbytes() != t.numel() * t.element_size()):
            state_dict[k] = t.detach().clone(memory_format=torch.contiguous_format)
    return state_dict


def run_training_forward(model, trainer, step_fn):
    """model: nn.Module whose named_parameters() carry the reference names; trainer: the HIP trainer (load_state_dict / grad_dict, and for the zero-copy
    route alias_views / import_piece / export_grad_piece / stores); step_fn(trainer) -> dict of outputs incl. 'loss' (runs forward + backward on the HIP
    path, zeroing the gradient stores first).  Returns (loss with grad_fn, outputs)."""
    bridge = getattr(model, "_hip_bridge", None)
    if bridge is None or bridge.trainer is not trainer or len(bridge.named) != sum(1 for _ in model.parameters()):
        bridge = _Bridge(model, trainer)
        object.__setattr__(model, "_hip_bridge", bridge)
    loss = HipStep.apply(bridge, step_fn, *[p for _, p in bridge.named])
    return loss, bridge.outputs
