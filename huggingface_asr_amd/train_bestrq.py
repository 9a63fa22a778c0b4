"""BEST-RQ pre-training on the HIP path: forward, loss and training step of the reference's
`BestRQEBranchformerForPreTraining` (src/models/bestrq.py:66-152, 179-190; https://arxiv.org/abs/2202.01855).

    targets  = argmin_c || CB[c] - normalize(stack4(features) · P) ||        (frozen random projection + codebook, fp32; mi_rpq_targets;
                                                                              the reference normalises over the BOOKS axis — kept)
    input    = feature projection with the masked frames replaced by N(0, 0.1) noise          (mi_mask_noise_f32)
    logits_k = classifier_k(encoder(input)),   loss = sum_{masked frames} CE(logits_k, targets_k) / num_books

The encoder is `train.EncoderCTCTrainer(head=False)`; the classifiers and their cross-entropy hook into its backward at the encoder
output exactly like the attention decoder of the joint model does.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from . import ops_train as T
from .train import BF16, F32, EncoderCTCTrainer, GradSync, ParamStore, Spec


class BestRQTrainer:
    def __init__(self, cfg: dict, device="cuda:0", *, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 dp_sync=True, seed=0, noise_std=0.1):
        c = self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.C, self.cd = int(c.get("best_rq_codebook_size", 8192)), int(c.get("best_rq_codebook_dim", 16))
        self.nb, self.in_dim = int(c.get("best_rq_num_books", 1)), int(c.get("best_rq_in_dim", 320))
        self.noise_std = float(noise_std)
        enc_cfg = dict(c, apply_spec_augment=False)          # BestRQMask replaces the SpecAugment hook (bestrq.py:84-97)
        self.enc = EncoderCTCTrainer(enc_cfg, device, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                     group=group, dp_sync=dp_sync, seed=seed, head=False)
        d = c["hidden_size"]
        specs = []
        for k in range(self.nb):
            specs += [Spec(f"cls{k}_w", (self.C, d), True, True), Spec(f"cls{k}_b", (self.C,), False, False)]
        self.store = ParamStore(specs, self.device)
        self.sync = GradSync(self.store.flat_g, group, enabled=dp_sync)
        self.hp = self.enc.hp
        self.P = self.CB = None

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: dict):
        dev = self.device
        self.enc.load_state_dict({k: v for k, v in sd.items() if k.startswith("wav2vec2.")})
        for k in range(self.nb):
            self.store.p(f"cls{k}_w").copy_(sd[f"classifiers.{k}.weight"].detach().to(dev, F32))
            self.store.p(f"cls{k}_b").copy_(sd[f"classifiers.{k}.bias"].detach().to(dev, F32))
        self.store.refresh_mirrors(cast=True)
        self.P = sd["rpq.P"].detach().to(dev, F32).contiguous()        # (books, in_dim, cd) frozen buffers
        self.CB = sd["rpq.CB"].detach().to(dev, F32).contiguous()      # (books, C, cd)

    def _export(self, enc_view, view):
        out = dict(enc_view())
        for k in range(self.nb):
            out[f"classifiers.{k}.weight"] = view(f"cls{k}_w").clone()
            out[f"classifiers.{k}.bias"] = view(f"cls{k}_b").clone()
        return out

    def state_dict(self):
        out = self._export(self.enc.state_dict, self.store.p)
        out["rpq.P"], out["rpq.CB"] = self.P.clone(), self.CB.clone()
        return out

    def grad_dict(self):
        return self._export(self.enc.grad_dict, self.store.g)

    # ------------------------------------------------------------------ pieces
    def targets(self, feats, mask_time_indices):
        """RandomProjectionQuantizer on input_values.view(B, T', -1) (bestrq.py:127): (B, books, T') int64, -100 where not masked."""
        B, T2 = mask_time_indices.shape
        x = feats.to(F32).contiguous().view(B * T2, -1)
        if x.shape[1] != self.in_dim:
            raise ValueError(f"stacked feature size {x.shape[1]} != best_rq_in_dim {self.in_dim}")
        out = torch.empty((self.nb, B * T2), dtype=torch.long, device=self.device)
        _lib.check(_lib.lib().mi_rpq_targets(x.data_ptr(), x.stride(0), self.P.data_ptr(), self.CB.data_ptr(), out.data_ptr(), B * T2,
                                             self.in_dim, self.cd, self.C, self.nb, torch.cuda.current_stream().cuda_stream), "mi_rpq_targets")
        tg = out.view(self.nb, B, T2).transpose(0, 1).contiguous()
        return tg.masked_fill(~mask_time_indices[:, None, :], -100)

    def forward_backward(self, feats, feat_lengths, mask_time_indices, *, backward=True):
        """feats (B,T,F) f32; mask_time_indices (B,T') bool.  -> dict(loss, last_hidden (B,T',d), targets (B,books,T'), logits list)"""
        dev = self.device
        mask_time_indices = mask_time_indices.to(dev, torch.bool)
        B, T2 = mask_time_indices.shape
        tg = self.targets(feats, mask_time_indices)
        tm = mask_time_indices.reshape(-1).to(torch.uint8).contiguous()
        st = self.store
        P, G, W, WT = st.p, st.g, st.bf, st.bfT
        d = self.cfg["hidden_size"]
        Cp = T.pad64(self.C)
        out = {}
        world = self.sync.world

        def heads(last_hidden, want_grad):
            M = last_hidden.shape[0]
            hb = ops.cast_bf16(last_hidden)
            loss = None
            dh = torch.zeros((M, d), device=dev, dtype=F32) if want_grad else None
            logits = []
            for k in range(self.nb):
                buf = torch.empty((B, T2, Cp), device=dev, dtype=F32)
                ops.gemm(hb, W(f"cls{k}_w"), P(f"cls{k}_b"), out=buf.view(M, Cp))
                lg = buf[..., :self.C]
                lab = tg[:, k].contiguous()
                acc = torch.zeros((2,), device=dev, dtype=F32)
                rows = torch.empty((B * T2,), device=dev, dtype=F32)
                _lib.check(_lib.lib().mi_ce_label_smoothing(lg.data_ptr(), lg.stride(1), lab.data_ptr(), B, T2, 0, self.C, 0.0, acc.data_ptr(), rows.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream), "mi_ce_label_smoothing")
                lk = acc[0] / self.nb                           # reduction="sum", then / num_books (bestrq.py:141-142)
                loss = lk if loss is None else loss + lk
                logits.append(lg)
                if want_grad:
                    one = torch.tensor([0.0, 1.0], device=dev)  # weight / count = 1 / (books * world): gradient of the SUM
                    dl = T.ce_label_smoothing_bwd(lg, lab, one, shift=0, eps=0.0, weight=1.0 / (self.nb * world), ldo=Cp)
                    ops.gemm(dl, WT(f"cls{k}_w"), out=dh, resid=dh, alpha=1.0)
                    T.gemm_tn_(G(f"cls{k}_w"), dl, hb, n_store=self.C, db=G(f"cls{k}_b"))
            out.update(loss=loss, logits=logits)
            return dh

        if backward:
            def hook(last_hidden, outer_len):
                dh = heads(last_hidden, True)
                self.sync.launch(0, st.n)
                return dh
            eo = self.enc.forward_backward(feats, feat_lengths, None, extra_hidden_grad=hook, noise_mask=(tm, self.noise_std))
        else:
            eo = self.enc.forward_backward(feats, feat_lengths, None, backward=False, train_mode=True, noise_mask=(tm, self.noise_std))
            heads(eo["last_hidden"].reshape(B * T2, d), False)
        out.update(last_hidden=eo["last_hidden"], targets=tg)
        return out

    def optimizer_step(self, lr=None):
        hp = self.hp
        self.enc.sync.wait(); self.sync.wait()
        sc = self.enc._scal
        sc.zero_()
        T.sumsq_(sc[0:1], self.enc.store.flat_g)
        T.sumsq_(sc[0:1], self.store.flat_g)
        T.clip_coef(sc[0:1], hp["max_grad_norm"] if hp["max_grad_norm"] else 0.0, sc[1:4], hp.get("grad_norm_skip", 0.0))
        for s_ in (self.enc.store, self.store):
            s_.step_count += 1
            T.adamw_step_(s_.flat_p, s_.flat_g, s_.flat_m, s_.flat_v, s_.decay, lr=hp["lr"] if lr is None else lr, betas=hp["betas"], eps=hp["eps"],
                          weight_decay=hp["weight_decay"], step=s_.step_count, norm_coef=sc[1:4], mirror=s_.flat_bf)
            s_.refresh_mirrors(cast=False)
        return sc[1]

    def train_step(self, feats, feat_lengths, mask_time_indices, lr=None):
        self.enc.store.zero_grad(); self.store.zero_grad()
        out = self.forward_backward(feats, feat_lengths, mask_time_indices)
        out["grad_norm"] = self.optimizer_step(lr)
        return out
