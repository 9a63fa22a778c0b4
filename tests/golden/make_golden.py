"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (CPU, this container).

Run once here (the reference never travels to the GPU box; only the .npz fixtures do):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Harness-side shims for the transformers 4.39 -> 5.15 skew are those listed in SURVEY.md §8c:
attn_implementation="eager", and setattr of num_fbanks / conv_padding / context_awareness_type on
the config.  Reference files are untouched.  Weights come from huggingface_asr_amd.synth (seeded,
counter-based), so fixtures store only seeds + outputs (+ a weight checksum).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

from huggingface_asr_amd import synth  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

TINY = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
            conv_dim=[32, 32], conv_kernel=[3, 3], conv_stride=[2, 2], vocab_size=50)
BASE = dict(hidden_size=512, num_hidden_layers=16, num_attention_heads=4, intermediate_size=2048,
            conv_dim=[256, 256], conv_kernel=[3, 3], conv_stride=[2, 2], vocab_size=5000)
SMALL = dict(hidden_size=256, num_hidden_layers=12, num_attention_heads=4, intermediate_size=1024,
             conv_dim=[256, 256], conv_kernel=[3, 3], conv_stride=[2, 2], vocab_size=5000)


def build_reference(cfg_kwargs, **extra):
    from models.encoders.e_branchformer import Wav2Vec2EBranchformerConfig, Wav2Vec2EBranchformerForCTC

    cfg = Wav2Vec2EBranchformerConfig(**cfg_kwargs, attn_implementation="eager", ctc_zero_infinity=True,
                                      ctc_loss_reduction="mean", layerdrop=0.0, **extra)
    # `context_awareness_type` is a CustomFEConfig argument (extractors.py:16-20) that the multiple-inheritance chain no longer reaches: shim (2) sets it
    for k, v in dict(num_fbanks=80, conv_padding=[1, 1], context_awareness_type=extra.get("context_awareness_type")).items():
        setattr(cfg, k, v)
    model = Wav2Vec2EBranchformerForCTC(cfg).eval()
    return cfg, model


def load_seeded(model, seed):
    # parameters only: buffers (e.g. the rotary `embed_positions.inv_freq`) keep their computed values
    new = {k: torch.from_numpy(synth.init_param(seed, k, tuple(v.shape))) for k, v in model.named_parameters()}
    missing, unexpected = model.load_state_dict(new, strict=False)
    assert not unexpected and all(("inv_freq" in m or ".attn.bias" in m or "masked_bias" in m or m.startswith("rpq.")) for m in missing), (missing, unexpected)
    return float(sum(v.double().sum() for v in new.values()))


def synth_feats(seed, B, T, lengths):
    x = synth.normal(seed, "feats", (B, T, 80), 1.0)
    am = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lengths):
        am[b, :n] = 1
        x[b, n:] = 0.0
    return x, am


def synth_labels(seed, B, U, vocab, tgt_lens):
    lab = synth.labels(seed, B, U, vocab, lo=0)
    for b, n in enumerate(tgt_lens):
        lab[b, n:] = -100
    return lab


def run_encoder_case(name, cfg_kwargs, seed, B, T, lengths, U, tgt_lens, full=True, with_bf16=False, **extra):
    cfg, model = build_reference(cfg_kwargs, **extra)
    wsum = load_seeded(model, seed)
    x, am = synth_feats(seed, B, T, lengths)
    lab = synth_labels(seed, B, U, cfg.vocab_size, tgt_lens)
    xt, amt, labt = torch.from_numpy(x), torch.from_numpy(am), torch.from_numpy(lab)
    with torch.no_grad():
        out = model(xt.clone(), attention_mask=amt, labels=labt, output_hidden_states=True)
    logits = out.logits.float().numpy()
    hs = [h.float().numpy() for h in out.hidden_states]
    rec = dict(seed=seed, weight_sum=wsum, lengths=np.array(lengths), tgt_lens=np.array(tgt_lens),
               shape=np.array([B, T, U]), loss=float(out.loss),
               outer_lens=model._get_feat_extract_output_lengths(amt.sum(-1)).numpy(),
               inner_lens=model.wav2vec2._get_feat_extract_output_lengths(amt.sum(-1)).numpy())
    if full:
        rec["logits"] = logits
        rec["last_hidden"] = hs[-1]
        for i, h in enumerate(hs[:-1]):
            rec[f"layer_in_{i}"] = h
    else:
        rec["logits_slice"] = logits[:, ::25, :64].copy()
        rec["logits_blank"] = logits[:, :, -1].copy()
        rec["logits_absmax"] = float(np.abs(logits).max())
        rec["logits_std"] = float(logits.std())
        rec["layer_norms"] = np.array([float(np.linalg.norm(h)) for h in hs])
        rec["last_hidden_slice"] = hs[-1][:, ::25, :64].copy()
    if with_bf16:
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            ob = model(xt.clone(), attention_mask=amt, labels=labt)
        lb = ob.logits.float().numpy()
        rec["bf16_loss"] = float(ob.loss)
        rec["bf16_max_dlogit"] = float(np.abs(lb - logits).max())
        rec["bf16_mean_dlogit"] = float(np.abs(lb - logits).mean())
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "loss", rec["loss"], "logit std", float(logits.std()), {k: v for k, v in rec.items() if k.startswith("bf16")})


# the CSGU dropout's constructor argument is `ebranchformer_conv_dropout` (stored as config.csgu_conv_dropout, e_branchformer.py:44,57)
NO_DROPOUT = dict(hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, final_dropout=0.0, feat_proj_dropout=0.0,
                  ebranchformer_conv_dropout=0.0, apply_spec_augment=False)


def run_grad_case(name, cfg_kwargs, seed, B, T, lengths, U, tgt_lens, np_seed=None, **extra):
    """training-mode forward + backward of the reference model (dropouts 0; SpecAugment only when `extra` turns it on, its numpy RNG seeded
    with np_seed right before the forward): loss and every parameter gradient."""
    cfg, model = build_reference(cfg_kwargs, **{**NO_DROPOUT, **extra})
    model.train()
    wsum = load_seeded(model, seed)
    x, am = synth_feats(seed, B, T, lengths)
    lab = synth_labels(seed, B, U, cfg.vocab_size, tgt_lens)
    if np_seed is not None:
        np.random.seed(np_seed)
    out = model(torch.from_numpy(x), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab))
    out.loss.backward()
    rec = dict(seed=seed, weight_sum=wsum, lengths=np.array(lengths), tgt_lens=np.array(tgt_lens), shape=np.array([B, T, U]), loss=float(out.loss),
               np_seed=-1 if np_seed is None else np_seed)
    assert cfg.csgu_conv_dropout == 0.0
    g32 = {k: v.grad.float().clone() for k, v in model.named_parameters() if v.grad is not None}
    for k, v in g32.items():
        rec["grad:" + k] = v.numpy()
    # the reference's own bf16-autocast backward vs its fp32 backward: the yard-stick for the HIP path's bf16 gradients
    model.zero_grad()
    if np_seed is not None:
        np.random.seed(np_seed)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ob = model(torch.from_numpy(x), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab))
    ob.loss.float().backward()
    rec["bf16_loss"] = float(ob.loss)
    gaps = []
    for k, v in model.named_parameters():
        if v.grad is not None and float(g32[k].norm()) > 1e-6:
            gaps.append(float((v.grad.float() - g32[k]).norm() / g32[k].norm()))
    rec["bf16_grad_relerr_max"] = max(gaps)
    rec["bf16_grad_relerr_mean"] = float(np.mean(gaps))
    print("  autocast-vs-fp32 gradient gap: max", max(gaps), "mean", float(np.mean(gaps)))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "loss", rec["loss"], "n grads", sum(k.startswith("grad:") for k in rec))


def run_grad_case_strided(name, cfg_kwargs, seed, B, T, lengths, U, tgt_lens, **extra):
    """run_grad_case for a FULL-SIZE model (base: 129 M parameters, head size 128 = the LDS-staged attention and its backward): the loss, and for every parameter
    gradient its L2 norm and a strided sample of 256 elements — fp32 and under the reference stack's own bf16 autocast (the yard-stick)."""
    cfg, model = build_reference(cfg_kwargs, **{**NO_DROPOUT, **extra})
    model.train()
    wsum = load_seeded(model, seed)
    x, am = synth_feats(seed, B, T, lengths)
    lab = synth_labels(seed, B, U, cfg.vocab_size, tgt_lens)
    out = model(torch.from_numpy(x), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab))
    out.loss.backward()
    rec = dict(seed=seed, weight_sum=wsum, lengths=np.array(lengths), tgt_lens=np.array(tgt_lens), shape=np.array([B, T, U]), loss=float(out.loss))
    g32 = {k: v.grad.float().clone() for k, v in model.named_parameters() if v.grad is not None}

    def sample(t):
        f = t.reshape(-1)
        return f[:: max(1, f.numel() // 256) | 1][:256].numpy()          # odd stride: the sample walks across the columns instead of sitting in one
    for k, v in g32.items():
        rec["norm:" + k] = np.float64(v.double().norm())
        rec["samp:" + k] = sample(v)
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ob = model(torch.from_numpy(x), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab))
    ob.loss.float().backward()
    rec["bf16_loss"] = float(ob.loss)
    gaps, sgaps = [], []
    for k, v in model.named_parameters():
        if v.grad is not None and float(g32[k].norm()) > 1e-6:
            gaps.append(float((v.grad.float() - g32[k]).norm() / g32[k].norm()))
            a, b = sample(v.grad.float()), sample(g32[k])
            if np.linalg.norm(b) > 1e-9:
                sgaps.append(float(np.linalg.norm(a - b) / np.linalg.norm(b)))
    rec["bf16_grad_relerr_max"], rec["bf16_grad_relerr_mean"] = max(gaps), float(np.mean(gaps))
    rec["bf16_samp_relerr_max"], rec["bf16_samp_relerr_mean"] = max(sgaps), float(np.mean(sgaps))
    print("  autocast-vs-fp32 gradient gap: whole tensors max", max(gaps), "mean", float(np.mean(gaps)), "| strided samples max", max(sgaps), "mean", float(np.mean(sgaps)))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(name, "loss", rec["loss"], "n grads", len(g32))


def run_fbank_cases():
    from utilities.feature_extractors import CustomFeatureExtractor

    fe = CustomFeatureExtractor(feature_size=80, norm_type="utterance")
    fe_raw = CustomFeatureExtractor(feature_size=80, norm_type="utterance", do_ceptral_normalize=False)
    n = 16000 * 2
    t = np.arange(n) / 16000.0
    waves = {
        "sweep": (0.5 * np.sin(2 * np.pi * (100 + 1800 * t) * t)).astype(np.float32),
        "noise": synth.normal(0, "fbank_noise", (n,), 0.1),
        "silence_padded": np.concatenate([np.zeros(3000, np.float32), synth.normal(1, "fbank_sp", (n - 8000,), 0.05),
                                          np.zeros(5000, np.float32)]),
    }
    rec = {}
    for k, w in waves.items():
        raw = fe_raw(w, sampling_rate=16000, padding=False, return_attention_mask=False, return_tensors="np")["input_features"][0]
        nrm = fe(w, sampling_rate=16000, padding=False, return_attention_mask=False, return_tensors="np")["input_features"][0]
        rec[f"{k}_wave"] = w
        rec[f"{k}_raw"] = np.asarray(raw, np.float32)
        rec[f"{k}_cmvn"] = np.asarray(nrm, np.float32)
        print("fbank", k, raw.shape, float(np.mean(nrm)), float(np.std(nrm)))
    # global-stat normalisation
    means = np.linspace(5.0, 9.0, 80).astype(np.float32)
    stds = np.linspace(2.0, 4.0, 80).astype(np.float32)
    feg = CustomFeatureExtractor(feature_size=80, norm_type="global", global_means=means.tolist(), global_stds=stds.tolist())
    rec["noise_global"] = np.asarray(feg(waves["noise"], sampling_rate=16000, padding=False, return_attention_mask=False,
                                         return_tensors="np")["input_features"][0], np.float32)
    rec["global_means"], rec["global_stds"] = means, stds
    np.savez_compressed(os.path.join(HERE, "fbank.npz"), **rec)


def _stub_absent_third_party():
    """`utilities.callbacks` / `utilities.collators` import the reference's whole glue layer (wandb, jiwer, librosa, torchaudio, ... are absent from this image);
    nothing of those is executed by the transform chain or the collator.  Harness side, after transformers is loaded (it probes the same names itself)."""
    import importlib.abc
    import importlib.machinery
    import types
    import transformers.generation.utils as gu
    for n in ("BeamSearchOutput", "GreedySearchOutput", "SampleOutput", "BeamSampleOutput"):          # names of transformers 4.39 the glue imports
        if not hasattr(gu, n):
            setattr(gu, n, getattr(gu, "GenerateBeamOutput", object))
    stubs = ("librosa", "jiwer", "wandb", "torchaudio", "evaluate", "kaldiio", "soundfile", "flashlight", "pyannote")

    class Stub(types.ModuleType):
        __path__ = []

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            v = Stub(self.__name__ + "." + name)
            setattr(self, name, v)
            return v

        def __call__(self, *a, **k):
            return self

        def __mro_entries__(self, bases):
            return (object,)

    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            return importlib.machinery.ModuleSpec(name, self, is_package=True) if name.split(".")[0] in stubs else None

        def create_module(self, spec):
            return Stub(spec.name)

        def exec_module(self, module):
            pass
    if not any(type(f).__name__ == "Finder" for f in sys.meta_path):
        sys.meta_path.insert(0, Finder())


def run_harness_cases():
    """the reference's OWN dataloader-side harness — `DataPreprocessingManagerCallback.default_transform` (callbacks.py:108-118: zero strip
    through `audio_object_stripper`, data_utils.py:173-177, zero-pad to >= 8000 samples, the evaluation transform chain of configs/default_data_preprocessing2d.json =
    feature extractor only) and `SpeechCollatorWithPadding.__call__` (collators.py:65-106: pad to a multiple of 100, attention mask, labels with pad -> -100 and
    unk -> -100, rename to the model's input name) — run on seeded clips with silent edges / fewer than 8000 samples / interior zeros, then the reference model on
    the collated batch."""
    import transformers  # noqa: F401
    from transformers import PreTrainedTokenizerFast
    _stub_absent_third_party()
    from tokenizers import Tokenizer, models, pre_tokenizers
    from utilities.callbacks import DataPreprocessingManagerCallback
    from utilities.collators import SpeechCollatorWithPadding
    from utilities.feature_extractors import CustomFeatureExtractor

    fe = CustomFeatureExtractor(feature_size=80, norm_type="utterance")
    cfg_pre = {"default_preprocessing": [{"name": "feature_extractor", "steps_before_activation": 0,
                                          "fn_call_params": {"return_attention_mask": False, "padding": False, "sampling_rate": 16000, "return_tensors": "pt"},
                                          "return_behaviour": ["input_features[0]"]}]}
    cb = DataPreprocessingManagerCallback(cfg_pre, {}, "audio", fe)
    from transformers import TrainerState
    cb.propagate_state_to_transforms(TrainerState())        # what on_init_end does (callbacks.py:133): step 0 activates the chain's steps_before_activation = 0 entries
    N = 16000 * 3
    clips = {   # stored padded to N samples with zeros (what a device batch looks like), true lengths alongside
        "silent_edges": np.concatenate([np.zeros(1000, np.float32), synth.normal(2, "h_a", (26000,), 0.1), np.zeros(3000, np.float32)]),
        "short": synth.normal(3, "h_b", (4800,), 0.08),                                           # < 8000 samples: zero-padded to 8000
        "interior_zeros": np.concatenate([synth.normal(4, "h_c", (9000,), 0.05), np.zeros(2000, np.float32), synth.normal(5, "h_d", (13000,), 0.05)]),
        "leading_only": np.concatenate([np.zeros(7, np.float32), synth.normal(6, "h_e", (40001,), 0.2)]),
    }
    names = list(clips)
    lens = np.array([len(clips[k]) for k in names], np.int32)
    waves = np.zeros((len(names), N), np.float32)
    for i, k in enumerate(names):
        waves[i, : lens[i]] = clips[k]
    out = cb.default_transform({"audio": [{"array": clips[k]} for k in names]}, "default_preprocessing")["audio"]
    feats = [np.asarray(o, np.float32) for o in out]
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "a": 4, "b": 5, "c": 6, "d": 7, "e": 8, "f": 9}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()

    class Tok(PreTrainedTokenizerFast):            # transformers 5.x dropped `batch_encode_plus` (4.39: the batched form of __call__); harness-side shim
        def batch_encode_plus(self, batch_text, **kw):
            return self(batch_text, **kw)
    tok = Tok(tokenizer_object=tk, pad_token="<pad>", unk_token="<unk>", bos_token="<s>", eos_token="</s>")
    texts = ["a b c d", "b zzz a", "f e d c b a", "c"]
    coll = SpeechCollatorWithPadding(feature_extractor=fe, tokenizer=tok, padding=True, pad_to_multiple_of=100, audio_path="audio", text_path="text",
                                     model_input_name="input_values", mask_unks=True)
    batch = coll([{"audio": torch.from_numpy(f)[None], "text": t} for f, t in zip(feats, texts)])
    assert "input_features" not in batch
    rec = dict(waves=waves, lens=lens, frames=np.array([f.shape[0] for f in feats], np.int32), feats_cat=np.concatenate(feats, 0),
               input_values=batch["input_values"].numpy().astype(np.float32), attention_mask=batch["attention_mask"].numpy().astype(np.int64),
               labels=batch["labels"].numpy().astype(np.int64), texts=np.array(texts), vocab_keys=np.array(list(vocab)), vocab_ids=np.array(list(vocab.values())))
    # the reference model on the collated batch
    cfg, model = build_reference(TINY)
    rec["weight_sum"] = load_seeded(model, 31)
    with torch.no_grad():
        o = model(batch["input_values"].clone(), attention_mask=batch["attention_mask"], labels=batch["labels"])
    rec["loss"], rec["logits"], rec["seed"] = np.float32(o.loss.item()), o.logits.numpy().astype(np.float32), np.int64(31)
    print("harness", [f.shape for f in feats], batch["input_values"].shape, batch["labels"].tolist(), float(o.loss))
    np.savez_compressed(os.path.join(HERE, "harness.npz"), **rec)


def _ckpt_avg_state_dicts():
    """three small 'checkpoints' from the seeded generator: fp32 tensors, a bf16 tensor, an int64 step counter, and a key the middle checkpoint lacks"""
    sds = []
    for i in range(3):
        sd = {"enc.weight": torch.from_numpy(synth.normal(40 + i, "ck_w", (7, 5), 1.0)), "enc.bias": torch.from_numpy(synth.normal(40 + i, "ck_b", (5,), 0.3)),
              "head.weight": torch.from_numpy(synth.normal(40 + i, "ck_h", (4, 6), 2.0)).to(torch.bfloat16), "step_counter": torch.tensor(10 * (i + 1) + i, dtype=torch.int64)}
        if i != 1:
            sd["extra.scale"] = torch.from_numpy(synth.normal(40 + i, "ck_e", (3,), 1.0))
        sds.append(sd)
    return sds


def run_ckpt_average_case():
    """the reference's OWN `average_checkpoints` (model_utils.py:54-65, with `average_dicts`, general_utils.py:88-101) run on three seeded checkpoints
    in a temporary experiment directory; the fixture keeps what it wrote (the tests rebuild the same checkpoints from the seeds)."""
    import tempfile
    import transformers  # noqa: F401
    _stub_absent_third_party()
    import utilities.model_utils as MU
    with tempfile.TemporaryDirectory() as td:
        for i, sd in enumerate(_ckpt_avg_state_dicts()):
            os.makedirs(os.path.join(td, f"checkpoint-{(i + 1) * 500}"))
            torch.save(sd, os.path.join(td, f"checkpoint-{(i + 1) * 500}", "pytorch_model.bin"))
            with open(os.path.join(td, f"checkpoint-{(i + 1) * 500}", "config.json"), "w") as f:
                f.write('{"ckpt": %d}' % i)
        for aux, fn in (("tokenizer", "tokenizer.json"), ("feature_extractor", "preprocessor_config.json")):
            os.makedirs(os.path.join(td, aux))
            open(os.path.join(td, aux, fn), "w").write("{}")
        import glob
        order = [os.path.basename(os.path.dirname(p)) for p in glob.glob(f"{td}/checkpoint*/pytorch_model.bin")]
        dst = MU.average_checkpoints(td)
        avg = torch.load(os.path.join(dst, "pytorch_model.bin"), weights_only=True)
        files = sorted(os.listdir(dst))
        first_cfg = open(os.path.join(dst, "config.json")).read()
    rec = {"files": np.array(files), "glob_order": np.array(order), "first_cfg": np.array(first_cfg)}
    for k, v in avg.items():
        rec["avg/" + k] = v.float().numpy() if v.dtype == torch.bfloat16 else v.numpy()
        rec["dtype/" + k] = np.array(str(v.dtype))
    print("ckpt_avg", order, {k: (str(v.dtype), tuple(v.shape)) for k, v in avg.items()}, files)
    np.savez_compressed(os.path.join(HERE, "ckpt_avg.npz"), **rec)


def run_length_tables():
    cfg, model = build_reference(TINY)
    L = torch.arange(50, 3001)
    rec = dict(L=L.numpy(), inner=model.wav2vec2._get_feat_extract_output_lengths(L).numpy(),
               outer=model._get_feat_extract_output_lengths(L).numpy())
    cfgc, modelc = build_reference(TINY, is_causal=True)
    rec["inner_causal"] = modelc.wav2vec2._get_feat_extract_output_lengths(L).numpy()
    np.savez_compressed(os.path.join(HERE, "lengths.npz"), **rec)
    print("lengths 998 ->", int(rec["inner"][998 - 50]), int(rec["outer"][998 - 50]))


def run_ctc_known_answers():
    """torch.nn.functional.ctc_loss as called at e_branchformer.py:480-488 (aten, fp32)."""
    rec = {}
    g = torch.Generator().manual_seed(0)
    cases = {
        "basic": dict(T=12, V=7, labels=[[1, 2, 3], [4, 4, 0]], in_len=[12, 9], tl=[3, 2]),
        "repeat": dict(T=8, V=5, labels=[[2, 2, 2], [1, 0, 0]], in_len=[8, 5], tl=[3, 1]),
        "infeasible": dict(T=6, V=5, labels=[[1, 1, 1, 1], [2, 3, 0, 0]], in_len=[4, 6], tl=[4, 2]),
        "empty_target": dict(T=5, V=4, labels=[[0, 0], [1, 2]], in_len=[5, 5], tl=[0, 2]),
    }
    for name, c in cases.items():
        B = len(c["labels"])
        logits = torch.randn(B, c["T"], c["V"] + 1, generator=g)
        lp = torch.log_softmax(logits, -1)
        lab = torch.tensor(c["labels"])
        tl = torch.tensor(c["tl"])
        flat = torch.cat([lab[b, : tl[b]] for b in range(B)])
        for zi in (False, True):
            for red in ("mean", "sum", "none"):
                v = torch.nn.functional.ctc_loss(lp.transpose(0, 1), flat, torch.tensor(c["in_len"]), tl, blank=c["V"],
                                                 reduction=red, zero_infinity=zi)
                rec[f"{name}/{red}/{int(zi)}"] = v.numpy()
        lab_p = lab.clone()
        for b in range(B):
            lab_p[b, tl[b]:] = -100
        rec[f"{name}/logits"] = logits.numpy()
        rec[f"{name}/labels"] = lab_p.numpy()
        rec[f"{name}/in_len"] = np.array(c["in_len"])
    np.savez_compressed(os.path.join(HERE, "ctc_known.npz"), **rec)


TINY_DEC = dict(vocab_size=51, n_embd=128, n_layer=3, n_head=2, n_positions=64, head_locations=[1], head_weights=[0.4, 0.6])


def build_reference_aed(pos_emb_fixed=False, enc_extra=None):
    from utilities.bind import bind_all
    bind_all()
    from models.ctc_encoder_plus_autoregressive_decoder import JointCTCAttentionEncoderDecoder, JointCTCAttentionEncoderDecoderConfig
    from models.decoders.multi_head_gpt2 import GPT2LMMultiHeadModel, GPT2MultiHeadConfig
    from models.embeddings import AdaptiveEmbedding, PositionalEmbedding
    from transformers.models.gpt2.modeling_gpt2 import GPT2LMHeadModel

    class Dec(GPT2LMMultiHeadModel):          # shim (4) of SURVEY.md §8c
        model_parallel = False

        def tie_weights(self, *a, **k):
            return GPT2LMHeadModel.tie_weights(self, *a, **k)

    ecfg, enc = build_reference(TINY, **(enc_extra or {}))
    dcfg = GPT2MultiHeadConfig(**TINY_DEC, add_cross_attention=True, attn_implementation="eager", bos_token_id=2, eos_token_id=1,
                               pad_token_id=50, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, tie_word_embeddings=False)
    dcfg.lsm_factor = 0.1
    dcfg.cross_attention_hidden_size = None
    dcfg.pos_emb_fixed = pos_emb_fixed
    dec = Dec(dcfg)
    if pos_emb_fixed:                         # what PositionalEncodingInitModifier does (auto_wrappers.py:186-209)
        dec.transformer.wte = AdaptiveEmbedding(n_token=dcfg.vocab_size, d_embed=dcfg.hidden_size, d_proj=dcfg.hidden_size, cutoffs=[])
        dec.transformer.wpe = PositionalEmbedding(demb=dcfg.hidden_size)
    dec = dec.eval()
    # shim (5): transformers 5.15's GPT2Model.forward DROPS `encoder_attention_mask` (it rebuilds the cross mask from None),
    # whereas the pinned 4.39.3 applies invert_attention_mask(encoder_attention_mask) = (1 - mask) * finfo.min.  Restore the
    # pinned behaviour harness-side: remember the mask given to the decoder and hand it to the block as the 4-D additive mask.
    import transformers.models.gpt2.modeling_gpt2 as mg
    holder = {}
    orig_forward = dec.forward

    def fwd(*a, **k):
        holder["mask"] = k.get("encoder_attention_mask")
        return orig_forward(*a, **k)

    def bidir(config=None, inputs_embeds=None, attention_mask=None, encoder_hidden_states=None, **kw):
        m = holder.get("mask")
        if m is None:
            return None
        return (1.0 - m[:, None, None, :].to(inputs_embeds.dtype)) * torch.finfo(inputs_embeds.dtype).min

    dec.forward = fwd
    mg.create_bidirectional_mask = bidir
    jcfg = JointCTCAttentionEncoderDecoderConfig.from_encoder_decoder_configs(ecfg, dcfg, ctc_weight=0.3, lsm_factor=0.1, pad_token_id=50,
                                                                               decoder_start_token_id=2, shared_lm_head=False)
    return JointCTCAttentionEncoderDecoder(config=jcfg, encoder=enc, decoder=dec).eval()


def run_aed_cases():
    for name, fixed, seed in (("aed_tiny", False, 31), ("aed_tiny_fixedpos", True, 32)):
        try:
            model = build_reference_aed(fixed)
            wsum = load_seeded(model, seed)
            B, T, U = 2, 200, 9
            x, am = synth_feats(seed, B, T, [198, 150])
            lab = synth_labels(seed, B, U, 50, [9, 6])
            lab[lab >= 0] = np.maximum(lab[lab >= 0], 3)
            with torch.no_grad():
                out = model(input_values=torch.from_numpy(x), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab))
            rec = dict(seed=seed, weight_sum=wsum, lengths=np.array([198, 150]), shape=np.array([B, T, U]), labels=lab,
                       loss=float(out.loss), enc_loss=float(out.enc_loss), dec_loss=float(out.dec_loss), logits=out.logits.numpy(),
                       encoder_logits=out.encoder_logits.numpy(), encoder_hidden=out.encoder_last_hidden_state.numpy(),
                       param_names=np.array([k for k, _ in model.named_parameters()]),
                       param_shapes=np.array([str(tuple(v.shape)) for _, v in model.named_parameters()]))
            np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
            print(name, "loss", rec["loss"], rec["enc_loss"], rec["dec_loss"])
        except Exception as e:      # noqa: BLE001
            print(name, "FAILED in the reference under transformers 5.15:", type(e).__name__, str(e)[:200])


def run_aed_grad_cases():
    """training-mode forward + backward of the reference joint model (all dropouts 0): three losses and every parameter gradient."""
    for name, fixed, seed in (("grads_aed_tiny", False, 31), ("grads_aed_tiny_fixedpos", True, 32)):
        model = build_reference_aed(fixed, enc_extra=NO_DROPOUT)
        model.train()
        wsum = load_seeded(model, seed)
        B, T, U = 2, 200, 9
        x, am = synth_feats(seed, B, T, [198, 150])
        lab = synth_labels(seed, B, U, 50, [9, 6])
        lab[lab >= 0] = np.maximum(lab[lab >= 0], 3)
        out = model(input_values=torch.from_numpy(x), attention_mask=torch.from_numpy(am), labels=torch.from_numpy(lab))
        out.loss.backward()
        rec = dict(seed=seed, weight_sum=wsum, lengths=np.array([198, 150]), shape=np.array([B, T, U]), labels=lab,
                   loss=float(out.loss), enc_loss=float(out.enc_loss), dec_loss=float(out.dec_loss),
                   param_names=np.array([k for k, _ in model.named_parameters()]),
                   param_shapes=np.array([str(tuple(v.shape)) for _, v in model.named_parameters()]))
        n = 0
        for k, v in model.named_parameters():
            if v.grad is not None:
                rec["grad:" + k] = v.grad.float().numpy(); n += 1
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
        print(name, "loss", rec["loss"], rec["enc_loss"], rec["dec_loss"], "n grads", n)


BESTRQ = dict(best_rq_codebook_size=96, best_rq_codebook_dim=8, best_rq_in_dim=320, best_rq_num_books=2)


def run_bestrq_case():
    """BEST-RQ pre-training (src/models/bestrq.py): quantizer targets, loss and every gradient of the reference model in train() mode.
    Harness-side shims: (2') the BestRQConfig attributes are set with setattr (its __init__ is not reached through the multiple-inheritance
    chain under transformers 5.15, same cause as shim (2) of SURVEY.md §8c); (6) `BestRQMask._mask_hidden_states` draws N(0, 0.1) from
    torch's RNG — the harness substitutes a version that writes a GIVEN noise tensor (huggingface_asr_amd.synth.mask_noise, the kernels'
    counter-based noise) at the masked frames, so that the fixture is reproducible by the HIP path."""
    from models import bestrq as BQ
    seed, B, T, lengths = 51, 2, 200, [200, 168]
    cfg = BQ.BestRQEBranchformerForPreTrainingConfig(**TINY, attn_implementation="eager", layerdrop=0.0, **NO_DROPOUT)
    for k, v in dict(num_fbanks=80, conv_padding=[1, 1], context_awareness_type=None, **BESTRQ).items():
        setattr(cfg, k, v)
    model = BQ.BestRQEBranchformerForPreTraining(cfg).train()
    wsum = load_seeded(model, seed)
    x, am = synth_feats(seed, B, T, lengths)
    T2, d, L = 50, cfg.hidden_size, cfg.num_hidden_layers
    mask = np.zeros((B, T2), dtype=bool)
    mask[0, 4:12] = True; mask[0, 30:37] = True; mask[1, 10:22] = True
    sid = ((0 * 64 + L) * 16 + 3) & 0xFFFFFFFF
    noise = torch.from_numpy(synth.mask_noise(seed, sid, (B, T2, d), 0.1))
    mt = torch.from_numpy(mask)

    def patched(self, hidden_states, mask_time_indices=None, attention_mask=None, std=0.1):
        if mask_time_indices is not None:
            hidden_states[mask_time_indices] = noise[mask_time_indices]
        return hidden_states
    BQ.BestRQMask._mask_hidden_states = patched
    out = model(torch.from_numpy(x), attention_mask=torch.from_numpy(am), mask_time_indices=mt)
    out.loss.backward()
    with torch.no_grad():
        tg = model.rpq(torch.from_numpy(x).view(B, T2, -1))
    rec = dict(seed=seed, weight_sum=wsum, lengths=np.array(lengths), shape=np.array([B, T, T2]), mask=mask, loss=float(out.loss),
               rpq_P=model.rpq.P.numpy(), rpq_CB=model.rpq.CB.numpy(), targets=tg.numpy(), last_hidden=out.projected_states.detach().numpy(),
               param_names=np.array([k for k, _ in model.named_parameters()]),
               param_shapes=np.array([str(tuple(v.shape)) for _, v in model.named_parameters()]))
    for k, v in model.named_parameters():
        if v.grad is not None:
            rec["grad:" + k] = v.grad.float().numpy()
    np.savez_compressed(os.path.join(HERE, "bestrq_tiny.npz"), **rec)
    print("bestrq_tiny loss", rec["loss"], "targets", tg.shape, "n grads", sum(k.startswith("grad:") for k in rec))


SPECAUG_RECIPE = dict(apply_time_warp=True, time_warp_window=5, time_warp_mode="bicubic", apply_freq_mask=True, freq_mask_width_range=[0, 27],
                      num_freq_mask=2, apply_time_mask=True, time_mask_width_ratio_range=[0, 0.05], num_time_mask=5)


def run_finetune_cases():
    """CTC fine-tuning head of a BEST-RQ encoder with the recipes' two options (src/models/bestrq.py:192-322;
    `finetune_with_additional_layer=True,finetune_with_layer_mixing=True` in recipes/librispeech/ssl/*/lumi/finetune_frozen*.sh):
    eval logits + loss, and train-mode (dropouts 0) loss + every parameter gradient.  Same shim (2') as run_bestrq_case."""
    from models import bestrq as BQ
    for name, seed, lengths, tgt, flags in [("finetune_tiny_mix_extra", 61, [200, 142], [6, 4], dict(finetune_with_additional_layer=True, finetune_with_layer_mixing=True)),
                                            ("finetune_tiny_mix", 62, [188, 200], [5, 6], dict(finetune_with_additional_layer=False, finetune_with_layer_mixing=True)),
                                            ("finetune_tiny_extra", 63, [200, 120], [6, 3], dict(finetune_with_additional_layer=True, finetune_with_layer_mixing=False))]:
        B, T, U = 2, 200, 6
        cfg = BQ.BestRQEBranchformerForPreTrainingConfig(**TINY, attn_implementation="eager", layerdrop=0.0, ctc_zero_infinity=True,
                                                         ctc_loss_reduction="mean", **NO_DROPOUT, **flags)
        for k, v in dict(num_fbanks=80, conv_padding=[1, 1], context_awareness_type=None, **BESTRQ, **flags).items():
            setattr(cfg, k, v)
        model = BQ.BestRQEBranchformerForCTC(cfg).eval()
        wsum = load_seeded(model, seed)
        x, am = synth_feats(seed, B, T, lengths)
        lab = synth_labels(seed, B, U, cfg.vocab_size, tgt)
        xs, ams, labs = torch.from_numpy(x), torch.from_numpy(am), torch.from_numpy(lab)
        with torch.no_grad():
            ev = model(xs, attention_mask=ams, labels=labs)
        rec = dict(seed=seed, weight_sum=wsum, lengths=np.array(lengths), tgt_lens=np.array(tgt), shape=np.array([B, T, U]),
                   eval_loss=float(ev.loss), eval_logits=ev.logits.float().numpy(), flags=np.array([int(flags["finetune_with_additional_layer"]), int(flags["finetune_with_layer_mixing"])]))
        model.train()
        out = model(xs, attention_mask=ams, labels=labs)
        out.loss.backward()
        rec["loss"] = float(out.loss)
        for k, v in model.named_parameters():
            if v.grad is not None:
                rec["grad:" + k] = v.grad.float().numpy()
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
        print(name, "eval loss", rec["eval_loss"], "train loss", rec["loss"], "n grads", sum(k.startswith("grad:") for k in rec),
              "params", [k for k, _ in model.named_parameters() if "additional" in k or "per_layer" in k][:3])


def run_specaug_cases():
    """reference src/augmentations/spec_aug.py (recipe parameters of configs/default_data_preprocessing2d.json:36-58) on seeded features."""
    from augmentations.spec_aug import SpecAug
    rec = {}
    cases = {"equal": (2, 240, None, 3), "single": (1, 333, None, 4), "ragged": (3, 260, [260, 197, 121], 5), "short": (2, 9, None, 6)}
    for name, (B, T, lens, seed) in cases.items():
        x = torch.from_numpy(synth.normal(seed, "specaug/" + name, (B, T, 80), 1.0))
        aug = SpecAug(**SPECAUG_RECIPE)
        torch.manual_seed(100 + seed)
        y, _ = aug(x.clone(), None if lens is None else torch.tensor(lens))
        rec[name + "/shape"] = np.array([B, T, seed]); rec[name + "/lens"] = np.array(lens if lens else [])
        rec[name + "/out"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "specaug.npz"), **rec)
    print("specaug cases", {k: v.shape for k, v in rec.items() if k.endswith("/out")})


WHISPER_TINY = dict(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, num_mel_bins=80, max_source_positions=100)


def run_whisper_cases():
    """transformers WhisperFeatureExtractor (numpy path = the pinned 4.39.3 behaviour) and WhisperEncoder."""
    from transformers import WhisperConfig, WhisperFeatureExtractor
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    fe = WhisperFeatureExtractor(feature_size=80)
    rec = {}
    waves = {"noise": synth.normal(3, "wh_noise", (16000 * 3,), 0.1),
             "tone": (0.3 * np.sin(2 * np.pi * 440 * np.arange(16000 * 2) / 16000)).astype(np.float32)}
    for k, w in waves.items():
        padded = np.zeros(fe.n_samples, np.float32); padded[: len(w)] = w
        rec[f"fe/{k}_wave"] = w
        rec[f"fe/{k}_logmel"] = fe._np_extract_fbank_features(padded[None], "cpu")[0].astype(np.float32)   # (80, 3000)
    cfg = WhisperConfig(**WHISPER_TINY, attn_implementation="eager", dropout=0.0, activation_function="gelu")
    enc = WhisperEncoder(cfg).eval()
    wsum = load_seeded(enc, 41)
    x = synth.normal(41, "wh_feats", (2, 80, 200), 0.5)
    with torch.no_grad():
        out = enc(torch.from_numpy(x)).last_hidden_state
    rec.update(seed=41, weight_sum=wsum, enc_out=out.numpy(),
               param_names=np.array([k for k, _ in enc.named_parameters()]),
               param_shapes=np.array([str(tuple(v.shape)) for _, v in enc.named_parameters()]))
    np.savez_compressed(os.path.join(HERE, "whisper.npz"), **rec)
    print("whisper cases written", out.shape, float(out.std()))


def run_whisper_small_case():
    """BASELINE config 4 at its REAL shape: transformers' WhisperEncoder with the whisper-small configuration (d 768, 12 layers, 12 heads x 64, FFN 3072,
    1500 positions), seeded weights, two 30 s inputs (80 x 3000).  Stored: a strided slice of the output, per-frame L2 norms, and the same slice under the
    reference stack's own bf16 autocast (the size of the bf16 gap the HIP path is held to)."""
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    cfg = WhisperConfig(d_model=768, encoder_layers=12, encoder_attention_heads=12, encoder_ffn_dim=3072, num_mel_bins=80, max_source_positions=1500,
                        attn_implementation="eager", dropout=0.0, activation_function="gelu")
    enc = WhisperEncoder(cfg).eval()
    wsum = load_seeded(enc, 43)
    # keep transformers' sinusoid position table (it is a non-trained embedding that load_seeded would otherwise overwrite with noise)
    from transformers.models.whisper.modeling_whisper import sinusoids
    with torch.no_grad():
        enc.embed_positions.weight.copy_(sinusoids(1500, 768))
    x = synth.normal(43, "wh_small_feats", (2, 80, 3000), 0.5)
    with torch.no_grad():
        out = enc(torch.from_numpy(x)).last_hidden_state
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out16 = enc(torch.from_numpy(x)).last_hidden_state.float()
    rec = dict(seed=43, weight_sum=wsum, out_slice=out[:, ::50, ::16].numpy(), out_norm=out.norm(dim=-1).numpy(), out_slice_bf16=out16[:, ::50, ::16].numpy(),
               out_std=np.float32(out.std()), param_names=np.array([k for k, _ in enc.named_parameters()]),
               param_shapes=np.array([str(tuple(v.shape)) for _, v in enc.named_parameters()]))
    d = (out16 - out).abs()
    print("whisper small", out.shape, float(out.std()), "bf16 autocast gap max/mean", float(d.max()), float(d.mean()))
    rec["bf16_gap_max"], rec["bf16_gap_mean"] = np.float32(d.max()), np.float32(d.mean())
    np.savez_compressed(os.path.join(HERE, "whisper_small.npz"), **rec)


def run_ctc_prefix_cases():
    """reference src/decoding/ctc_scorer.py: CTCRescorerLogitsProcessor over 4 decoding steps (B=2, W=3)."""
    from decoding.ctc_scorer import CTCPrefixScoreTH, CTCRescorerLogitsProcessor, LogSoftmaxProcessor

    class Recorder(CTCPrefixScoreTH):
        def __call__(self, y, state, scoring_ids=None, att_w=None):
            out = super().__call__(y, state, scoring_ids, att_w)
            self.last_scores = out[0].clone()
            return out

    rec = {}
    # case "m" = case "a" with ctc_margin 6: the reference's processor never hands attention weights to the scorer (ctc_scorer.py:330), so the margin changes nothing
    for case, (B, W, T, O, lens, trick, margin) in {"a": (2, 3, 40, 30, [40, 33], False, 0), "b": (1, 5, 75, 51, [75], True, 0),
                                                    "c": (3, 2, 24, 17, [20, 24, 9], False, 0), "m": (2, 3, 40, 30, [40, 33], False, 6)}.items():
        g = torch.Generator().manual_seed(7)
        enc_logits = torch.randn(B, T, O, generator=g) * 2.0
        blank, eos, space = O - 1, 1, 5
        proc = CTCRescorerLogitsProcessor(enc_logits.clone(), torch.tensor(lens), blank, eos, margin, 0.3, W, space, trick, 0.8)
        proc.ctc_prefix_scorer.__class__ = Recorder
        ids = torch.zeros((B * W, 1), dtype=torch.long) + 2                      # start token
        rec[f"{case}/enc_logits"] = enc_logits.numpy(); rec[f"{case}/lens"] = np.array(lens)
        rec[f"{case}/meta"] = np.array([B, W, T, O, blank, eos, space, int(trick), margin])
        for step in range(4):
            att = torch.log_softmax(torch.randn(B * W, O, generator=g) * 1.5, -1)
            if trick and step == 2:
                att[:, eos] = 0.0                                                 # provoke the eos/space branch
            out = proc(ids, att.clone())
            rec[f"{case}/step{step}/input_ids"] = ids.numpy().copy()
            rec[f"{case}/step{step}/att"] = att.numpy()
            rec[f"{case}/step{step}/ctc"] = proc.ctc_prefix_scorer.last_scores.numpy().copy()
            rec[f"{case}/step{step}/out"] = out.numpy().copy()
            # beam k of every utterance takes its (k+1)-th best non-blank token -> distinct prefixes per beam
            order = out.clone(); order[:, blank] = -1e30
            top = order.topk(W, dim=1).indices
            nxt = torch.stack([top[i, i % W] for i in range(B * W)])
            ids = torch.cat([ids, nxt[:, None]], 1)
        lsm = LogSoftmaxProcessor()(ids, att.clone())
        rec[f"{case}/logsoftmax"] = lsm.numpy()
    np.savez_compressed(os.path.join(HERE, "ctc_prefix.npz"), **rec)
    print("ctc_prefix cases written")


def _install_generate_adapters():
    """Harness-side adapters for the `generate()` call chain (transformers 4.39 -> 5.15): the reference's overrides keep the 4.39 signatures —
    `_prepare_encoder_decoder_kwargs_for_generation(self, inputs_tensor, model_kwargs, model_input_name)` calls its parent without the `generation_config`
    argument 5.15 added (ctc_encoder_plus_autoregressive_decoder.py:406-418), `_get_logits_processor` (:360-404) has no `device` parameter and hands its
    parent `model_kwargs, negative_prompt_ids, negative_prompt_attention_mask` positionally where 5.15 expects `device` first.  The adapters restore the
    missing argument on the parent's side from a stash a thin subclass fills on the way in; the reference's own bodies run unchanged.  Returns the class
    factory `adapt(model)` and the recorder of the loop's decisions (instrumentation only: it reads the tensors transformers' beam loop computes)."""
    from transformers.generation.utils import GenerationMixin
    if getattr(GenerationMixin, "_hfasr_gen_adapters", None) is not None:
        return GenerationMixin._hfasr_gen_adapters
    stash, rec = {}, {"margin": [], "stop_gap": []}
    orig_prep = GenerationMixin._prepare_encoder_decoder_kwargs_for_generation
    orig_glp = GenerationMixin._get_logits_processor
    orig_topk = GenerationMixin._get_top_k_continuations
    orig_heur = GenerationMixin._check_early_stop_heuristic

    def prep(self, inputs_tensor, model_kwargs, model_input_name=None, generation_config=None):
        return orig_prep(self, inputs_tensor, model_kwargs, model_input_name, generation_config if generation_config is not None else stash["generation_config"])

    def glp(self, generation_config, input_ids_seq_length=None, encoder_input_ids=None, prefix_allowed_tokens_fn=None, logits_processor=None, *rest, **kw):
        if rest:                                   # the 4.39 positional order
            kw.update(dict(zip(("model_kwargs", "negative_prompt_ids", "negative_prompt_attention_mask"), rest)))
            kw.setdefault("device", stash.get("device"))
        return orig_glp(self, generation_config, input_ids_seq_length, encoder_input_ids, prefix_allowed_tokens_fn, logits_processor, **kw)

    def topk(self, accumulated_log_probs, *a, num_beams, **k):
        v = accumulated_log_probs.topk(num_beams + 1, dim=1).values
        rec["margin"].append((v[:, :-1] - v[:, 1:]).min(1).values.tolist())      # per utterance: the smallest gap among the top W + 1 candidates of this step
        return orig_topk(self, accumulated_log_probs, *a, num_beams=num_beams, **k)

    def heur(is_early_stop_heuristic_unsatisfied, running_beam_scores, beam_scores, is_sent_finished, cur_len, max_length, decoder_prompt_len, early_stopping, length_penalty):
        hl = (max_length if (early_stopping == "never" and length_penalty > 0.0) else cur_len) - decoder_prompt_len
        full = is_sent_finished.all(1)
        gap = (running_beam_scores[:, 0] / (hl ** length_penalty) - beam_scores.min(1).values).abs()
        rec["stop_gap"].append([float(g) if bool(f) and bool(u) else float("inf") for g, f, u in zip(gap, full, is_early_stop_heuristic_unsatisfied[:, 0])])
        return orig_heur(is_early_stop_heuristic_unsatisfied, running_beam_scores, beam_scores, is_sent_finished, cur_len, max_length, decoder_prompt_len,
                         early_stopping, length_penalty)
    GenerationMixin._prepare_encoder_decoder_kwargs_for_generation = prep
    GenerationMixin._get_logits_processor = glp
    GenerationMixin._get_top_k_continuations = topk
    GenerationMixin._check_early_stop_heuristic = staticmethod(heur)

    def adapt(model):
        ref = type(model)

        class Adapted(ref):
            def _prepare_encoder_decoder_kwargs_for_generation(self, inputs_tensor, model_kwargs, model_input_name=None, generation_config=None):
                stash["generation_config"] = generation_config
                return ref._prepare_encoder_decoder_kwargs_for_generation(self, inputs_tensor, model_kwargs, model_input_name)

            def _get_logits_processor(self, generation_config, input_ids_seq_length=None, encoder_input_ids=None, prefix_allowed_tokens_fn=None,
                                      logits_processor=None, device=None, model_kwargs=None, negative_prompt_ids=None, negative_prompt_attention_mask=None):
                stash["device"] = device
                return ref._get_logits_processor(self, generation_config, input_ids_seq_length, encoder_input_ids, prefix_allowed_tokens_fn, logits_processor,
                                                 model_kwargs, negative_prompt_ids, negative_prompt_attention_mask)
        model.__class__ = Adapted
        return model
    GenerationMixin._hfasr_gen_adapters = (adapt, rec)
    return adapt, rec


def run_generate_cases():
    """the reference joint model's OWN `generate()` (ctc_encoder_plus_autoregressive_decoder.py:450-482 -> transformers' GenerationMixin with the
    reference's processors, :360-404) on the tiny AED configuration with the structured decoder of tests/gen_model.py, decoded the way `do_generate` asks
    (general_utils.py:198-218: `num_return_sequences = num_beams`, `return_dict_in_generate`, `output_scores`; the generation configuration assigned to the model as
    train_enc_dec_asr.py:61-85 does): greedy and beams of 3 / 5, three length penalties, `early_stopping` False / True / "never", a `max_length` at which some beams close on
    the end-of-sequence token and the rest when the length runs out, and one short enough that nothing closes before it.  Stored per setting: sequences, sequence scores, and
    the smallest decision margins of the loop (candidate gaps, stopping-rule gaps) so that the tests can assert the fixture is decided far above bf16 noise."""
    sys.path.insert(0, os.path.dirname(HERE))
    import gen_model as GM
    from decoding.config import GenerationConfigCustom
    adapt, rec = _install_generate_adapters()
    for name, (seed, fixed, lengths) in GM.CASES.items():
        model = adapt(build_reference_aed(fixed))
        wsum = load_seeded(model, seed)
        ov = GM.overrides(seed, fixed)
        missing, unexpected = model.load_state_dict(ov, strict=False)
        assert not unexpected, unexpected
        B, T = len(lengths), 200
        x, am = synth_feats(seed, B, T, lengths)
        out = dict(seed=seed, weight_sum=wsum, lengths=np.array(lengths), shape=np.array([B, T]), fixed_pos=np.array(int(fixed)),
                   override_sum=np.float64(sum(float(v.double().sum()) for v in ov.values())),
                   param_names=np.array([k for k, _ in model.named_parameters()]), param_shapes=np.array([str(tuple(v.shape)) for _, v in model.named_parameters()]))
        for W, lp, es, ml in GM.SETTINGS:
            g = GenerationConfigCustom(bos_token_id=GM.START, pad_token_id=GM.PAD, decoder_start_token_id=GM.START, length_penalty=lp, early_stopping=es, eos_token_id=GM.EOS,
                                       max_length=ml, num_beams=W, ctc_weight=0.3, ctc_margin=0, lm_weight=0, lm_model=None, space_token_id=-1, apply_eos_space_trick=False,
                                       eos_space_trick_weight=1.0)
            model.generation_config = g                                           # train_enc_dec_asr.py:85
            g.num_return_sequences, g.return_dict_in_generate, g.output_scores = W, True, True     # general_utils.py:198-201
            rec["margin"].clear(); rec["stop_gap"].clear()
            o = model.generate(generation_config=g, input_values=torch.from_numpy(x), attention_mask=torch.from_numpy(am))
            key = GM.setting_key(W, lp, es, ml)
            out[key + "/sequences"] = o.sequences.numpy()
            if W > 1:
                out[key + "/sequences_scores"] = o.sequences_scores.numpy()
                out[key + "/min_margin"] = np.float32(min(min(m) for m in rec["margin"]))
                sg = [v for row in rec["stop_gap"] for v in row if np.isfinite(v)]
                out[key + "/min_stop_gap"] = np.float32(min(sg) if sg else np.inf)
            else:                                     # greedy: the gap between the two best tokens of every step before the utterance closed
                gaps = []
                seqs = o.sequences.numpy()
                for t, sc in enumerate(o.scores):
                    top2 = sc.topk(2, dim=1).values
                    for b in range(B):
                        if t == 0 or (seqs[b, t] != GM.EOS and seqs[b, t] != GM.PAD):
                            gaps.append(float(top2[b, 0] - top2[b, 1]))
                out[key + "/min_margin"] = np.float32(min(gaps))
            print(name, key, "margin", float(out[key + "/min_margin"]), "stop gap", float(out.get(key + "/min_stop_gap", np.inf)))
            for i, s in enumerate(o.sequences.tolist()):
                print("    ", s, float(o.sequences_scores[i]) if W > 1 else "")
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "grads", "base", "basegrads", "fbank", "harness", "ckptavg", "lengths", "ctc", "prefix", "aed", "aedgrads", "gen", "bestrq", "finetune", "specaug", "whisper", "whispersmall"]
    if "tiny" in which:
        run_encoder_case("tiny_rel", TINY, seed=11, B=2, T=200, lengths=[198, 150], U=7, tgt_lens=[7, 5])
        run_encoder_case("tiny_rotary", TINY, seed=12, B=2, T=200, lengths=[200, 131], U=6, tgt_lens=[6, 4],
                         position_embeddings_type="rotary")
        run_encoder_case("tiny_causal", TINY, seed=13, B=2, T=160, lengths=[160, 97], U=5, tgt_lens=[5, 3], is_causal=True)
        run_encoder_case("tiny_nomacaron", TINY, seed=14, B=1, T=120, lengths=[120], U=4, tgt_lens=[4], use_macaron_ff=True,
                         csgu_activation="gelu", csgu_use_linear_after_conv=True)
    if "gated" in which or "tiny" in which:
        # context-aware Conv2d front ends (extractors.py:23-65): `gated` = conv * sigmoid(gate), same geometry; `gated_shared` = one gate per four time steps from a
        # (12,3) / stride (8,2) / padding (4,1) conv (its `view` needs both conv outputs' time axes divisible by 4: T = 208 -> 104 -> 52); any other string —
        # incl. the recipes' `shared_gated` (train_shared_gated_baseline.sh:94) — falls through to the plain nn.Conv2d (extractors.py:60-61)
        run_encoder_case("tiny_gated", TINY, seed=26, B=2, T=200, lengths=[200, 139], U=6, tgt_lens=[6, 4], context_awareness_type="gated")
        run_encoder_case("tiny_gated_shared", TINY, seed=27, B=2, T=208, lengths=[208, 150], U=6, tgt_lens=[6, 4], context_awareness_type="gated_shared")
        run_encoder_case("tiny_shared_gated_fallthrough", TINY, seed=26, B=2, T=200, lengths=[200, 139], U=6, tgt_lens=[6, 4], context_awareness_type="shared_gated")
    if "gatedgrads" in which or "grads" in which:
        run_grad_case("grads_tiny_gated", TINY, seed=26, B=2, T=200, lengths=[200, 139], U=6, tgt_lens=[6, 4], context_awareness_type="gated")
        run_grad_case("grads_tiny_gated_shared", TINY, seed=27, B=2, T=208, lengths=[208, 150], U=6, tgt_lens=[6, 4], context_awareness_type="gated_shared")
    if "gatedbase" in which or "base" in which:
        # the recipes' size (train_gated_baseline.sh:94 on the 256-channel front end): the fused kernels (two filter banks in conv1, conv ⊕ gate rows in conv2's implicit GEMM)
        run_encoder_case("small_gated", SMALL, seed=28, B=2, T=1000, lengths=[998, 700], U=40, tgt_lens=[40, 31], full=False, with_bf16=True, context_awareness_type="gated")
    if "grads" in which:
        run_grad_case("grads_tiny_rel", TINY, seed=11, B=2, T=200, lengths=[198, 150], U=7, tgt_lens=[7, 5])
        run_grad_case("grads_tiny_rotary", TINY, seed=12, B=2, T=200, lengths=[200, 131], U=6, tgt_lens=[6, 4], position_embeddings_type="rotary")
        run_grad_case("grads_tiny_specaug", TINY, seed=16, B=2, T=200, lengths=[200, 140], U=5, tgt_lens=[5, 3], np_seed=5, apply_spec_augment=True,
                      mask_time_prob=0.3, mask_time_length=4, mask_time_min_masks=2, mask_feature_prob=0.2, mask_feature_length=3, mask_feature_min_masks=1)
    if "gradscausal" in which or "grads" in which:
        # the streaming encoder: left-padded Conv2d front end, triu attention mask, the CSGU conv dilated by 15 (T' = 100 frames: 7 of its 31 taps see data)
        run_grad_case("grads_tiny_causal", TINY, seed=17, B=2, T=400, lengths=[400, 263], U=6, tgt_lens=[6, 4], is_causal=True)
        # T' = 475 frames > the dilated conv's 450-frame reach: every one of the 31 taps of the causal CSGU conv meets data, forward and backward
        run_grad_case("grads_tiny_causal_long", TINY, seed=20, B=1, T=1900, lengths=[1900], U=12, tgt_lens=[12], is_causal=True)
    if "gradscsgu" in which or "grads" in which:
        # the CSGU's optional pieces (no reference recipe turns them on): Linear after the conv + GELU before the gate; SiLU without the Linear
        run_grad_case("grads_tiny_csgu_linear", TINY, seed=18, B=2, T=120, lengths=[120, 88], U=4, tgt_lens=[4, 3], csgu_activation="gelu", csgu_use_linear_after_conv=True)
        run_grad_case("grads_tiny_csgu_silu", TINY, seed=19, B=1, T=120, lengths=[120], U=4, tgt_lens=[4], csgu_activation="silu")
        # (use_macaron_ff=False is not runnable in the reference: its layer forward reads self.ff1 unconditionally, e_branchformer.py:271)
    if "basegrads" in which:
        run_grad_case_strided("grads_base_rel", BASE, seed=24, B=2, T=1000, lengths=[998, 700], U=40, tgt_lens=[40, 31])
    if "base" in which:
        run_encoder_case("small_rel", SMALL, seed=21, B=2, T=1000, lengths=[998, 700], U=40, tgt_lens=[40, 31], full=False)
        # the streaming model at a real size: causal front end, triu mask in the LDS-staged attention (head 64), the CSGU conv dilated by 15 over 250 frames
        run_encoder_case("small_causal", SMALL, seed=25, B=2, T=1000, lengths=[998, 700], U=40, tgt_lens=[40, 31], full=False, is_causal=True)
        run_encoder_case("base_rel", BASE, seed=22, B=2, T=1000, lengths=[998, 700], U=40, tgt_lens=[40, 31], full=False,
                         with_bf16=True)
        run_encoder_case("base_rotary", BASE, seed=23, B=2, T=1000, lengths=[998, 700], U=40, tgt_lens=[40, 31], full=False,
                         position_embeddings_type="rotary")
    if "fbank" in which:
        run_fbank_cases()
    if "harness" in which:
        run_harness_cases()
    if "ckptavg" in which:
        run_ckpt_average_case()
    if "lengths" in which:
        run_length_tables()
    if "ctc" in which:
        run_ctc_known_answers()
    if "prefix" in which:
        run_ctc_prefix_cases()
    if "aed" in which:
        run_aed_cases()
    if "aedgrads" in which:
        run_aed_grad_cases()
    if "gen" in which:
        run_generate_cases()
    if "bestrq" in which:
        run_bestrq_case()
    if "finetune" in which:
        run_finetune_cases()
    if "specaug" in which:
        run_specaug_cases()
    if "whisper" in which:
        run_whisper_cases()
    if "whispersmall" in which:
        run_whisper_small_case()
