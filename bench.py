#!/usr/bin/env python
"""Headline benchmark: audio-seconds/sec, encoder forward + CTC, E-Branchformer-base (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 outside torchrun: the bench starts its N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --train                             (BASELINE config 3: data-parallel training step, gradient all-reduce over RCCL)

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  32 x 10 s of 16 kHz audio -> log-mel + utterance CMVN (HIP) -> pad to 1000 frames -> Conv2d sub-sampling ->
  16 E-Branchformer layers (relative-position attention) -> lm_head ⊕ blank -> logits -> CTC loss (labels U=40).
The path shards by utterance with no exchange step: every rank runs the same per-GPU batch
("weak" scaling), value = audio-seconds all ranks processed / max-over-ranks time.  Weights are seeded random
(no checkpoints offline), data is synthetic.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from huggingface_asr_amd import _lib, fbank as FB, ops, shapes, synth  # noqa: E402
from huggingface_asr_amd.engine import EBranchformerEngine  # noqa: E402

BATCH, SECONDS, SR, U = 32, 10, 16000, 40
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"


def algorithmic_gflop_per_utt(cfg, T2):
    """2 x MACs of the GEMM/conv contractions only, per 10 s utterance."""
    d, I, L, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], cfg["vocab_size"]
    C1, C2 = cfg["conv_dim"]
    T1, F1, F2 = 500, 40, 20
    mac = T1 * F1 * C1 * 9 + T2 * F2 * C2 * 9 * C1 + T2 * (F2 * C2) * d + T2 * d * d
    hd = d // cfg["num_attention_heads"]
    per_layer = 2 * (2 * T2 * d * I) + 4 * T2 * d * d + (T2 * T2 * d) * 2 + T2 * (2 * T2 - 1) * d \
        + T2 * d * I + T2 * (I // 2) * 31 + T2 * (I // 2) * d + T2 * 2 * d * 31 + T2 * 2 * d * d
    mac += L * per_layer + T2 * d * (V + 1)
    return 2.0 * mac / 1e9


def pmc_traffic_bytes(kernel_prefix="gemm8p_kernel<false, 1"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*pmc*per_launch.txt: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs of this command, FETCH_SIZE doubled for 16-B/lane reads as MI355X_MICROARCH.md prescribes).  PMC counters cannot
    be collected inside the timed region, so the bench line carries the recorded value and names its source; None if the file is absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*per_launch*.txt")))
    if not files:
        return None, None
    for f in reversed(files):                                  # the newest table that has the kernel (tables of other commands — the training step — live beside it)
        for line in open(f):
            targs = line.split("|")[0].split("<", 1)[-1].rsplit(">", 1)[0].split(", ")
            if line.startswith(kernel_prefix) and len(targs) >= 5 and targs[4].strip() == "true":      # the LayerNorm-folded consumer the forward step runs (template argument LNF)
                cols = [c.strip() for c in line.split("|")]
                return (float(cols[3]) + float(cols[4])) * 1e6, os.path.relpath(f, ROOT)
    return None, None


FAMILIES = ["gemm8p 256x256 (QKV: bf16 out)", "gemm8p 256x256 + GELU (FFN in x2, cgMLP in)", "gemm8p implicit-GEMM conv2 + GELU", "gemm8p fp32 out (CTC head)",
            "gemm8p128 128x128 (FFN out x2, attention out, cgMLP out, merge, front-end out, feature projection)", "gemm_glds", "gemm_bf16 (generic)"]


def kernel_family(name: str):
    """csrc/gemm_args.hpp PF_* index of a dense-contraction kernel name as rocprofv3 prints it; None for every other kernel"""
    if "gemm8p128" in name:
        return 4
    if "gemm8p_kernel<true" in name:
        return 2
    if "gemm8p_kernel<false, 0, true" in name:
        return 3
    if "gemm8p_kernel<false, 0" in name:
        return 0
    if "gemm8p_kernel<false, 1" in name or "gemm8p_kernel<false, 2" in name:
        return 1
    if "gemm_glds_kernel" in name:
        return 5
    if "gemm_bf16_kernel" in name:
        return 6
    return None


def rocprof_child_dense_us(args, B, per_step):
    """The dense kernels' durations as they run in the REAL step — back to back, nothing between them — read from the dispatch timestamps of a rocprofv3 --kernel-trace
    run of this same script (a child process: the parent never execs; an event bracket would isolate every launch, and isolated launches run ~5 % faster than
    pipelined ones).  per_step: {family: launches per step}.  -> ({family: microseconds per step}, steps, note) or (None, 0, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, 0, "rocprofv3 not found"
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        return None, 0, "this process already runs under a profiler (no nested rocprofv3)"
    steps = max(3, min(10, args.event_steps))
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "2",
               "--batch", str(B), "--pos", args.pos, "--streams", "1", "--no-cpu-baseline", "--no-kernel-events", "--no-secondary"]
        try:
            r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)      # ~20 s when all is well
        except Exception as e:  # noqa: BLE001
            return None, 0, f"rocprofv3 child failed: {type(e).__name__}"
        files = glob.glob(os.path.join(td, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, 0, f"rocprofv3 child rc={r.returncode}, trace files: {len(files)}"
        if args.keep_profile:                                # the rocprofv3 summary this leg's numbers come from (profiles/ keeps a copy per round)
            os.makedirs(args.keep_profile, exist_ok=True)
            for f in glob.glob(os.path.join(td, "**", "*kernel_stats.csv"), recursive=True):
                shutil.copy(f, os.path.join(args.keep_profile, "roofline_child_kernel_stats.csv"))
        fam = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                k = kernel_family(row["Kernel_Name"])
                if k is not None:
                    fam.setdefault(k, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    out = {}
    for k, n in per_step.items():
        rows = sorted(fam.get(k, []))
        if len(rows) < n * steps:
            return None, 0, f"rocprofv3 child: family {k} has {len(rows)} launches, expected >= {n * steps}"
        out[k] = sum(d for _, d in rows[-n * steps:]) / steps / 1e3          # the timed steps are the last ones (warm-up first); ns -> us per step
    return out, steps, "ok"


def rocprof_child_in_flight(args, B):
    """The headline's OWN mode (steps in flight, wide tiles) as a kernel trace: a rocprofv3 --kernel-trace child run of this script with the default lanes and no one-step
    comparison, summarised by tools/lanes_trace.py — the share of the wall in which a dense kernel / only other kernels / nothing runs, and the non-GEMM share of the busy
    time.  Tracing serialises dispatch bookkeeping, so the traced wall is longer than the timed region's (both are reported); the shares are what the trace is for."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not found"}
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        return {"error": "this process already runs under a profiler (no nested rocprofv3)"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lanes_trace
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable, os.path.abspath(__file__), "--steps", "24", "--warmup", "8", "--batch", str(B),
               "--pos", args.pos, "--streams", str(args.streams), "--wide-tiles", str(int(args.wide_tiles)), "--no-cpu-baseline", "--no-kernel-events", "--no-secondary", "--no-one-step"]
        try:
            r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)      # ~20 s when all is well
        except Exception as e:  # noqa: BLE001
            return {"error": f"rocprofv3 child failed: {type(e).__name__}"}
        files = glob.glob(os.path.join(td, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"error": f"rocprofv3 child rc={r.returncode}, trace files: {len(files)}"}
        rec = lanes_trace.summarise(files, 0.4)
        if args.keep_profile:
            os.makedirs(args.keep_profile, exist_ok=True)
            json.dump(rec, open(os.path.join(args.keep_profile, "in_flight_trace.json"), "w"), indent=1)
    return rec


def cpu_baseline(cfg, sd, seconds_budget=25.0):
    """Time the ORACLE (CPU restatement; checker only, never the product path) on the host cores:
    feature extraction + encoder forward + CTC head, fp32, bounded sample."""
    from oracle import ebranchformer_ref as R
    from oracle import fbank_ref
    # a 1-GPU box grants a 16-CPU share of the host (more threads only oversubscribe the cgroup quota)
    n_thr = max(1, min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    torch.set_num_threads(n_thr)
    B = 4
    wave = synth.waveforms(7, B, SR * SECONDS)
    cfgd = dict(cfg)

    def one():
        feats = np.stack([np.pad(fbank_ref.extract(w), ((0, 2), (0, 0))) for w in wave])
        am = torch.zeros(B, 1000, dtype=torch.long); am[:, :998] = 1
        with torch.no_grad():
            h = R.encoder_forward(sd, cfgd, torch.from_numpy(feats), am)
            R.ctc_head(sd, h)
    one()
    t0 = time.perf_counter(); it = 0
    while True:
        one(); it += 1
        if time.perf_counter() - t0 > seconds_budget * 0.5 or it >= 40:
            break
    dt = time.perf_counter() - t0
    return dict(value=round(it * B * SECONDS / dt, 2), unit="audio-seconds/sec", cores=n_thr, kind="port",
                sample=f"{it} x (B={B} x {SECONDS}s clips): numpy float64 fbank+CMVN, torch-CPU fp32 encoder + CTC head")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks on this node; without a torchrun environment the bench starts them itself")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 32 forward / 96 --train)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--head-lse", action="store_true", help="the CTC loss's row log-sum-exp out of the head GEMM's epilogue (mi_ebf_forward_lse) instead of a pass of its own over the logits: same-box A/B (tools/head_lse_ab.sh)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the separate HIP-event pass over the dense GEMM launches (roofline leg)")
    ap.add_argument("--event-steps", type=int, default=5, help="steps of the separate roofline pass (every dense launch timed, stride 1; never inside the timed region)")
    ap.add_argument("--keep-profile", default=None, help="directory that receives the kernel_stats.csv of the roofline leg's rocprofv3 child run")
    ap.add_argument("--pos", default="relative", choices=["relative", "rotary"])
    ap.add_argument("--wide-tiles", type=int, default=None, choices=[0, 1],
                    help="forward bench, with --streams >= 3: the N = d GEMMs of a layer on 256 x 256 tiles (mi_ebf_config.wide_tiles) — a quarter of the blocks per launch, twice "
                         "the FLOP per ingested byte; kernels of the steps in flight then share the chip by CU")
    ap.add_argument("--streams", type=int, default=4, choices=[1, 2, 3, 4, 5, 6],
                    help="forward bench: steps in flight. Every step is one pass over its own batch of 32 clips; with k > 1 step j runs on HIP stream j %% k (k engines: own "
                         "workspace, same weights), so the ramp / drain of one step's kernels is filled by the other's. 1 = strictly one step at a time (the mode every "
                         "per-kernel figure — roofline, profiles/ — is taken in)")
    ap.add_argument("--train", action="store_true", help="BASELINE config 3 instead of the headline: data-parallel TRAINING step of the joint AED model "
                                                         "(small encoder + 6x256 GPT-2 decoder, per-GPU batch 96, 1-20 s clips), gradient all-reduce over RCCL")
    ap.add_argument("--overlap", type=int, default=0, choices=[0, 1], help="--train: gradient all-reduce schedule. 0 = ONE collective per parameter store after the backward "
                                                                           "(default); 1 = one async all-reduce per layer range as soon as it is final, overlapped with the "
                                                                           "backward of the earlier layers (what DDP's bucket hooks give the reference; DESIGN.md section 6 before enabling on RCCL)")
    ap.add_argument("--dropout", type=float, default=0.1, help="--train: dropout probability at every site of encoder and decoder (the reference's config defaults, which the "
                    "recipes leave in place: hidden / activation / attention / final / conv 0.1, GPT-2 resid / embd / attn 0.1); masks are counter-based, identical on re-runs")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo: tests with several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (tests on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous check on a box without GPUs: gloo, no HIP work, value null")
    ap.add_argument("--secondary", default=None, choices=["whisper", "decode", "train"], help="run ONE bounded secondary measurement (config 4 / 5 / 3) on cuda:0 and print its "
                                                                                                "JSON; the default run starts these as children after its timed region")
    ap.add_argument("--no-one-step", action="store_true", help="skip the one-step-at-a-time comparison after the timed region (a kernel trace of this command then shows the "
                                                                "headline's own mode only: tools/lanes_trace.py)")
    ap.add_argument("--no-in-flight-trace", action="store_true", help="skip the kernel trace of the headline's own mode (`roofline.in_flight.trace`)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary records (configs 3, 4, 5) of the default single-GPU run")
    ap.add_argument("--secondary-timeout", type=float, default=150.0, help="seconds allowed per secondary child")
    ap.add_argument("--no-train-dp", action="store_true", help="more than one rank: skip the config-3 training step under both gradient all-reduce schedules")
    ap.add_argument("--train-dp-batch", type=int, default=96, help="per-GPU batch of the `train_dp` record (config 3: 96)")
    ap.add_argument("--train-dp-steps", type=int, default=5, help="timed steps per schedule of the `train_dp` record")
    args = ap.parse_args(argv)
    if args.wide_tiles is None:
        args.wide_tiles = int(args.streams >= 3)         # wide tiles leave one step alone with a quarter of the blocks per launch: they come with the lanes that fill the rest
    return args


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script (one per GPU) with torch.distributed.run and hand their exit code on.
    This process has not touched the GPU (no torch.cuda / HIP call), and it stays the parent: nothing is exec'ed over it."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


def dry_run(args, world, rank):
    """no GPU on this box: exercise the launcher, the rendezvous and the timing collectives only (explicit flag; never a silent fallback)"""
    from huggingface_asr_amd import parallel as PL
    PL.init("gloo")
    import torch.distributed as td
    dt = PL.timed(lambda: time.sleep(0.001), args.steps)
    if rank == 0:
        print(json.dumps({"metric": "audio-seconds/sec encoder fwd+CTC, E-Branchformer-base", "value": None, "unit": "audio-seconds/sec", "n_gpus": world,
                          "rccl_ranks": td.get_world_size() if td.is_initialized() else 1, "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "none", "dry_run": True, "config": {"workload": "dry run: no HIP work (launcher / rendezvous / timing collectives only)"}}), flush=True)
    if td.is_initialized():
        td.barrier(); td.destroy_process_group()


def config3_gflop_per_step(cfg, dcfg, B, T2, U):
    """algorithmic GFLOP of ONE training step at config 3's padded shape: 3 x (forward contractions) — forward, data gradient, weight gradient — of the Conv2d front end,
    the encoder layers (as `algorithmic_gflop_per_utt`, at T' = T2), the CTC head and the decoder (self-attention over U tokens, cross-attention over T2 frames, MLP, lm_head)."""
    d, I, L, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], cfg["vocab_size"]
    C1, C2 = cfg["conv_dim"]
    T1, F1, F2 = 2 * T2, 40, 20
    mac = T1 * F1 * C1 * 9 + T2 * F2 * C2 * 9 * C1 + T2 * (F2 * C2) * d + T2 * d * d
    mac += L * (2 * (2 * T2 * d * I) + 4 * T2 * d * d + (T2 * T2 * d) * 2 + T2 * (2 * T2 - 1) * d + T2 * d * I + T2 * (I // 2) * 31 + T2 * (I // 2) * d + T2 * 2 * d * 31
                + T2 * 2 * d * d) + T2 * d * (V + 1)
    dd, Ld, Vd = dcfg["n_embd"], dcfg["n_layer"], dcfg["vocab_size"]
    mac += Ld * (U * dd * 3 * dd + 2 * U * U * dd + U * dd * dd + U * dd * dd + T2 * d * 2 * dd + 2 * U * T2 * dd + U * dd * dd + 8 * U * dd * dd) + U * dd * Vd
    return 3.0 * 2.0 * mac * B / 1e9


def config3_gflop_per_step_unpadded(cfg, dcfg, frame_lengths, label_lengths):
    """the same accounting on the work the clips actually hold: every clip with its OWN frame count (T1 = ceil(frames / 2), T' = ceil(T1 / 2): the Conv2d front end with k 3 /
    s 2 / p 1) and its own number of decoder tokens instead of the padded (2000 frames, U tokens) shape.  Several kernels skip padded tiles (key lengths, `m_valid`), so the
    padded figure credits the step with FLOPs nobody asked for — `model_frac_of_bf16_peak` keeps that figure (comparable across rounds), this one is reported beside it."""
    d, I, L, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"], cfg["vocab_size"]
    C1, C2 = cfg["conv_dim"]
    F1, F2 = 40, 20
    dd, Ld, Vd = dcfg["n_embd"], dcfg["n_layer"], dcfg["vocab_size"]
    mac = 0.0
    for fr, U in zip(frame_lengths, label_lengths):
        T1 = (int(fr) + 1) // 2
        T2 = (T1 + 1) // 2
        mac += T1 * F1 * C1 * 9 + T2 * F2 * C2 * 9 * C1 + T2 * (F2 * C2) * d + T2 * d * d
        mac += L * (2 * (2 * T2 * d * I) + 4 * T2 * d * d + (T2 * T2 * d) * 2 + T2 * (2 * T2 - 1) * d + T2 * d * I + T2 * (I // 2) * 31 + T2 * (I // 2) * d + T2 * 2 * d * 31
                    + T2 * 2 * d * d) + T2 * d * (V + 1)
        U = int(U)
        mac += Ld * (U * dd * 3 * dd + 2 * U * U * dd + U * dd * dd + U * dd * dd + T2 * d * 2 * dd + 2 * U * T2 * dd + U * dd * dd + 8 * U * dd * dd) + U * dd * Vd
    return 3.0 * 2.0 * mac / 1e9


def build_config3(args, rank, dev, B, overlap):
    """BASELINE config 3's trainer and one rank's shard (recipes_v0.0.1/librispeech_aed/train_small_baseline.sh: small encoder + 6x256 GPT-2 decoder, ctc_weight 0.3, label
    smoothing 0.1, fixed positions, AdamW 2e-3 / wd 1e-6, per-GPU batch 96, clips 1-20 s sorted into the batch); seeded weights, the same on every rank."""
    from huggingface_asr_amd.train_aed import JointAEDTrainer
    pd = float(args.dropout)
    cfg = dict(shapes.SMALL, position_embeddings_type=args.pos, ctc_zero_infinity=True, ctc_loss_reduction="mean", hidden_dropout=pd, activation_dropout=pd,
               attention_dropout=pd, final_dropout=pd, feat_proj_dropout=0.0, csgu_conv_dropout=pd, layerdrop=0.0, apply_spec_augment=False)
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    dcfg = dict(vocab_size=5000, n_embd=256, n_layer=6, n_head=4, n_positions=1024, head_locations=[], head_weights=[1.0], lsm_factor=0.1,
                layer_norm_epsilon=1e-5, pos_emb_fixed=True, tie_word_embeddings=False, resid_pdrop=pd, embd_pdrop=pd, attn_pdrop=pd)
    jcfg = dict(ctc_weight=0.3, pad_token_id=3, decoder_start_token_id=1)
    tr = JointAEDTrainer(cfg, dcfg, jcfg, dev, lr=2e-3, weight_decay=1e-6)
    for sync in (tr.enc.sync, tr.sync):                     # the two parameter stores (encoder, decoder) each own their gradient all-reduce
        sync.overlap = bool(overlap)
    tr.enc.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    for s_ in tr.store.specs.values():                      # seeded decoder weights straight into the packed store (same on every rank)
        zero = s_.name.endswith(("_b", "bqkv", "bq", "bkv", "bo", "bco", "bfc", "bpr"))
        tr.store.p(s_.name).copy_((torch.ones(s_.shape) if s_.name.endswith("_g") else torch.randn(s_.shape, generator=g) * (0.0 if zero else 0.02)).to(dev))
    tr.store.refresh_mirrors(cast=True)
    rng = np.random.default_rng(rank)
    fl = np.sort(rng.integers(100, 2001, size=B))[::-1].copy()
    T = 2000                                                # every rank pads to the same 20 s bucket (weak scaling: same work per rank)
    feats = torch.from_numpy(synth.normal(100 + rank, "feats", (B, T, 80), 1.0)).to(dev)
    lens = torch.from_numpy(fl.astype(np.int32)).to(dev)
    labels = torch.from_numpy(synth.labels(rank, B, 60, cfg["vocab_size"], lo=5)).to(dev)
    for b in range(B):
        labels[b, max(2, int(fl[b] / 100 * 3)):] = -100
    ul = (labels != -100).sum(1).cpu().numpy()
    return tr, (feats, lens, labels), fl, cfg, dcfg, ul


def measure_config3(args, world, rank, dev, PL, B, overlap, steps, warmup):
    """-> dict(ms_per_step, first_step_grad_norm, per_rank rows [ms, encoder checksum, decoder checksum], loss, audio seconds of this rank's shard, trainer, config dicts)"""
    import torch.distributed as td
    tr, (feats, lens, labels), fl, cfg, dcfg, ul = build_config3(args, rank, dev, B, overlap)
    state = {}

    def one():
        state["o"] = tr.train_step(feats, lens, labels)
    one()                                                   # untimed first step from the seeded weights: its global gradient norm (deterministic reduction of the all-reduced
    first_norm = float(state["o"]["grad_norm"])             # gradients) identifies the step's gradients independently of the schedule
    for _ in range(warmup):
        one()
    dt = PL.timed(one, steps, sync=torch.cuda.synchronize, device=dev)
    # per-rank view: every rank's own wall time per step and a checksum of its weights after the timed steps (replicas must stay bit-identical)
    t0 = time.perf_counter()
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    mine = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3, float(tr.enc.store.flat_p.double().sum()), float(tr.store.flat_p.double().sum())], dtype=torch.float64)
    buf = mine.to(dev) if (world > 1 and td.get_backend() == "nccl") else mine
    per_rank = [torch.zeros_like(buf) for _ in range(world)]
    if world > 1:
        td.all_gather(per_rank, buf)
    else:
        per_rank = [buf]
    per_rank = [t.cpu() for t in per_rank]
    return dict(dt=dt, ms_per_step=round(dt / steps * 1e3, 3), first_norm=first_norm, per_rank=per_rank, loss=float(state["o"]["loss"]), fl=fl, ul=ul, tr=tr, cfg=cfg, dcfg=dcfg)


def all_reduce_alone_ms(tr, world):
    """the step's gradient all-reduce on its own: same buffers, same message sizes, timed on the compute stream"""
    import torch.distributed as td
    if world <= 1:
        return None
    bufs = [tr.enc.store.flat_g, tr.store.flat_g]
    for _ in range(3):
        for b_ in bufs: td.all_reduce(b_)
    torch.cuda.synchronize(); td.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for b_ in bufs: td.all_reduce(b_)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 10, 3)


def train_bench(args, world, rank, dev, PL):
    """BASELINE config 3: one data-parallel training step of the joint CTC/attention model per rank shard."""
    import torch.distributed as td
    B = args.batch or 96
    m = measure_config3(args, world, rank, dev, PL, B, args.overlap, args.steps, args.warmup)
    tr, per_rank, fl, pd = m["tr"], m["per_rank"], m["fl"], float(args.dropout)
    ar_ms = all_reduce_alone_ms(tr, world)
    if rank == 0:
        n_params = tr.store.n + tr.enc.store.n
        # every rank drew its own lengths; the job's audio = sum over ranks (same distribution): use this rank's sum x world
        sec = world * float(fl.sum()) / 100.0 * args.steps
        gf = config3_gflop_per_step(m["cfg"], m["dcfg"], B, 500, 60)
        gfu = config3_gflop_per_step_unpadded(m["cfg"], m["dcfg"], fl, m["ul"])
        print(json.dumps({"metric": "audio-seconds/sec, joint CTC/attention TRAINING step (fwd+bwd+AdamW), ED-small, DP", "value": round(sec / m["dt"], 1),
                          "unit": "audio-seconds/sec (un-padded audio)", "n_gpus": world, "rccl_ranks": td.get_world_size() if td.is_initialized() else 1,
                          "backend": td.get_backend() if td.is_initialized() else "none", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "synthetic",
                          "config": {"workload": "BASELINE config 3: small E-Branchformer encoder + 6x256 GPT-2 decoder, joint CTC/attention loss, AdamW; "
                                                 f"{B} clips of 1-20 s per GPU padded to 2000 frames", "per_gpu_batch": B, "frames": 2000, "n_params": n_params, "dropout": pd,
                                     "parallelism": f"dp{world}: SUM all-reduce of {n_params * 4 / 1e6:.0f} MB fp32 gradients after the backward",
                                     "loss": round(m["loss"], 4)},
                          "schedule": "overlap: one async all-reduce per layer range, launched while the backward of the earlier layers runs" if args.overlap
                                      else "after-backward: one all-reduce per parameter store once the backward is done",
                          "first_step_grad_norm": m["first_norm"], "ms_per_step_by_rank": [round(float(t[0]), 3) for t in per_rank],
                          "weights_checksum_by_rank": [[repr(float(t[1])), repr(float(t[2]))] for t in per_rank],
                          "replicas_identical": all(bool(torch.equal(t[1:], per_rank[0][1:])) for t in per_rank),
                          "all_reduce_ms": ar_ms, "all_reduce_note": "the step's gradient all-reduce timed on its own (same buffers), per step" if ar_ms else None,
                          "model_gflop_per_step": round(gf, 1), "model_tflops_per_gpu": round(gf / (m["dt"] / args.steps) / 1e3, 1),
                          "model_frac_of_bf16_peak": round(gf / (m["dt"] / args.steps) / 1e3 / PEAK_BF16_TFLOPS, 4),
                          "model_gflop_per_step_unpadded": round(gfu, 1), "model_frac_of_bf16_peak_unpadded": round(gfu / (m["dt"] / args.steps) / 1e3 / PEAK_BF16_TFLOPS, 4),
                          "flop_note": "padded = every clip counted at the 2000-frame / 60-token bucket it is padded to; unpadded = every clip at its own length",
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
    if world > 1:
        td.barrier(); td.destroy_process_group()


def train_dp_record(args, world, rank, dev, PL):
    """world > 1, inside the DEFAULT command: the config-3 training step under BOTH gradient all-reduce schedules of `GradSync` (0 = one collective per parameter store after
    the backward; 1 = one async all-reduce per layer range, overlapped with the backward — what DDP's bucket hooks give the reference), each from the same seeded weights, so
    that the run that measures the 1 -> N forward curve also exercises the 516-MB-class gradient all-reduce over RCCL.  Every rank takes part (collectives); rank 0 reports.
    The backward has no float atomics (round 4): `first_step_grad_norm` and the weight checksums of the two schedules are comparable for EQUALITY."""
    B = args.train_dp_batch
    out = {"per_gpu_batch": B, "steps": args.train_dp_steps, "ms_per_step": [], "first_step_grad_norm": [], "replicas_identical": [], "weights_checksum_rank0": []}
    ar = None
    for ov in (0, 1):
        m = measure_config3(args, world, rank, dev, PL, B, ov, args.train_dp_steps, 1)
        out["ms_per_step"].append(m["ms_per_step"])
        out["first_step_grad_norm"].append(m["first_norm"])
        out["replicas_identical"].append(all(bool(torch.equal(t[1:], m["per_rank"][0][1:])) for t in m["per_rank"]))
        out["weights_checksum_rank0"].append([repr(float(m["per_rank"][0][1])), repr(float(m["per_rank"][0][2]))])
        if ov == 0:
            ar = all_reduce_alone_ms(m["tr"], world)
            out["n_params"] = m["tr"].store.n + m["tr"].enc.store.n
        del m
        torch.cuda.empty_cache()
    out["all_reduce_ms"] = ar
    out["schedules_agree"] = out["first_step_grad_norm"][0] == out["first_step_grad_norm"][1] and out["weights_checksum_rank0"][0] == out["weights_checksum_rank0"][1]
    out["schedules"] = ["after-backward: one all-reduce per parameter store", "overlap: one async all-reduce per layer range during the backward"]
    return out


# ------------------------------------------------------------------------------------------------------------------ secondary records (configs 3, 4, 5)
def secondary(name, args):
    """One bounded measurement of another BASELINE.json config on cuda:0, as its own process (`bench.py --secondary NAME`, started by the default run after its timed region:
    whatever happens here cannot take the headline line down).  Prints one JSON line."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from huggingface_asr_amd import parallel as PL
    if name == "whisper":                                    # config 4: Whisper-small encoder, 16 x 30 s, log-mel on the device
        from huggingface_asr_amd.whisper import WhisperEncoderEngine, WhisperFrontend
        d, F, Bw = 768, 3072, 16
        sd = {}
        for n, shp in [("conv1.weight", (d, 80, 3)), ("conv1.bias", (d,)), ("conv2.weight", (d, d, 3)), ("conv2.bias", (d,)), ("embed_positions.weight", (1500, d)),
                       ("layer_norm.weight", (d,)), ("layer_norm.bias", (d,))]:
            sd[n] = torch.from_numpy(synth.init_param(0, n, shp))
        for l in range(12):
            for n, shp in [("self_attn_layer_norm.weight", (d,)), ("self_attn_layer_norm.bias", (d,)), ("self_attn.q_proj.weight", (d, d)), ("self_attn.q_proj.bias", (d,)),
                           ("self_attn.k_proj.weight", (d, d)), ("self_attn.v_proj.weight", (d, d)), ("self_attn.v_proj.bias", (d,)), ("self_attn.out_proj.weight", (d, d)),
                           ("self_attn.out_proj.bias", (d,)), ("final_layer_norm.weight", (d,)), ("final_layer_norm.bias", (d,)), ("fc1.weight", (F, d)), ("fc1.bias", (F,)),
                           ("fc2.weight", (d, F)), ("fc2.bias", (d,))]:
                sd[f"layers.{l}.{n}"] = torch.from_numpy(synth.init_param(0, f"layers.{l}.{n}", shp))
        eng = WhisperEncoderEngine(dict(d_model=d, encoder_layers=12, encoder_attention_heads=12, encoder_ffn_dim=F), dev)
        eng.load_state_dict(sd)
        wave = torch.from_numpy(synth.waveforms(5, Bw, 480000)).to(dev)
        fe = WhisperFrontend(80)

        def step():
            _, cl = fe(wave, want_features=False)
            return eng.forward(features_cl=cl)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize(); dt1 = (time.perf_counter() - t0) / 8
        dt3 = dt1
        if args.streams > 1:                                 # (`--secondary whisper --streams 1`: one batch at a time only — what a kernel trace of this command should show)
            streams = [torch.cuda.Stream() for _ in range(3)]
            eng.tail_split = False                            # the other batches' blocks fill an under-filled last round of tiles: no split of the GEMMs (gemm_glds.hip)

            def run(n):
                for j in range(n):
                    with torch.cuda.stream(streams[j % 3]):
                        step()
            run(6); torch.cuda.synchronize(); t0 = time.perf_counter(); run(18); torch.cuda.synchronize(); dt3 = (time.perf_counter() - t0) / 18
        gf = 2.0 * Bw * 1500 * (12 * (4 * d * d + 2 * d * F + 2 * 1500 * d) + 3 * d * d + 2 * 3 * 80 * d) / 1e9
        return {"config": "BASELINE config 4: Whisper-small encoder (12 x 768, 12 heads, FFN 3072) + log-mel, 16 x 30 s per step, bf16, random weights", "ms_per_step": round(dt1 * 1e3, 3),
                "value": round(Bw * 30 / dt1, 1), "unit": "audio-seconds/sec", "three_batches_in_flight_ms_per_step": round(dt3 * 1e3, 3),
                "three_batches_in_flight_value": round(Bw * 30 / dt3, 1), "model_tflops": round(gf / dt1 / 1e3, 1), "model_frac_of_bf16_peak": round(gf / dt1 / 1e3 / PEAK_BF16_TFLOPS, 4)}
    if name == "decode":                                     # config 5: DeCRED_base-shaped joint model, bs = 1, one 10 s clip, 40 tokens
        from huggingface_asr_amd.decoder import JointAEDEngine, generate
        enc_cfg = dict(shapes.BASE, vocab_size=5000, ctc_zero_infinity=True, ctc_loss_reduction="mean")
        D, V = 512, 5001
        dec_cfg = dict(vocab_size=V, n_embd=D, n_layer=8, n_head=8, n_positions=256, head_locations=[5], head_weights=[0.4, 0.6], lsm_factor=0.1, pos_emb_fixed=True)
        jcfg = dict(ctc_weight=0.3, pad_token_id=5000, decoder_start_token_id=2)
        sd = {"encoder." + k: torch.from_numpy(synth.init_param(0, "encoder." + k, shp)) for k, shp in shapes.param_shapes(enc_cfg).items()}

        def P(n, shp):
            sd[n] = torch.from_numpy(synth.init_param(0, n, shp))
        P("decoder.transformer.wte.emb_layers.0.weight", (V, D))
        for l in range(8):
            for n, shp in [("ln_1.weight", (D,)), ("ln_1.bias", (D,)), ("attn.c_attn.weight", (D, 3 * D)), ("attn.c_attn.bias", (3 * D,)), ("attn.c_proj.weight", (D, D)),
                           ("attn.c_proj.bias", (D,)), ("ln_cross_attn.weight", (D,)), ("ln_cross_attn.bias", (D,)), ("crossattention.q_attn.weight", (D, D)),
                           ("crossattention.q_attn.bias", (D,)), ("crossattention.c_attn.weight", (D, 2 * D)), ("crossattention.c_attn.bias", (2 * D,)),
                           ("crossattention.c_proj.weight", (D, D)), ("crossattention.c_proj.bias", (D,)), ("ln_2.weight", (D,)), ("ln_2.bias", (D,)),
                           ("mlp.c_fc.weight", (D, 4 * D)), ("mlp.c_fc.bias", (4 * D,)), ("mlp.c_proj.weight", (4 * D, D)), ("mlp.c_proj.bias", (D,))]:
                P(f"decoder.transformer.h.{l}.{n}", shp)
        P("decoder.transformer.ln_f.weight", (D,)); P("decoder.transformer.ln_f.bias", (D,))
        P("decoder.lm_head.weight", (V, D)); P("decoder.additional_lm_heads.0.weight", (V, D))
        eng = JointAEDEngine(enc_cfg, dec_cfg, jcfg, dev)
        eng.load_state_dict(sd)
        wave = torch.from_numpy(synth.waveforms(1, 1, 160000)).to(dev)
        tb = FB.FbankTables(80)
        rec = {"config": "BASELINE config 5: DeCRED_base-shaped joint model (E-Branchformer-base + 8 x 512 GPT-2, auxiliary head), bs = 1, one 10 s clip, device-resident "
                         "decoding loop, joint CTC / attention scoring (ctc_weight 0.3), 40 tokens (random weights: fixed-length decode)"}
        feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
        for W in (1, 5):                                      # one untimed decode per mode (first-use costs: kernel images, workspaces, the prefix scorer's buffers)
            generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=0.3, eos_token_id=1)
        for W, key in ((1, "greedy"), (5, "beam5")):
            best, n = 1e9, 0
            for _ in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                feats, frames = FB.fbank_gpu(wave, tb, pad_frames_to=100)
                out = generate(eng, feats, frames, num_beams=W, max_length=40, ctc_weight=0.3, eos_token_id=1)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                n = len(out[0]["tokens"])
            rec[key + "_end_to_end_ms"] = round(best * 1e3, 2)
            rec[key + "_tokens"] = n
            rec[key + "_ms_per_token_incl_encoder"] = round(best * 1e3 / max(n - 1, 1), 3)
        return rec
    if name == "train":                                      # config 3 on one GPU (the data-parallel form is `bench.py --train --gpus N`)
        m = measure_config3(args, 1, 0, dev, PL, 96, 0, 5, 2)
        gf = config3_gflop_per_step(m["cfg"], m["dcfg"], 96, 500, 60)
        gfu = config3_gflop_per_step_unpadded(m["cfg"], m["dcfg"], m["fl"], m["ul"])
        sec = float(m["fl"].sum()) / 100.0
        return {"config": "BASELINE config 3 on one GPU: small E-Branchformer encoder + 6x256 GPT-2 decoder, joint CTC/attention loss, fwd + bwd + AdamW, 96 clips of 1-20 s "
                          f"padded to 2000 frames, dropout {args.dropout}", "ms_per_step": m["ms_per_step"], "value": round(sec / (m["dt"] / 5), 1),
                "unit": "audio-seconds/sec (un-padded audio)", "model_gflop_per_step": round(gf, 1), "model_tflops": round(gf / (m["dt"] / 5) / 1e3, 1),
                "model_frac_of_bf16_peak": round(gf / (m["dt"] / 5) / 1e3 / PEAK_BF16_TFLOPS, 4),
                "model_gflop_per_step_unpadded": round(gfu, 1), "model_frac_of_bf16_peak_unpadded": round(gfu / (m["dt"] / 5) / 1e3 / PEAK_BF16_TFLOPS, 4),
                "flop_note": "padded = every clip counted at the 2000-frame / 60-token bucket it is padded to; unpadded = every clip at its own length (several kernels skip padded tiles)",
                "loss": round(m["loss"], 4)}
    raise SystemExit(f"bench.py: unknown --secondary {name}")


def run_secondaries(args):
    """configs 4, 5 and 3 as child processes after the timed region (each bounded by a timeout; a failure becomes an `error` entry, never an exception here)"""
    import subprocess
    out = {}
    for name in ("whisper", "decode", "train"):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--secondary", name, "--pos", args.pos], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                               timeout=args.secondary_timeout, env=dict(os.environ))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[name] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"error": f"rc={r.returncode}: " + (r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "no output")}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def main():
    args = parse_args()
    from huggingface_asr_amd import parallel as PL
    from huggingface_asr_amd.pipeline import reserve_hw_queues
    hwq = 4 if args.train else reserve_hw_queues(args.streams)          # before the first HIP call of this process (and inherited by the ranks it starts)
    in_torchrun = "WORLD_SIZE" in os.environ
    if args.secondary:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py --secondary: no GPU visible (HIP-only path)")
        print(json.dumps(secondary(args.secondary, args)), flush=True)
        return
    if not in_torchrun and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world, rank, local = PL.env_world()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); pass the same number to both")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible. The hot path is HIP-only (no CPU fallback); use --dry-run to check the launcher on a GPU-less box.")
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("bench.py: --share-gpu puts every rank on cuda:0, which RCCL refuses (one device per rank): use --backend gloo")
    dev = torch.device("cuda", 0 if args.share_gpu else local)
    torch.cuda.set_device(dev)
    PL.init(args.backend, dev if args.backend == "nccl" else None)      # nccl = RCCL; no-op for one process
    dist = world > 1
    import torch.distributed as td
    if args.train:
        return train_bench(args, world, rank, dev, PL)

    cfg = dict(shapes.BASE, position_embeddings_type=args.pos, ctc_zero_infinity=True, ctc_loss_reduction="mean")
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict_numpy(shapes.param_shapes(cfg), 0).items()}
    # k engines (own workspace, same weights) on k HIP streams: step j runs on lane j % k (huggingface_asr_amd/pipeline.py).  A step is still one pass over one batch of B
    # clips; what overlaps is the tail of one step's kernels with the ramp of the other's (one tile per CU leaves every launch with a fill and a drain nothing else covers).
    from huggingface_asr_amd.pipeline import ForwardPipeline
    nstr = max(1, args.streams)
    pipe = ForwardPipeline(cfg, dev, sd, lanes=nstr, wide_tiles=bool(args.wide_tiles))
    eng = pipe.engines[0]
    if args.wide_tiles:                      # the one-step-at-a-time comparison and the per-kernel roofline leg run the product's own tiles (wide tiles are for steps in flight)
        eng = EBranchformerEngine(cfg, dev)
        eng.load_state_dict(sd)
    if args.head_lse:
        for e_ in list(pipe.engines) + [eng]:
            e_.head_lse = True
    B = args.batch or BATCH
    wave = torch.from_numpy(synth.waveforms(100 + rank, B, SR * SECONDS)).to(dev)       # resident in HBM
    labels = torch.from_numpy(synth.labels(rank, B, U, cfg["vocab_size"])).to(dev)
    tables = FB.FbankTables(80)
    tables.device(dev)

    def step(e=eng, w=wave, lab=labels):
        feats, frames = FB.fbank_gpu(w, tables, pad_frames_to=100)
        out = e.forward(feats, frames, want_hidden=False)
        loss, _, _ = ops.ctc_loss(out["logits"], lab, out["outer_len"], reduction="mean", zero_infinity=True, lse=out["lse"])     # lse: None (a pass over the logits inside ctc_loss) unless --head-lse
        return loss

    batches = [(wave, labels)] + [(torch.from_numpy(synth.waveforms(100 + rank + 1000 * i, B, SR * SECONDS)).to(dev),
                                   torch.from_numpy(synth.labels(rank + 1000 * i, B, U, cfg["vocab_size"])).to(dev)) for i in range(1, nstr)]

    def step_pipelined():
        return pipe.submit(lambda e, lane: step(e, *batches[lane]))

    L = _lib.lib()
    use_events = not args.no_kernel_events and rank == 0
    n_gemm_per_step = 4 + cfg["num_hidden_layers"] * 10 + 1
    for _ in range(max(args.warmup, nstr)):
        step_pipelined()
    torch.cuda.synchronize()
    pipe.reset()
    state = {}

    def one():
        lane, v = step_pipelined()
        if lane == 0:
            state["loss"] = v                # lane 0 is the batch every earlier round's line reports the loss of
    dt = PL.timed(one, args.steps, sync=torch.cuda.synchronize, device=dev)      # barrier + sync both sides, MAX over ranks
    loss_v = float(state["loss"])
    single = None
    if nstr > 1 and not args.no_one_step:    # the same K steps strictly one at a time (outside the timed region): what the pipelining buys
        for _ in range(2):
            step()
        dt1 = PL.timed(lambda: step(), args.steps, sync=torch.cuda.synchronize, device=dev)
        single = dict(ms_per_step=round(dt1 / args.steps * 1e3, 3), value=round(world * B * SECONDS * args.steps / dt1, 1),
                      same_loss_bits=bool(float(step()) == loss_v), loss=float(step()))        # lane 0's batch alone on the default stream against its last pipelined pass

    # ---- roofline leg, AFTER the timed region: every dense contraction launch of `event_steps` further steps is bracketed by HIP events recorded on the
    # launch stream (mi_profile_*; the nn.Linear GEMMs and the implicit-GEMM conv), achieved = sum of their algorithmic FLOPs / sum of their durations
    roof = None
    if use_events:
        _lib.check(L.mi_profile_create(n_gemm_per_step * args.event_steps + 64), "mi_profile_create")
        L.mi_profile_reset(); L.mi_profile_enable(1)
        for _ in range(args.event_steps):
            step()
        torch.cuda.synchronize()
        L.mi_profile_enable(0)
        ms, fl = C.c_double(0), C.c_double(0)
        _lib.check(L.mi_profile_summary(C.byref(ms), C.byref(fl)), "mi_profile_summary")
        n = L.mi_profile_count()
        if n and ms.value > 0:
            # FLOP accounting + event-timed durations: every dense launch of `event_steps` steps carries a hipExtLaunchKernelGGL (start, stop) event pair (csrc/gemm_args.hpp).
            # Un-profiled, that pair reads within ~1 % of the dispatch's begin -> end time rocprofv3 reports for the same kernel in the pipelined step (r03: 3266 vs 3260 us
            # per step over the 148 launches); with a profiler attached to THIS process it reads ~4 us per launch high.  Nothing is subtracted.
            per_step, fl_step, iso_us = {}, {}, {}
            for f in range(len(FAMILIES)):
                fms, ffl, fn = C.c_double(0), C.c_double(0), C.c_int(0)
                _lib.check(L.mi_profile_summary_family(f, C.byref(fms), C.byref(ffl), C.byref(fn)), "mi_profile_summary_family")
                if fn.value:
                    per_step[f] = fn.value // args.event_steps
                    fl_step[f] = ffl.value / args.event_steps
                    iso_us[f] = fms.value * 1e3 / args.event_steps
            # PIPELINED durations (the roofline figure): the dispatch timestamps of the same kernels in a rocprofv3 --kernel-trace child run of this script
            try:
                pipe_us, child_steps, why = (None, 0, "more than one rank") if world > 1 else rocprof_child_dense_us(args, B, per_step)
            except Exception as e:  # noqa: BLE001 — the bench line must come out whatever happens to the profiler child
                pipe_us, child_steps, why = None, 0, f"rocprofv3 child: {type(e).__name__}: {e}"[:200]
            dur = pipe_us if pipe_us is not None else iso_us
            tot_fl, tot_us = sum(fl_step.values()), sum(dur[f] for f in per_step)
            ach = tot_fl / (tot_us * 1e-6) / 1e12
            fams = {}
            for f in per_step:
                fams[FAMILIES[f]] = dict(launches_per_step=per_step[f], avg_us=round(dur[f] / per_step[f], 2), us_per_step=round(dur[f], 1), gflop_per_step=round(fl_step[f] / 1e9, 1),
                                         tflops=round(fl_step[f] / (dur[f] * 1e-6) / 1e12, 1), frac=round(fl_step[f] / (dur[f] * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                         event_timed_avg_us=round(iso_us[f] / per_step[f], 2))
            traffic, tsrc = pmc_traffic_bytes()
            roof = dict(bound="mfma", kernel="all dense contraction launches of the step (per-family split in `families`)",
                        achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=traffic,
                        traffic_unit="HBM bytes per launch of gemm8p_kernel<false,1> (PMC FETCH_SIZE x2 + WRITE_SIZE)", traffic_source=tsrc,
                        launches_per_step=sum(per_step.values()), avg_launch_us=round(tot_us / sum(per_step.values()), 2), dense_us_per_step=round(tot_us, 1),
                        gflop_per_step=round(tot_fl / 1e9, 1), families=fams,
                        event_timed_tflops=round(tot_fl / (sum(iso_us.values()) * 1e-6) / 1e12, 1),
                        measured=(f"dispatch begin -> end timestamps of the dense kernels over {child_steps} back-to-back steps of a rocprofv3 --kernel-trace --stats child run of this script "
                                  "(= what `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-kernel-events` reports); `event_timed_*` = the same launches of this process, each "
                                  "carrying a hipExtLaunchKernelGGL (start, stop) event pair, separate pass after the timed region, nothing subtracted.  Both legs run ONE step at a time "
                                  "(the child with --streams 1): with steps pipelined over streams kernels of different steps share the chip and a dispatch's duration is no longer its own work") if pipe_us is not None else
                                 (f"FALL-BACK ({why}): every dense launch carries a hipExtLaunchKernelGGL (start, stop) event pair, separate pass after the timed region, nothing subtracted "
                                  "(reads within ~1 % of rocprofv3's kernel durations when no profiler is attached to this process, ~4 us per launch high under one)"))

    if rank == 0:
        audio_s = world * B * SECONDS * args.steps
        T2 = eng.out_frames(1000)
        rec = {
            "metric": "audio-seconds/sec encoder fwd+CTC, E-Branchformer-base", "value": round(audio_s / dt, 1),
            "unit": "audio-seconds/sec", "n_gpus": world, "rccl_ranks": td.get_world_size() if td.is_initialized() else 1,
            "backend": td.get_backend() if td.is_initialized() else "none", "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"E-Branchformer-base enc+CTC ({args.pos}-pos), {B}x{SECONDS}s 16kHz clips/GPU: "
                                   "fbank+CMVN -> conv2d sub -> 16 layers -> CTC head -> CTC loss"
                                   + (" (the batch-independent projection of the relative-position table, 0.2 % of the FLOPs, is cached across steps)" if args.pos == "relative" else ""),
                       "per_gpu_batch": B, "frames": 1000, "encoder_frames": T2, "parallelism": f"replicas x{world} (no exchange step)",
                       "steps_in_flight": nstr, "wide_tiles": bool(args.wide_tiles), "hw_queues": hwq, "one_step_at_a_time": single,
                       "algorithmic_gflop_per_audio_s": round(algorithmic_gflop_per_utt(cfg, T2) / SECONDS, 3),
                       "ctc_loss": round(loss_v, 4)},
            "roofline": roof,
        }
        if roof is not None:
            # the headline's own mode: dense GFLOP of a step / the timed region's wall per step (every non-GEMM kernel, gap and ramp included), and — from a kernel trace of the
            # same mode — how the wall divides between dense kernels, other kernels and nothing
            dense_tf = roof["gflop_per_step"] / (dt / args.steps) / 1e3
            inf = dict(steps_in_flight=nstr, wide_tiles=bool(args.wide_tiles), dense_gflop_per_step=roof["gflop_per_step"], wall_ms_per_step=round(dt / args.steps * 1e3, 3),
                       dense_tflops_over_wall=round(dense_tf, 1), frac_of_bf16_peak=round(dense_tf / PEAK_BF16_TFLOPS, 4))
            if world == 1 and not args.no_in_flight_trace:
                try:
                    tr_ = rocprof_child_in_flight(args, B)
                except Exception as e:  # noqa: BLE001 — never take the line down
                    tr_ = {"error": f"{type(e).__name__}: {e}"[:200]}
                if "error" in tr_:
                    inf["trace"] = tr_
                else:
                    inf["trace"] = dict(traced_wall_ms_per_step=tr_["wall_ms_per_step"], wall_share_dense_running=tr_["wall_share_dense_running"],
                                        wall_share_only_other_kernels=tr_["wall_share_only_other_kernels"], wall_share_idle=tr_["wall_share_idle"],
                                        non_gemm_share_of_busy=tr_["non_gemm_share_of_busy"], dense_busy_us_per_wall_ms=tr_["dense_busy_us_per_wall_ms"],
                                        other_busy_us_per_wall_ms=tr_["other_busy_us_per_wall_ms"],
                                        busy_us_per_step_by_family={k: v["busy_us_per_step"] for k, v in tr_["families"].items()},
                                        note="rocprofv3 --kernel-trace child of this command in the headline's mode (tools/lanes_trace.py); busy = sum of dispatch durations, which "
                                             "overlap across lanes; tracing lengthens the wall, the shares are what it is for")
            roof["in_flight"] = inf
        rec["model_tflops_per_gpu"] = round(rec["config"]["algorithmic_gflop_per_audio_s"] * rec["value"] / world / 1e3, 2)
        # flat copies of what `config` nests (a reader of the parsed line sees the headline's mode and the single-step latency beside it: ADVICE r3)
        rec["mode"] = f"throughput: {nstr} independent steps in flight on {nstr} HIP streams" if nstr > 1 else "one step at a time"
        rec["steps_in_flight"] = nstr
        rec["one_step_ms"] = single["ms_per_step"] if single else rec["ms_per_step"]
        rec["one_step_value"] = single["value"] if single else rec["value"]
        rec["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfg, sd)
    if world > 1 and not args.no_train_dp:
        # The forward curve is what a multi-GPU run of this command is for: its line goes to stdout BEFORE any training collective starts, and the line with `train_dp`
        # merged in follows as the LAST line.  A rank that fails inside `train_dp` leaves at once with a non-zero code — it never waits in a collective, and the launcher
        # (torch.distributed.run) takes the other ranks down with it; stdout keeps the forward line.  Collectives time out after PL.COLLECTIVE_TIMEOUT_S.
        if rank == 0:
            print(json.dumps(dict(rec, train_dp=None, secondary=None, note="forward headline; the line with `train_dp` follows as the last line")), flush=True)
        import gc
        pipe = eng = batches = step = step_pipelined = one = wave = labels = state = None      # the closures above held the forward engines and their batches alive
        gc.collect()
        torch.cuda.empty_cache()
        try:
            if os.environ.get("HFASR_BENCH_FAIL_RANK") == str(rank):                              # test knob: a rank-local failure at the worst moment
                raise RuntimeError("injected failure (HFASR_BENCH_FAIL_RANK)")
            train_dp = train_dp_record(args, world, rank, dev, PL)
        except BaseException as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: train_dp failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(3)
        if rank == 0:
            rec["train_dp"] = train_dp
            rec["secondary"] = None
            print(json.dumps(rec), flush=True)
    elif rank == 0:
        rec["train_dp"] = None
        rec["secondary"] = None
        if world == 1 and not args.no_secondary:
            del pipe
            torch.cuda.empty_cache()
            rec["secondary"] = run_secondaries(args)
        print(json.dumps(rec), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
