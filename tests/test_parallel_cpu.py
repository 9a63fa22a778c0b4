"""CPU, world_size 2 over gloo: the N>1 plumbing bench.py uses (sharding, barrier-bracketed timing with MAX over ranks,
loss averaging).  The forward path itself has no collective (replicas)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    import time

    from huggingface_asr_amd import parallel as P
    w, r, _ = P.init("gloo")
    assert (w, r) == (world, rank)
    lo, hi = P.shard_range(7, rank, world)
    loss = torch.tensor(float(rank + 1))                  # stand-in for a per-rank batch-mean loss
    mean = float(P.mean_over_ranks(loss))
    dt = P.timed(lambda: time.sleep(0.01 * (rank + 1)), 3)  # rank 1 is slower: everyone must report ITS time
    out[rank] = (lo, hi, mean, dt)
    P.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_world2_sharding_timing_and_loss_mean():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (lo0, hi0, m0, t0), (lo1, hi1, m1, t1) = out[0], out[1]
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)           # balanced, contiguous, covering
    assert m0 == m1 == pytest.approx(1.5)
    assert t0 == pytest.approx(t1, abs=1e-9) and t0 >= 0.06   # MAX over ranks: 3 x 20 ms of the slow rank


def test_shard_range_properties():
    from huggingface_asr_amd.parallel import shard_range
    for n in (0, 1, 5, 32, 33):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself; --dry-run: gloo, no HIP work, on this GPU-less box.
    A --gpus that disagrees with the launcher's world size is refused."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                    # rank 0 prints ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dry_run"] is True and rec["value"] is None and rec["steps"] == 2
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(root, "bench.py"), "--gpus", "3", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 3 but the launcher started 2" in (r.stdout + r.stderr)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in (r.stdout + r.stderr)      # the real bench never falls back to the CPU


def test_bench_roofline_traffic_comes_from_the_committed_pmc_table():
    """bench.py's `roofline.traffic` is read from the newest profiles/*pmc*per_launch*.txt by kernel name: a renamed kernel (a new template parameter) must not
    silently turn it into null"""
    import importlib.util, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    traffic, src = mod.pmc_traffic_bytes()
    assert traffic is not None and 30e6 < traffic < 120e6 and src.startswith("profiles/"), (traffic, src)


def test_unpadded_flop_accounting_equals_the_padded_one_on_full_length_clips():
    """bench.py's config-3 FLOP accounting: counting every clip at its own length must reproduce the padded figure when every clip fills the bucket,
    and must be smaller — by about half for uniform 1-20 s clips — otherwise."""
    import importlib.util
    import os
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from huggingface_asr_amd import shapes
    cfg, dcfg = dict(shapes.SMALL), dict(vocab_size=5000, n_embd=256, n_layer=6)
    full = bench.config3_gflop_per_step(cfg, dcfg, 96, 500, 60)
    assert abs(bench.config3_gflop_per_step_unpadded(cfg, dcfg, [2000] * 96, [60] * 96) - full) < 1e-6 * full
    fl = np.sort(np.random.default_rng(0).integers(100, 2001, size=96))[::-1]
    part = bench.config3_gflop_per_step_unpadded(cfg, dcfg, fl, [max(2, int(f / 100 * 3)) for f in fl])
    assert 0.4 * full < part < 0.65 * full


def test_lanes_trace_summary_on_a_synthetic_trace(tmp_path):
    """tools/lanes_trace.py (the `roofline.in_flight.trace` block): wall shares by interval union, busy time by family, on a hand-made two-lane trace."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lanes_trace
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X,Workgroup_Size_X"]
    t = 0
    for step in range(6):                                  # per step: fbank 10 us, a GEMM 100 us overlapping a dwconv (from the other lane) for 50 us, 20 us of nothing
        rows.append(f"fbank_kernel,{t},{t + 10000},1024,256")
        rows.append(f"\"void gemm8p_kernel<false, 1, false>(GemmArgs)\",{t + 10000},{t + 110000},131072,512")
        rows.append(f"\"void dwconv31_kernel<true>(DwArgs)\",{t + 60000},{t + 130000},2048,256")
        t += 150000
    p = tmp_path / "kernel_trace.csv"
    p.write_text("\n".join(rows) + "\n")
    rec = lanes_trace.summarise([str(p)], skip=0.0)
    assert rec["steps"] == 5                                # from the first fbank launch to the last one
    assert abs(rec["wall_share_dense_running"] - 100 / 150) < 1e-3 and abs(rec["wall_share_only_other_kernels"] - 30 / 150) < 1e-3 and abs(rec["wall_share_idle"] - 20 / 150) < 1e-3
    assert abs(rec["non_gemm_share_of_busy"] - 80 / 180) < 1e-3
    assert rec["families"]["dense: 256x256 GEMMs (FFN in, cgMLP in, QKV)"]["launches_per_step"] == 1.0
