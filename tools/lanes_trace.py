"""The headline's own mode as a kernel trace: `rocprofv3 --kernel-trace` of the DEFAULT command (four steps in flight, wide tiles) summarised per kernel family as busy
microseconds per wall millisecond, with the share of the wall in which dense (GEMM / implicit-GEMM) kernels, only other kernels, or nothing at all was running.

    python tools/lanes_trace.py <kernel_trace.csv> [--skip-frac 0.4] [--json OUT]        (trace of `bench.py --no-one-step --no-kernel-events --no-secondary --no-cpu-baseline`)

With steps in flight the dispatches of different steps overlap, so a family's busy time is the sum of its dispatch durations (can exceed the wall) and the wall is
covered by the UNION of the intervals.  The window is the steady state: from the first fbank launch after `skip-frac` of all fbank launches to the last one.
"""
import collections
import csv
import json
import sys

def family(name, blocks=0):
    n = name
    if "gemm8p128" in n:
        return "dense: N = d GEMMs on 128x128 tiles"
    if "gemm8p_kernel<true" in n:
        return "dense: conv2 implicit GEMM"
    if "gemm8p_kernel<false, 0, true" in n:        # the fp32-out instance: the CTC head (N = 5001: 640 blocks) and, with wide tiles, the N = d residual GEMMs (64 blocks)
        return "dense: CTC head (fp32 out)" if blocks > 128 else "dense: N = d GEMMs on 256x256 tiles (wide tiles, fp32 + residual)"
    if "gemm8p_kernel" in n:
        return "dense: 256x256 GEMMs (FFN in, cgMLP in, QKV)" if blocks > 128 else "dense: N = d GEMMs on 256x256 tiles (wide tiles, bf16 out)"
    if "gemm_glds" in n or "gemm_bf16" in n:
        return "dense: other GEMM kernels"
    if "attn" in n:
        return "attention"
    if "dwconv" in n or "row_stats" in n:
        return "depthwise convs + CSGU row statistics"
    if "ln_" in n or "layernorm" in n:
        return "LayerNorm"
    if "fbank" in n or "cmvn" in n or "trim" in n:
        return "log-mel + CMVN"
    if "conv2d_first" in n:
        return "Conv2d #1"
    if "ctc" in n or "row_lse" in n or "lse" in n:
        return "CTC loss"
    return "other"


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def summarise(paths, skip=0.4):
    """-> the record described in the module docstring, over the kernel_trace CSV(s) of one run"""
    rows = [r for f in paths for r in csv.DictReader(open(f))]
    fb = sorted(int(r["Start_Timestamp"]) for r in rows if "fbank_kernel" in r["Kernel_Name"])
    lo = fb[int(len(fb) * skip)]
    hi = fb[-1]
    nsteps = sum(1 for t in fb if lo <= t < hi)
    fam = collections.defaultdict(lambda: [0, 0])
    dense, anyk = [], []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < lo or s >= hi:
            continue
        try:
            blocks = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
        except (TypeError, ValueError):
            blocks = 0
        f = family(r["Kernel_Name"], blocks)
        fam[f][0] += 1
        fam[f][1] += e - s
        (dense if f.startswith("dense") else anyk).append((s, e))
    wall = hi - lo
    u_dense = union(list(dense))
    u_all = union(dense + anyk)
    busy_dense = sum(v[1] for k, v in fam.items() if k.startswith("dense"))
    busy_other = sum(v[1] for k, v in fam.items() if not k.startswith("dense"))
    return dict(steps=nsteps, wall_ms_per_step=round(wall / nsteps / 1e6, 4),
                wall_share_dense_running=round(u_dense / wall, 4), wall_share_only_other_kernels=round((u_all - u_dense) / wall, 4), wall_share_idle=round(1 - u_all / wall, 4),
                dense_busy_us_per_wall_ms=round(busy_dense / wall * 1e3, 1), other_busy_us_per_wall_ms=round(busy_other / wall * 1e3, 1),
                non_gemm_share_of_busy=round(busy_other / (busy_dense + busy_other), 4),
                families={k: dict(launches_per_step=round(v[0] / nsteps, 1), avg_us=round(v[1] / v[0] / 1e3, 2), busy_us_per_step=round(v[1] / nsteps / 1e3, 1),
                                  busy_us_per_wall_ms=round(v[1] / wall * 1e3, 1)) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])})


def main():
    a = sys.argv[1:]
    skip, out = 0.4, None
    if "--skip-frac" in a:
        i = a.index("--skip-frac"); skip = float(a[i + 1]); del a[i:i + 2]
    if "--json" in a:
        i = a.index("--json"); out = a[i + 1]; del a[i:i + 2]
    rec = summarise(a[:1], skip)
    print(f"{rec['steps']} steps in the window: wall {rec['wall_ms_per_step']:.3f} ms per step; a dense kernel is running {100 * rec['wall_share_dense_running']:.1f} % of the wall, "
          f"only other kernels {100 * rec['wall_share_only_other_kernels']:.1f} %, nothing {100 * rec['wall_share_idle']:.1f} %")
    print(f"busy microseconds per wall millisecond (sum of dispatch durations; > 1000 = overlap): dense {rec['dense_busy_us_per_wall_ms']}, other {rec['other_busy_us_per_wall_ms']} "
          f"(non-GEMM share of busy time {100 * rec['non_gemm_share_of_busy']:.1f} %)")
    for k, v in rec["families"].items():
        print(f"{v['launches_per_step']:7.1f} x {v['avg_us']:8.2f} us = {v['busy_us_per_step']:8.1f} us/step = {v['busy_us_per_wall_ms']:7.1f} us per wall-ms  {k}")
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
