"""Per-kernel totals of a rocprofv3 --kernel-trace CSV over the last `frac` of the run (default: the second half = steady state): microseconds per `n_steps`.

    python tools/trace_breakdown.py <kernel_trace.csv> <n_steps_in_window> [marker] [--grid]      (--grid: one row per (kernel, grid size): tells a kernel's call sites apart)
"""
import collections
import csv
import sys

by_grid = "--grid" in sys.argv
if by_grid:
    sys.argv.remove("--grid")
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = idx[-n - 1] + 1, idx[-1] + 1          # the last n steps, each ending with the optimizer kernel
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[lo:hi]:
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
    if by_grid:
        k = f"{k[:52]} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')}x{r.get('Grid_Size_Z', '')} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}"
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
wall = int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print(f"{n} steps: kernel time {tot / n / 1e3:.1f} us/step, wall {wall / n / 1e3:.1f} us/step, {sum(v[0] for v in agg.values()) / n:.0f} launches/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[0] / n:7.1f} x {v[1] / v[0] / 1e3:8.2f} us = {v[1] / n / 1e3:8.1f} us/step  {k}")
