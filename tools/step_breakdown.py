"""Per-kernel microseconds per step from a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W --no-kernel-events` (steps delimited by the fbank kernel).

    python tools/step_breakdown.py <kernel_trace.csv> [first_step] [n_steps]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
fb = sorted(int(r["Start_Timestamp"]) for r in rows if "fbank_kernel" in r["Kernel_Name"])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else len(fb) - first - 1
t0, t1 = fb[first], fb[first + n]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s = int(r["Start_Timestamp"])
    if t0 <= s < t1:
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:64]
        agg[k][0] += 1
        agg[k][1] += int(r["End_Timestamp"]) - s
tot = sum(v[1] for v in agg.values())
print(f"{n} steps: kernel time {tot / n / 1e3:.1f} us/step, wall {(t1 - t0) / n / 1e3:.1f} us/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[0] / n:7.1f} x {v[1] / v[0] / 1e3:8.2f} us = {v[1] / n / 1e3:8.1f} us/step  {k}")
