"""Per-kernel durations and the idle gap in front of each kernel over the token loop of one decode: reads a rocprofv3 --kernel-trace csv.
usage: python tools/decode_trace.py <kernel_trace.csv> [first_kernel_substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the last decode of the process: from the last fbank kernel on
last = max(i for i, e in enumerate(ev) if "fbank" in e[2])
ev = ev[last:]
first_step = next(i for i, e in enumerate(ev) if "embed" in e[2])
loop = ev[first_step:]
span = (loop[-1][1] - loop[0][0]) / 1e3
busy = collections.defaultdict(float); cnt = collections.Counter(); gap = collections.defaultdict(float)
prev_end = loop[0][0]
tot_gap = 0.0
for s, e, n in loop:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    busy[short] += (e - s) / 1e3; cnt[short] += 1
    g = max(0, s - prev_end) / 1e3
    gap[short] += g; tot_gap += g
    prev_end = max(prev_end, e)
print(f"token loop: {span:.0f} us wall, {sum(busy.values()):.0f} us of kernels, {tot_gap:.0f} us idle in front of kernels, {len(loop)} launches")
for k in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
    print(f"{k:62s} n={cnt[k]:5d}  avg {busy[k]/cnt[k]:7.2f} us  gap before {gap[k]/cnt[k]:6.2f} us  total {busy[k]+gap[k]:8.0f} us")
