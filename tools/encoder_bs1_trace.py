"""per-kernel totals of ONE bs = 1 encoder forward (10 s clip) from a rocprofv3 --kernel-trace csv of tools/decode_once.py: the span between the last fbank kernel and the first decoder embedding"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
last = max(i for i, e in enumerate(ev) if "fbank" in e[2])
end = next(i for i in range(last, len(ev)) if "embed" in ev[i][2])
seg = ev[last:end]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
print(f"{len(seg)} launches, {sum(v[1] for v in agg.values()):.0f} us of kernels, {(seg[-1][1] - seg[0][0]) / 1e3:.0f} us wall")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{v[0]:5d} x {v[1] / v[0]:7.2f} us = {v[1]:8.1f} us  {k}")
