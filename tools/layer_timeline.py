"""Per-launch timeline (duration, gap to the previous kernel) of one encoder layer of the forward step, from a rocprofv3 --kernel-trace CSV of bench.py.

    python tools/layer_timeline.py <kernel_trace.csv> [layer]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fb = [i for i, r in enumerate(rows) if "fbank_kernel" in r["Kernel_Name"]]
seg = rows[fb[-2]:fb[-1]]
ai = [i for i, r in enumerate(seg) if "attn_lds" in r["Kernel_Name"]]
per = ai[1] - ai[0]
lo = ai[layer] - (ai[1] - ai[0]) // 2
prev = None
for r in seg[lo:lo + per]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
    print("%8.2f us  gap %5.2f  %s" % ((e - s) / 1e3, gap, name))
    prev = e
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
print("step: kernel sum %.1f us, wall %.1f us, %d launches" % (tot, (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3, len(seg)))
