"""kernel timeline of one token of a decode (rocprofv3 --kernel-trace csv): start offset, duration, queue, name
usage: python tools/decode_timeline.py <kernel_trace.csv> [token index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
last = max(i for i, e in enumerate(ev) if "fbank" in e[2])
ev = ev[last:]
emb = [i for i, e in enumerate(ev) if "embed" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seg = ev[emb[k]:emb[k + 1] + 1]
t0 = seg[0][0]
print(f"token {k}: {(seg[-1][0] - t0) / 1e3:.1f} us from its embedding kernel to the next token's")
prev = {}
for s, e, n, q in seg:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  q{q}  gap {(s - prev.get(q, s)) / 1e3:6.1f}  {short}")
    prev[q] = e
