"""Per-kernel means of arbitrary PMC counters from several rocprofv3 --pmc passes (one directory per pass; counters of one pass are collected together, passes separately):
    python tools/pmc_table.py out.txt dir1 dir2 ...
Rows: kernels by total GRBM_GUI_ACTIVE (when collected) or launches; columns: every counter found, mean per launch."""
import csv, glob, re, sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
counters = []
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0][:56]
            c = r["Counter_Name"]
            if c not in counters:
                counters.append(c)
            a = acc[name][c]
            a[0] += 1; a[1] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))
with open(out, "w") as fo:
    fo.write("kernel | launches | " + " | ".join(counters) + "\n")
    for k, cs in rows[:20]:
        n = max(v[0] for v in cs.values())
        fo.write(f"{k} | {n} | " + " | ".join(f"{cs[c][1] / cs[c][0]:.4g}" if c in cs else "-" for c in counters) + "\n")
print(open(out).read())
