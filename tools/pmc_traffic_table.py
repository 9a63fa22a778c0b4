"""Build profiles/*pmc_hbm_traffic_per_launch.txt from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, no trace domains):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out_w -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events
    python tools/pmc_traffic_table.py out_f out_w profiles/r01_i_pmc_hbm_traffic_per_launch.txt
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE reports half of wide coalesced reads on gfx950 (MI355X_MICROARCH.md) -> the 'fetch x2' column."""
import csv, glob, re, sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0][:64]
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return acc


fa, wa = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k, (n, v) in fa.items():
    wn, wv = wa.get(k, [n, 0.0])
    rows.append((v, k, n, v / n, wv / max(wn, 1)))
rows.sort(reverse=True)
with open(sys.argv[3], "w") as out:
    out.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events\n")
    out.write("# HBM traffic per launch; FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950 (MI355X_MICROARCH.md) -> 'fetch x2' column\n")
    out.write("kernel | launches | FETCH_SIZE KB | fetch x2 MB | WRITE_SIZE MB\n")
    for _, k, n, f, w in rows[:24]:
        out.write(f"{k} | {n} | {f:.0f} | {2 * f / 1024:.1f} | {w / 1024:.1f}\n")
print(open(sys.argv[3]).read())
