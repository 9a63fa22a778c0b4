set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r5m}; mkdir -p $O
python3 bench.py --secondary decode > $O/decode.json 2>$O/decode.err; cat $O/decode.json | cut -c300-700
rocprofv3 --kernel-trace --output-format csv -d $O/d -o d -- python3 bench.py --secondary decode > $O/dec.log 2>&1
python3 - <<'PY' > $O/decode_by_grid.txt
import csv, glob, collections, os
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "*", "d", "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(sorted(f)[-1])))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"]
    if not any(k in n for k in ("fused_", "skinny", "beam_step", "prefix_chain", "row_lse", "kv_reorder", "embed_kernel")):
        continue
    key = (n.replace("(anonymous namespace)::", "")[:40], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    agg[key][0] += 1; agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[0]:6d} x {v[1] / v[0] / 1e3:8.2f} us  = {v[1] / 1e6:8.2f} ms   {k[0]}  blocks {k[1]}")
PY
rm -rf $O/d
cat $O/decode_by_grid.txt
