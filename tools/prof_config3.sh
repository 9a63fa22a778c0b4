set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c3
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 bench.py --train --steps 4 --warmup 2 > $O/tr.log 2>&1
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
head -1 $F > $O/trace_header.txt
python3 tools/trace_breakdown.py $F 3 > $O/c3_breakdown.txt 2>&1 || true
python3 tools/trace_breakdown.py $F 3 --grid > $O/c3_breakdown_grid.txt 2>&1 || true
rm -rf $O/tr
head -45 $O/c3_breakdown.txt
tail -2 $O/tr.log
