# base training step: per-(kernel, grid) breakdown
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tb -o tb -- python3 tools/train_bench.py --steps 4 --warmup 2 > $O/base.log 2>&1
F=$(find $O/tb -name "*kernel_trace.csv" | head -1)
python3 tools/trace_breakdown.py $F 3 > $O/base_breakdown.txt 2>&1 || true
python3 tools/trace_breakdown.py $F 3 --grid > $O/base_breakdown_grid.txt 2>&1 || true
rm -rf $O/tb
head -45 $O/base_breakdown_grid.txt
