set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4fwd
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1 > $O/kt.log 2>&1
python3 tools/step_breakdown.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/step_breakdown.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/kt
head -24 $O/step_breakdown.txt
