set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
bash tools/gemm_floor_ab.sh r5b_floor
rocprofv3 --kernel-trace --output-format csv -d $O/lt -o lt -- python3 bench.py --steps 40 --warmup 8 --no-kernel-events --no-secondary --no-cpu-baseline --no-one-step > $O/lanes_bench.log 2>&1
python3 tools/lanes_trace.py $(find $O/lt -name "*kernel_trace.csv" | head -1) --json $O/lanes_trace.json > $O/lanes_trace.txt 2>&1 || true
rm -rf $O/lt
cat $O/lanes_trace.txt
tail -1 $O/lanes_bench.log | cut -c1-200
