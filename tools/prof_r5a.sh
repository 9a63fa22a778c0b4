# round 5 baseline: the GPU suite, the driver's bench command (roofline child's kernel stats kept), and a kernel trace of the headline's own four-lane mode
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5a}
rm -rf $O; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
fi
python3 bench.py --keep-profile $O/roof > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench_default.json
python3 -c "import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],{k:v for k,v in d.items() if k.startswith('one_step')})"
rocprofv3 --kernel-trace --output-format csv -d $O/lt -o lt -- python3 bench.py --steps 40 --warmup 8 --no-kernel-events --no-secondary --no-cpu-baseline --no-one-step > $O/lanes_bench.log 2>&1
python3 tools/lanes_trace.py $(find $O/lt -name "*kernel_trace.csv" | head -1) --json $O/lanes_trace.json > $O/lanes_trace.txt 2>&1 || true
rm -rf $O/lt
cat $O/lanes_trace.txt
