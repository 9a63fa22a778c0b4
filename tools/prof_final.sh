# end-of-round artifacts: the full GPU suite, the driver's bench command (kernel stats of its roofline child kept), config 3's and the base step's breakdowns
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 bench.py --keep-profile $O/roof > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench_default.json
python3 -c "import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline'],{k:v for k,v in d.items() if k.startswith('one_step')})"
bash tools/prof_config3.sh > $O/c3.log 2>&1 || true
cp gpurun_out/r4c3/c3_breakdown.txt $O/ 2>/dev/null || true; tail -1 gpurun_out/r4c3/tr.log | grep -o "\"ms_per_step\": [0-9.]*" || true
rocprofv3 --kernel-trace --output-format csv -d $O/tb -o tb -- python3 tools/train_bench.py --steps 4 --warmup 2 > $O/base.log 2>&1
python3 tools/trace_breakdown.py $(find $O/tb -name "*kernel_trace.csv" | head -1) 3 > $O/base_breakdown.txt 2>&1 || true
rm -rf $O/tb
python3 tools/train_bench.py --steps 20 --warmup 4 > $O/base_plain.log 2>&1
python3 tools/train_bench.py --steps 20 --warmup 4 --dropout 0.1 > $O/base_dropout.log 2>&1
tail -1 $O/base_plain.log | cut -c1-140
tail -1 $O/base_dropout.log | cut -c1-140
head -3 $O/base_breakdown.txt
