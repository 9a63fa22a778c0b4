# round 4, late: dual LayerNorm backward / fewer reduction launches — tests of the training path, the base and config-3 steps, the base step's per-kernel breakdown
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c
rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_train_dp.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python3 tools/train_bench.py --steps 10 --warmup 3 > $O/base_plain.log 2>&1
tail -1 $O/base_plain.log
python3 bench.py --train --steps 6 --warmup 2 > $O/c3.log 2>&1
tail -1 $O/c3.log
rocprofv3 --kernel-trace --output-format csv -d $O/tb -o tb -- python3 tools/train_bench.py --steps 4 --warmup 2 > $O/base.log 2>&1
python3 tools/trace_breakdown.py $(find $O/tb -name "*kernel_trace.csv" | head -1) 3 > $O/base_breakdown.txt 2>&1 || true
rm -rf $O/tb
python3 tools/train_repro.py > $O/train_repro.jsonl 2>&1
tail -2 $O/train_repro.jsonl
head -40 $O/base_breakdown.txt
