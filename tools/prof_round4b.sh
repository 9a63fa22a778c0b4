# round 4, second half: the artifacts behind DESIGN §7 item 3 (profiles/r04_m ... r04_p)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4b
rm -rf $O; mkdir -p $O
python3 tools/tn_group_depth.py 128 256 > $O/tn_group_after.txt 2>&1
echo "tn done"
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 bench.py --train --steps 4 --warmup 2 > $O/c3.log 2>&1
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/trace_breakdown.py $F 3 > $O/c3_breakdown.txt 2>&1 || true
python3 tools/trace_breakdown.py $F 3 --grid > $O/c3_breakdown_grid.txt 2>&1 || true
rm -rf $O/tr
echo "c3 done"
rocprofv3 --kernel-trace --output-format csv -d $O/tb -o tb -- python3 tools/train_bench.py --steps 4 --warmup 2 > $O/base.log 2>&1
python3 tools/trace_breakdown.py $(find $O/tb -name "*kernel_trace.csv" | head -1) 3 > $O/base_breakdown.txt 2>&1 || true
rm -rf $O/tb
echo "base done"
python3 tools/train_bench.py --steps 10 --warmup 3 > $O/base_plain.log 2>&1
python3 tools/train_bench.py --steps 10 --warmup 3 --dropout 0.1 > $O/base_dropout.log 2>&1
python3 tools/train_repro.py > $O/train_repro.jsonl 2>&1
tail -1 $O/base_plain.log; tail -1 $O/base_dropout.log; tail -2 $O/train_repro.jsonl
head -12 $O/base_breakdown.txt
