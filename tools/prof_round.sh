set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5prof}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1 > $O/kt.log 2>&1
echo "kt done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1 > $O/pf.log 2>&1
echo "pf done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1 > $O/pw.log 2>&1
echo "pw done"
python3 tools/pmc_traffic_table.py $(dirname $(find $O/pf -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/pw -name "*counter_collection.csv" | head -1)) $O/pmc_table.txt > /dev/null
python3 tools/step_breakdown.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/step_breakdown.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 tools/train_bench.py --steps 4 --warmup 2 > $O/tr.log 2>&1
python3 tools/trace_breakdown.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 3 > $O/train_breakdown.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wh -o wh -- python3 bench.py --secondary whisper --streams 1 > $O/wh.log 2>&1
cp $(find $O/wh -name "*kernel_stats.csv" | head -1) $O/whisper_kernel_stats.csv
rm -rf $O/kt $O/pf $O/pw $O/tr $O/wh
head -30 $O/step_breakdown.txt
grep -i "attn" $O/pmc_table.txt $O/whisper_kernel_stats.csv | head
head -40 $O/train_breakdown.txt
timeout -k 10 300 python3 tools/gemm_vs_lib.py > $O/gemm_vs_lib.txt 2>&1 || true
grep -v amdgpu.ids $O/gemm_vs_lib.txt
grep "attn8" $O/whisper_kernel_stats.csv
