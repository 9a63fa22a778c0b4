set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4wh
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wh -o wh -- python3 bench.py --secondary whisper --streams 1 > $O/wh.log 2>&1
cp $(find $O/wh -name "*kernel_stats.csv" | head -1) $O/whisper_kernel_stats.csv
rm -rf $O/wh
grep -i "logmel\|attn8\|gemm8p_kernel<false, 0, true" $O/whisper_kernel_stats.csv | cut -c1-200
tail -1 $O/wh.log | cut -c1-250
