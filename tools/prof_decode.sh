# kernel stats of the config-5 decode secondary (bs = 1, 40 tokens, greedy + beam 5)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5dec}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -o d -- python3 bench.py --secondary decode > $O/dec.log 2>&1
cp $(find $O/d -name "*kernel_stats.csv" | head -1) $O/decode_kernel_stats.csv
rm -rf $O/d
head -25 $O/decode_kernel_stats.csv | cut -c1-170
tail -1 $O/dec.log | cut -c1-600
