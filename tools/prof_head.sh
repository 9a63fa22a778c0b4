# kernel stats of the one-step-at-a-time forward, with and without the head's lse epilogue
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/head; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o a -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1 --head-lse > $O/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary --streams 1 > $O/b.log 2>&1
cp $(find $O/a -name "*kernel_stats.csv" | head -1) $O/with_lse_kernel_stats.csv
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/without_lse_kernel_stats.csv
rm -rf $O/a $O/b
echo "with:"; grep -i "true, false, false\|lse\|ctc_alpha" $O/with_lse_kernel_stats.csv | cut -c1-200
echo "without:"; grep -i "true, false, false\|lse\|ctc_alpha\|0, true" $O/without_lse_kernel_stats.csv | cut -c1-200
