set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r5h}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_config5.py tests/test_gpu_aed.py tests/test_gpu_generate.py tests/test_gpu_decoding.py -x -q -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
grep "fused vs" $O/tests.log || true
tail -1 $O/tests.log
python3 bench.py --secondary decode > $O/decode.json 2>$O/decode.err; cat $O/decode.json | cut -c300-700
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -o d -- python3 bench.py --secondary decode > $O/dec.log 2>&1
cp $(find $O/d -name "*kernel_stats.csv" | head -1) $O/decode_kernel_stats.csv
rm -rf $O/d
head -12 $O/decode_kernel_stats.csv | cut -c1-150
