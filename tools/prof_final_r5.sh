# end-of-round artifacts (round 5): the full GPU suite, smoke, the driver's bench command (kernel stats of its roofline child and the in-flight trace kept), decode breakdown
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5final}
rm -rf $O; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 bench.py --keep-profile $O/roof > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench_default.json
python3 -c "import json;d=json.load(open('$O/bench_default.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],{k:v for k,v in d.items() if k.startswith('one_step')}); print({k:{kk:vv for kk,vv in v.items() if 'ms' in kk} for k,v in d['secondary'].items()})"
bash tools/run_r5m.sh $(basename $O)_dec > $O/decode.log 2>&1 || true
cp gpurun_out/$(basename $O)_dec/decode_by_grid.txt $O/ 2>/dev/null || true
head -14 $O/decode_by_grid.txt
