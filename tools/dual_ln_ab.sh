# same-box A/B of the dual LayerNorm backward: base step, three alternations
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/dual_ln_ab; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  python3 tools/train_bench.py --steps 20 --warmup 4 --no-dual-ln > $O/two_$i.log 2>&1
  python3 tools/train_bench.py --steps 20 --warmup 4 > $O/one_$i.log 2>&1
  echo "two-pass $(grep -o '"ms_per_step": [0-9.]*' $O/two_$i.log)   one-pass $(grep -o '"ms_per_step": [0-9.]*' $O/one_$i.log)"
done | tee $O/summary.txt
