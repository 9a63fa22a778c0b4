# same-box A/B of one training-path switch: bash tools/train_ab.sh <flag that selects the OLD form> [extra train_bench args]; three alternations
set -e
cd $GRAFT_REPO_ROOT
FLAG=$1; shift
O=gpurun_out/train_ab; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  python3 tools/train_bench.py --steps 20 --warmup 4 $FLAG "$@" > $O/old_$i.log 2>&1
  python3 tools/train_bench.py --steps 20 --warmup 4 "$@" > $O/new_$i.log 2>&1
  echo "$FLAG $(grep -o '"ms_per_step": [0-9.]*' $O/old_$i.log)   default $(grep -o '"ms_per_step": [0-9.]*' $O/new_$i.log)"
done | tee $O/summary.txt
