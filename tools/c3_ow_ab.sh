# same-box A/B of the weight-gradient overwrite mode at config 3 (bench.py --train), three alternations
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3_ow_ab; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  HFASR_DW_OVERWRITE=0 python3 bench.py --train --steps 8 --warmup 3 > $O/old_$i.log 2>&1
  python3 bench.py --train --steps 8 --warmup 3 > $O/new_$i.log 2>&1
  echo "accumulate $(grep -o '"ms_per_step": [0-9.]*' $O/old_$i.log | head -1)   write $(grep -o '"ms_per_step": [0-9.]*' $O/new_$i.log | head -1)"
done | tee $O/summary.txt
