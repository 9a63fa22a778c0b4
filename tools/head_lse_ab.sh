# same-box A/B of the CTC head's row log-sum-exp out of the GEMM epilogue: default bench (four steps in flight) and one step at a time, three alternations
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/head_lse_ab; rm -rf $O; mkdir -p $O
show() { python3 -c "import json,sys; r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%s  %.3f ms (four in flight)  %.3f ms (one at a time)' % (sys.argv[2], r['ms_per_step'], r['one_step_ms']))" $1 $2; }
for i in 1 2 3; do
  python3 bench.py --no-secondary --no-cpu-baseline --no-kernel-events > $O/old_$i.log 2>&1; show $O/old_$i.log "pass of its own "
  python3 bench.py --no-secondary --no-cpu-baseline --no-kernel-events --head-lse > $O/new_$i.log 2>&1; show $O/new_$i.log "GEMM epilogue   "
done | tee $O/summary.txt
