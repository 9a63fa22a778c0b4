# loss after k steps with the head's lse from the GEMM epilogue vs from the pass of its own: the trajectories separate by rounding only
cd $GRAFT_REPO_ROOT
for k in 1 2 4 8 13; do
  a=$(python3 tools/train_bench.py --steps $k --warmup 0 2>/dev/null | grep -o '"loss": [0-9.]*')
  b=$(HFASR_TRAIN_HEAD_LSE=0 python3 tools/train_bench.py --steps $k --warmup 0 2>/dev/null | grep -o '"loss": [0-9.]*')
  echo "steps $k: epilogue $a   pass $b"
done
