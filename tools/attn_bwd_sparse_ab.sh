# same-box A/B of the attention backward's sparse writes (HFASR_ATTN_BWD_SPARSE=0 / 1): config-3 training step (bench.py --train) and the base training step, three alternations
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for sp in 0 1; do
    c3=$(HFASR_ATTN_BWD_SPARSE=$sp python3 bench.py --train --steps 8 --warmup 3 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)
    b=$(HFASR_ATTN_BWD_SPARSE=$sp python3 tools/train_bench.py --steps 20 --warmup 4 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "sparse=$sp: config 3 $c3; base $b"
  done
done
