# same-box A/B of two builds on config 5's decode (bench.py --secondary decode): HFASR_HIP_LIB=tools/bin/libhfasr_prev.so against the product build, three alternations
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for lib in prev product; do
    if [ $lib = product ]; then unset HFASR_HIP_LIB; else export HFASR_HIP_LIB=$PWD/tools/bin/libhfasr_prev.so; fi
    python3 bench.py --secondary decode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', 'greedy', d['greedy_end_to_end_ms'], 'beam5', d['beam5_end_to_end_ms'])"
  done
done
