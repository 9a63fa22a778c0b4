# same-box A/B of the dispatch rule "loader / consumer form of the 128 x 128 GEMM from K = 2048 on" (-DGEMM128_LOADER=1) against the product ("always the register-pipelined form"):
# the forward step one at a time and the base training step.  `bash tools/gemm128l_step_ab.sh build` on the CPU box first (tools/bin/libhfasr_loader2048.so).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
C=huggingface_asr_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops"
  hipcc $F -DGEMM128_LOADER=1 -c $C/gemm_8p.hip -o /tmp/gemm_8p_ldr.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libhfasr_loader2048.so $(ls $C/build/*.o | grep -v "/gemm_8p.o") /tmp/gemm_8p_ldr.o
  ls -la tools/bin/libhfasr_loader2048.so
  exit 0
fi
for rep in 1 2 3; do
  for lib in product loader2048; do
    if [ $lib = product ]; then unset HFASR_HIP_LIB; else export HFASR_HIP_LIB=$ROOT/tools/bin/libhfasr_loader2048.so; fi
    f=$(python3 bench.py --streams 1 --steps 60 --warmup 10 --no-kernel-events --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    t=$(python3 tools/train_bench.py --steps 20 --warmup 4 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "$lib: forward one step at a time $f ms; base training step $t"
  done
done
