# same-box A/B of the GEMM tail-round split (gemm_glds.hip): config 4's secondary record with the split, then with a build that has it compiled out
set -e
cd $GRAFT_REPO_ROOT
C=huggingface_asr_amd/csrc
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops"
for i in 1 2; do python3 bench.py --secondary whisper 2>/dev/null | tail -1 | cut -c150-420; done
cp huggingface_asr_amd/libhfasr_hip.so /tmp/keep.so
trap 'cp /tmp/keep.so huggingface_asr_amd/libhfasr_hip.so' EXIT          # whatever happens below, the product build comes back
hipcc $F -DHFASR_NO_TAIL_SPLIT -c $C/gemm_glds.hip -o /tmp/gemm_glds_nosplit.o 2>/dev/null
OBJS=$(ls $C/build/*.o | grep -v gemm_glds.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o huggingface_asr_amd/libhfasr_hip.so $OBJS /tmp/gemm_glds_nosplit.o
echo "--- without the split"
for i in 1 2; do python3 bench.py --secondary whisper 2>/dev/null | tail -1 | cut -c150-420; done
cp /tmp/keep.so huggingface_asr_amd/libhfasr_hip.so
