# what bounds the phase-interleaved GEMM kernels' K loops?  The same kernels built without their fragment reads + MFMAs (GEMM_FLOOR=1: the LDS-DMA
# staging alone = the ingest floor no loader / consumer split can beat) and without their staging (GEMM_FLOOR=2: reads + MFMAs alone), timed beside the product build.
# The two variant libraries are built on the CPU box into tools/bin/ (git-ignored, travels with gpurun): `bash tools/gemm_floor_ab.sh build`; on the GPU box: no argument.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
C=huggingface_asr_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops"
  OBJS=$(ls $C/build/*.o | grep -v "/gemm_8p.o")
  for v in 1 2 3 4; do
    hipcc $F -DGEMM_FLOOR=$v -c $C/gemm_8p.hip -o /tmp/gemm_8p_floor$v.o 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libhfasr_floor$v.so $OBJS /tmp/gemm_8p_floor$v.o
  done
  ls -la tools/bin/
  exit 0
fi
O=gpurun_out/${1:-floor}; mkdir -p $O
python3 tools/gemm_floor.py > $O/product.txt 2>&1
HFASR_HIP_LIB=$ROOT/tools/bin/libhfasr_floor1.so python3 tools/gemm_floor.py > $O/staging_only.txt 2>&1
HFASR_HIP_LIB=$ROOT/tools/bin/libhfasr_floor2.so python3 tools/gemm_floor.py > $O/reads_mfma_only.txt 2>&1
HFASR_HIP_LIB=$ROOT/tools/bin/libhfasr_floor3.so python3 tools/gemm_floor.py > $O/staging_mfma_no_reads.txt 2>&1
HFASR_HIP_LIB=$ROOT/tools/bin/libhfasr_floor4.so python3 tools/gemm_floor.py > $O/staging_reads_no_mfma.txt 2>&1
for f in product staging_only reads_mfma_only staging_mfma_no_reads staging_reads_no_mfma; do grep "^8000" $O/$f.txt | awk '{print $3}' > $O/$f.col; done
echo "shape | product | staging only | reads + MFMA only | staging + MFMA (no reads) | staging + reads (no MFMA)   [us per launch]"
grep "^8000" $O/product.txt | awk '{print $1" "$2}' | paste -d' ' - $O/product.col $O/staging_only.col $O/reads_mfma_only.col $O/staging_mfma_no_reads.col $O/staging_reads_no_mfma.col
