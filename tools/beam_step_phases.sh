# where mi_beam_step's time goes: the product build and builds that return after the candidate pass / the selection rounds / the one-thread walk (-DBEAM_STOP=1..3)
# `bash tools/beam_step_phases.sh build` on the CPU box first
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
C=huggingface_asr_amd/csrc
if [ "$1" = build ]; then
  mkdir -p tools/bin
  F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops"
  for v in 1 2 3; do
    hipcc $F -DBEAM_STOP=$v -c $C/beam_step.hip -o /tmp/beam_stop$v.o 2>/dev/null
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libhfasr_beamstop$v.so $(ls $C/build/*.o | grep -v "/beam_step.o") /tmp/beam_stop$v.o
  done
  ls tools/bin/ | grep beamstop
  exit 0
fi
python3 tools/beam_step_time.py
for v in 1 2 3; do HFASR_HIP_LIB=$ROOT/tools/bin/libhfasr_beamstop$v.so python3 tools/beam_step_time.py; done
