#!/bin/bash
# same-box A/B of two builds of the library on the default bench: tools/ab_bench.sh <other.so> [runs]
other=$1; runs=${2:-3}
show() { python -c "import json,sys; r=json.loads(sys.stdin.readline()); print('  %.1f audio-s/s  %.3f ms/step  GEMM %.0f TF' % (r['value'], r['ms_per_step'], r['roofline']['achieved']))"; }
echo "current build:"; for i in $(seq $runs); do python bench.py --no-cpu-baseline | show; done
cp huggingface_asr_amd/libhfasr_hip.so /tmp/_cur.so && cp "$other" huggingface_asr_amd/libhfasr_hip.so
echo "other build ($other):"; for i in $(seq $runs); do python bench.py --no-cpu-baseline | show; done
cp /tmp/_cur.so huggingface_asr_amd/libhfasr_hip.so
