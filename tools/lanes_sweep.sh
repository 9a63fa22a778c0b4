#!/bin/bash
# bench.py's forward step with 1-4 steps in flight, twice each, on one box: "lanes value ms_per_step"
for k in 1 2 3 4 1 2 3 4; do
  python3 bench.py --no-kernel-events --no-cpu-baseline --steps 96 --streams $k 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print($k, d['value'], d['ms_per_step'])"
done
