#!/bin/bash
# bench.py's forward step with 1-6 steps in flight, product tiles and the throughput mode's wide tiles, on one box: "lanes wide value ms_per_step one_at_a_time_ms"
for cfg in "4 1" "2 0" "3 1" "4 0" "5 1" "6 1" "1 0" "4 1" "2 0" "4 1"; do
  set -- $cfg
  python3 bench.py --no-kernel-events --no-cpu-baseline --steps 96 --streams $1 --wide-tiles $2 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); o = d['config']['one_step_at_a_time']; print($1, $2, d['value'], d['ms_per_step'], o and o['ms_per_step'], d['config']['hw_queues'])"
done
