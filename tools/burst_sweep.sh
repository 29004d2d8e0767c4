#!/bin/bash
# The driver's 20-step burst against streams / deferral parameters: tools/burst_sweep.sh  (one line per configuration, two runs each)
for cfg in "4 100 100" "8 100 100" "6 100 100" "4 100 50" "8 100 50" "4 75 75" "8 75 100" "8 100 200" "16 100 100"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streams $1 --defer $2 --defer-budget $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams %2d defer %3d budget %3d: %.3f M solves/s, region %6.2f ms, main kernel avg %.3f ms' % ($1, $2, $3, d['value']/1e6, d['config']['timed_region_ms'], d['roofline']['kernel_avg_ms']))"
  done
done
