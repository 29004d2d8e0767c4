#!/bin/bash
# The driver's 20-step burst against streams / deferral parameters: tools/burst_sweep.sh  (one line per configuration, two runs each)
# budget -1: no bounded pass behind a step -- whatever is parked waits for the closing lpvmpc_join (the whole-CU tail kernel)
for cfg in "4 100 100" "4 100 50" "4 100 -1" "4 75 -1" "4 50 -1" "4 125 -1" "4 100 200" "3 100 -1" "5 100 -1" "6 100 -1" "8 100 -1" "4 75 75"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streams $1 --defer $2 --defer-budget $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['config'].get('region_breakdown_ms',{})
print('streams %2d defer %3d budget %3d: %.3f M solves/s, region %6.2f ms (main %5.2f + tail %5.2f), main kernel avg %.3f ms' % ($1, $2, $3, d['value']/1e6, d['config']['timed_region_ms'], b.get('main_phase',0), b.get('tail_only',0), d['roofline']['kernel_avg_ms']))"
  done
done
