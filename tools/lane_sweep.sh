#!/bin/bash
# Long-runner lane A/B on the driver's command (20 steps, 5 warm-up) or any other: tools/lane_sweep.sh [steps warmup] ["cfg" ...]
# one line per configuration "lane_cus promote_after ring promote_remaining promote_hard": value, timed region, main kernel average, promoted instances.
# Stops at the first failing run (a GPU fault must not be followed by further GPU work).
STEPS=${1:-20}; WARM=${2:-5}; shift 2
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=("0 200 64 0 0" "8 100 64 1500 300" "8 100 64 400 300" "8 200 64 400 300" "16 100 64 400 200" "16 100 64 1500 300" "8 100 64 0 0")
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras --lane-cus $1 --promote-after $2 --lane-ring $3 --promote-remaining ${4:-0} --promote-hard ${5:-0} > /tmp/lane_sweep.json 2> /tmp/lane_sweep.err
    rc=$?
    if [ $rc -ne 0 ] || grep -q "Memory access fault" /tmp/lane_sweep.err; then echo "lane $1 promote $2 ring $3 remaining $4: FAILED rc=$rc"; grep -v "^  File" /tmp/lane_sweep.err | tail -3 | cut -c1-300; exit 1; fi
    python - "$1" "$2" "$3" "${4:-0}" "${5:-0}" <<'PY'
import json,sys
d=json.load(open('/tmp/lane_sweep.json')); l=d['config'].get('lane') or {}
print('lane %2d promote %3d ring %3d remaining %4d hard %4d: %.3f M solves/s, region %6.2f ms, main kernel avg %.3f ms, agg frac %.3f, promoted %s' % (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), d['value']/1e6, d['config']['timed_region_ms'], d['roofline']['kernel_avg_ms'], d['roofline']['aggregate_frac_per_gpu'], l.get('promoted_instances')))
PY
  done
done
