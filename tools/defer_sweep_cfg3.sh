#!/bin/bash
# diagnostic: planner batch (configs[2]) throughput with and without straggler deferral
run() { python bench.py --workload cfg3 --steps ${STEPS:-64} --warmup ${WARMUP:-8} "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$*: value %.3e ms/step %.3f main kernel %.2f ms agg_frac %.3f resume %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['aggregate_frac_per_gpu'], (r.get('resume_launches') or {}).get('avg_ms')))"; }
run --defer 0
for d in ${DEFER:-500 1000 2000}; do for s in ${STREAMS:-4 8}; do for b in ${BUDGET:-500}; do
run --defer $d --streams $s --defer-budget $b
done; done; done
