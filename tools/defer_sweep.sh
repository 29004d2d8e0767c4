#!/bin/bash
# diagnostic: headline throughput with straggler deferral against the number of caller streams (distinct batch per stream)
for d in ${DEFER:-100}; do for s in ${STREAMS:-1 2 3 4 8}; do
python bench.py --steps ${STEPS:-400} --warmup ${WARMUP:-20} --streams $s --defer $d --defer-budget ${BUDGET:-100} --defer-pool ${POOL:-0} --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('defer $d streams $s: value %.3e ms/step %.3f main kernel %.2f ms frac %.3f agg_frac %.3f resume %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], r['aggregate_frac_per_gpu'], (r.get('resume_launches') or {}).get('avg_ms')))"
done; done
