#!/bin/bash
for steps in "20 5" "400 20"; do set -- $steps
for b in 100 200 400 800; do for s in 4 6 8; do
python bench.py --steps $1 --warmup $2 --streams $s --defer 100 --defer-budget $b --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('steps $1 budget $b streams $s: value %.3e ms/step %.3f main %.2f ms resume %s x %s' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], (r.get('resume_launches') or {}).get('count'), (r.get('resume_launches') or {}).get('avg_ms')))"
done; done; done
