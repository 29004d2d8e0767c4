# the driver's burst with other parking thresholds / budgets / stream counts
for cfg in "100 100 4" "50 100 4" "75 100 4" "150 100 4" "100 50 4" "75 75 4" "100 100 5" "100 100 8" "50 50 4"; do
  set -- $cfg
  timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --defer $1 --defer-budget $2 --streams $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('defer_after $1 budget $2 streams $3: value %.3f M  timed %.2f ms' % (d['value']/1e6, d['config']['timed_region_ms']))"
done
