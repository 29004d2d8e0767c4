for cfg in "100 4" "200 4" "400 4" "400 8" "1000 8" "200 8" "0 4" "0 8"; do
  set -- $cfg
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --defer-budget $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('budget $1 streams $2: value %.3f M  timed %.2f ms' % (d['value']/1e6, d['config']['timed_region_ms']))"
done
