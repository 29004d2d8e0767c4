for cfg in "0 0" "300 300" "500 500" "1000 -1" "300 -1" "200 200"; do
  set -- $cfg
  timeout 120 python bench.py --workload cfg4 --batch 8192 --steps 24 --warmup 2 --no-cpu-baseline --defer $1 --defer-budget $2 --defer-pool 1024 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 defer $1 budget $2: value %.3f M  ms/step %.2f' % (d['value']/1e6, d['ms_per_step']))"
done
