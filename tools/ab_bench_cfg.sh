#!/bin/bash
# A/B of the solve-kernel variants on another workload (diagnostic): tools/ab_bench_cfg.sh <workload> <batch> <steps> <variant> [...]
mkdir -p gpurun_out/r2
wl=$1; batch=$2; steps=$3; shift 3
for v in "$@"; do
  python bench.py --workload $wl --batch $batch --steps $steps --warmup 4 --no-cpu-baseline --kernel-variant $v > gpurun_out/r2/ab_${wl}_v$v.json 2> gpurun_out/r2/ab_${wl}_v$v.err || tail -5 gpurun_out/r2/ab_${wl}_v$v.err
  python - $wl $v <<'PY'
import json, sys
wl, v = sys.argv[1:3]
d = json.load(open("gpurun_out/r2/ab_%s_v%s.json" % (wl, v))); c = d["config"]; r = d["roofline"]
print("%s variant %s: value %.4e %s  ms/step %.3f  iters %s  kernel avg %.3f ms  frac %.3f agg %.3f" % (
    wl, v, d["value"], d["unit"], d["ms_per_step"], c.get("mean_admm_iters"), r["kernel_avg_ms"], r["frac"], r.get("aggregate_frac_per_gpu", 0)))
PY
done
