#!/bin/bash
# A/B of the solve-kernel variants on the headline workload (diagnostic): tools/ab_bench.sh <steps> <variant> [<variant> ...]
mkdir -p gpurun_out/r2
steps=$1; shift
for v in "$@"; do
  python bench.py --steps $steps --warmup 10 --no-cpu-baseline --kernel-variant $v > gpurun_out/r2/ab_v$v.json 2> gpurun_out/r2/ab_v$v.err || tail -5 gpurun_out/r2/ab_v$v.err
done
python - "$@" <<'PY'
import json, sys
for v in sys.argv[1:]:
    d = json.load(open("gpurun_out/r2/ab_v%s.json" % v)); c = d["config"]; r = d["roofline"]
    print("variant %s: value %.3e solves/s  ms/step %.3f  serial %.3e  p50 batch %.3f ms  iters %.2f  kernel avg %.3f ms  agg frac %.3f" % (
        v, d["value"], d["ms_per_step"], c["single_stream_solves_per_s_per_gpu"], c["p50_batch_latency_ms"], c["mean_admm_iters"],
        r["kernel_avg_ms"], r["aggregate_frac_per_gpu"]))
PY
