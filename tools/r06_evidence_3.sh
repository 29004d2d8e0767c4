#!/bin/bash
# Round 6 evidence, call 3 of 3: parity sweeps on the final build, reproducibility, the deferral sweep of the driver's burst.
mkdir -p gpurun_out/r06
timeout -k 10 500 python tests/diagnostics/seed_sweep.py > gpurun_out/r06/parity_sweep.txt 2>&1; grep -v OUTLIER gpurun_out/r06/parity_sweep.txt | tail -6
LPVMPC_SWEEP_VARIANT=9 timeout -k 10 300 python tests/diagnostics/seed_sweep.py > gpurun_out/r06/parity_sweep_four_wavefront_controller.txt 2>&1; grep -v OUTLIER gpurun_out/r06/parity_sweep_four_wavefront_controller.txt | tail -3
timeout -k 10 300 python tests/diagnostics/tail_sweep.py > gpurun_out/r06/tail_parity_sweep.txt 2>&1; tail -3 gpurun_out/r06/tail_parity_sweep.txt
timeout -k 10 200 python tests/diagnostics/wide_parity.py >> gpurun_out/r06/parity_sweep.txt 2>&1; tail -8 gpurun_out/r06/parity_sweep.txt
for cfg in "4 100 100" "4 50 100" "4 50 50" "4 25 100" "4 75 100" "4 125 100" "4 100 -1"; do set -- $cfg; for rep in 1 2; do timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --streams $1 --defer $2 --defer-budget $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
b=d['config'].get('region_breakdown_ms',{})
print('streams %2d defer %3d budget %3d: %.3f M solves/s, region %6.2f ms (main %5.2f + tail %5.2f), main kernel avg %.3f ms' % ($1, $2, $3, d['value']/1e6, d['config']['timed_region_ms'], b.get('main_phase',0), b.get('tail_only',0), d['roofline']['kernel_avg_ms']))"; done; done > gpurun_out/r06/burst_sweep_driver.txt 2>&1; cat gpurun_out/r06/burst_sweep_driver.txt
