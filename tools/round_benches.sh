# The round's bench lines and timings in one GPU call (outputs under gpurun_out/$ROUND, default r05; copy what is to be kept into profiles/)
mkdir -p gpurun_out/${ROUND:-r05}
python bench.py > gpurun_out/${ROUND:-r05}/bench_default.json 2> gpurun_out/${ROUND:-r05}/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${ROUND:-r05}/bench_k20_driver_style.json 2> gpurun_out/${ROUND:-r05}/bench_k20.err
python bench.py --defer 0 --no-cpu-baseline --no-extras > gpurun_out/${ROUND:-r05}/bench_default_no_deferral_64_streams.json 2>/dev/null
python bench.py --workload cfg3 --batch 4096 --steps 40 --warmup 4 --streams 32 > gpurun_out/${ROUND:-r05}/bench_cfg3_planner.json 2>/dev/null
python bench.py --workload cfg3 --batch 4096 --steps 40 --warmup 4 --streams 32 --kernel-variant 7 --no-cpu-baseline > gpurun_out/${ROUND:-r05}/bench_cfg3_planner_three_per_cu_dpp_kernel.json 2>/dev/null
python bench.py --workload cfg4 --batch 8192 --steps 24 --warmup 2 > gpurun_out/${ROUND:-r05}/bench_cfg4_mixed.json 2>/dev/null
python bench.py --workload cfg5 --steps 620 --warmup 10 > gpurun_out/${ROUND:-r05}/bench_cfg5_cascade.json 2>/dev/null
python tools/tail_timing.py > gpurun_out/${ROUND:-r05}/tail_timing.txt 2>&1
(python tools/gpu_stamps_tail.py; python tools/gpu_stamps_tail.py 1000) > gpurun_out/${ROUND:-r05}/tail_stamps.txt 2>&1
python tools/gpu_stamps.py > gpurun_out/${ROUND:-r05}/stamps.txt 2>&1
python tests/diagnostics/four_wave_variant.py > gpurun_out/${ROUND:-r05}/four_wave_n40.txt 2>&1
NPLAN=30 python tests/diagnostics/four_wave_variant.py > gpurun_out/${ROUND:-r05}/four_wave_n30.txt 2>&1
(python tools/planner_iter_timing.py; NPLAN=30 python tools/planner_iter_timing.py) > gpurun_out/${ROUND:-r05}/four_wave_iteration_timing.txt 2>&1
for f in gpurun_out/${ROUND:-r05}/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['metric'][:40], d['value'], d.get('roofline',{}).get('frac'))"; done
