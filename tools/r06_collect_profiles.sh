#!/bin/bash
# Copies what tools/r06_evidence_{1,2,3}.sh left under gpurun_out/r06 into profiles/ under the names profiles/README.md lists (run in the
# build container after the three GPU calls).
set -e
G=gpurun_out/r06
python3 tools/pmc_summary.py $G/pmc profiles/r06 > /dev/null
python3 tools/trace_by_kernel_grid.py $G/pmc_planner/trace_cfg3/t_kernel_trace.csv profiles/r06_bench_cfg3_kernel_classes.csv
python3 tools/trace_by_kernel_grid.py $G/pmc_planner/trace_cfg5/t_kernel_trace.csv profiles/r06_bench_cfg5_kernel_classes.csv
cp $G/pmc_planner/trace_cfg3/t_kernel_stats.csv profiles/r06_bench_cfg3_kernel_stats.csv
cp $G/pmc_planner/trace_cfg5/t_kernel_stats.csv profiles/r06_bench_cfg5_kernel_stats.csv
(python3 tools/planner_pmc_summary.py $G/pmc_planner/sq1_n30 512; python3 tools/planner_pmc_summary.py $G/pmc_planner/sq1_n40 512) > profiles/r06_planner_pmc_summary.txt
cp $G/pmc_planner/sq1_n30/c_counter_collection.csv profiles/r06_planner_pmc_n30_SQ1_counter_collection.csv
cp $G/pmc_planner/sq1_n40/c_counter_collection.csv profiles/r06_planner_pmc_n40_SQ1_counter_collection.csv
for f in bench_cfg3_planner.json bench_cfg3_planner_three_per_cu_dpp_kernel.json bench_cfg4_mixed.json bench_cfg5_cascade.json bench_default.json \
         bench_default_no_deferral_64_streams.json bench_k20_driver_style.json burst_sweep_driver.txt ctrl_iter_timing.txt dropin_latency.txt \
         four_wave_iteration_timing.txt four_wave_n30.txt four_wave_n40.txt parity_sweep.txt tail_parity_sweep.txt tail_stamps.txt tail_timing.txt \
         stamps.txt parity_sweep_four_wavefront_controller.txt; do cp $G/$f profiles/r06_$f; done
cp $G/c4_probe.txt profiles/r06_four_wavefront_controller_probe.txt
