#!/bin/bash
# Round 6 evidence, call 2 of 3: rocprofv3 kernel traces and counter passes (headline: tools/collect_pmc.sh; planner and cascade legs: kernel
# traces split by launch class, one counter pass each).  Outputs under gpurun_out/r06/pmc*.
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout -k 10 700 bash tools/collect_pmc.sh gpurun_out/r06/pmc > gpurun_out/r06/collect_pmc.log 2>&1; tail -3 gpurun_out/r06/collect_pmc.log
O=gpurun_out/r06/pmc_planner; mkdir -p $O
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3 -o t -- python3 bench.py --workload cfg3 --batch 4096 --steps 32 --warmup 4 --streams 16 --no-cpu-baseline > $O/trace_cfg3.log 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg5 -o t -- python3 bench.py --workload cfg5 --batch 8192 --steps 120 --warmup 4 --no-cpu-baseline > $O/trace_cfg5.log 2>&1
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES"
timeout -k 10 120 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/sq1_n30 -o c -- python3 tools/planner_pmc.py 30 0,8 512 > $O/sq1_n30.log 2>&1
timeout -k 10 120 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/sq1_n40 -o c -- python3 tools/planner_pmc.py 40 0 512 > $O/sq1_n40.log 2>&1
find $O -name "*.csv" | head -20
