#!/bin/bash
# Round 6 evidence, call 1 of 3 (one gpurun call each): the round's bench lines and timings.  Outputs under gpurun_out/r06.
export ROUND=r06
mkdir -p gpurun_out/r06
timeout -k 10 900 bash tools/round_benches.sh 2>&1 | tail -12
timeout -k 10 120 python tools/ctrl_iter_timing.py > gpurun_out/r06/ctrl_iter_timing.txt 2>&1; cat gpurun_out/r06/ctrl_iter_timing.txt
timeout -k 10 120 python tools/dropin_latency.py > gpurun_out/r06/dropin_latency.txt 2>&1; cat gpurun_out/r06/dropin_latency.txt
