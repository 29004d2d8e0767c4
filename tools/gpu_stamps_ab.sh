#!/bin/bash
# One GPU call: per-phase cycles of an iteration (stamps build 1), parts of a whole solve (stamps build 5), bitwise and perf A/B against another build:
# tools/gpu_stamps_ab.sh <old.so> <new.so> <tag>    (expects liblpvmpc_stamps1.so / liblpvmpc_stamps5.so in the package directory)
A=$1; B=$2; T=$3
P=autonomous-racing-lpv-mpp-mpc_amd
mkdir -p gpurun_out/r06
cp $P/liblpvmpc_stamps1.so $P/liblpvmpc_stamps.so; python tools/gpu_stamps.py > gpurun_out/r06/stamps_$T.txt 2>&1; cat gpurun_out/r06/stamps_$T.txt
cp $P/liblpvmpc_stamps5.so $P/liblpvmpc_stamps.so; python tools/gpu_stamps_solve.py > gpurun_out/r06/solve_stamps_$T.txt 2>&1; cat gpurun_out/r06/solve_stamps_$T.txt
python tools/ab_equal.py $A $B > gpurun_out/r06/ab_equal_$T.txt 2>&1; cat gpurun_out/r06/ab_equal_$T.txt
bash tools/ab_lib.sh $A $B > gpurun_out/r06/ab_lib_default_$T.txt 2>&1; cat gpurun_out/r06/ab_lib_default_$T.txt
