#!/bin/bash
# One GPU call: the GPU suite, the bitwise A/B of two builds (tools/ab_equal.py) and the perf A/B (tools/ab_lib.sh) on the default and the
# driver's command: tools/gpu_check_ab.sh <old.so> <new.so> <tag>     (file names inside the package directory)
A=$1; B=$2; T=$3
mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu > gpurun_out/r06/gputest_$T.txt 2>&1; tail -4 gpurun_out/r06/gputest_$T.txt
python tools/ab_equal.py $A $B > gpurun_out/r06/ab_equal_$T.txt 2>&1; cat gpurun_out/r06/ab_equal_$T.txt
bash tools/ab_lib.sh $A $B > gpurun_out/r06/ab_lib_default_$T.txt 2>&1; cat gpurun_out/r06/ab_lib_default_$T.txt
bash tools/ab_lib.sh $A $B --steps 20 --warmup 5 > gpurun_out/r06/ab_lib_driver_$T.txt 2>&1; cat gpurun_out/r06/ab_lib_driver_$T.txt
