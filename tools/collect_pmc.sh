#!/bin/bash
# Profiling passes of the headline kernel on the GPU box (one rocprofv3 run per counter group; --pmc only ever together with
# --kernel-trace).  Usage: tools/collect_pmc.sh <outdir>; then tools/pmc_summary.py <outdir> writes the JSON bench.py reads.
# The profiled command solves the seed-0 batch of configs[1] on ONE stream, 12 launches, nothing else on the GPU.
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 10 --warmup 2 --streams 1 --defer 0 --no-cpu-baseline --no-extras"
echo "$CMD" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq1 -o sq1 -- $CMD > $OUT/sq1.log 2>&1 || { tail -5 $OUT/sq1.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1 || { echo "second SQ pass failed (counter names differ?)"; tail -5 $OUT/sq2.log; }
# the default bench (straggler deferral, 4 streams) under the kernel trace: main and resume launches share the kernel name
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace32 -o trace32 -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/trace32.log 2>&1 || tail -5 $OUT/trace32.log
find $OUT -name "*.csv" | head -40
