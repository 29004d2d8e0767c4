#!/bin/bash
# Profiling passes of the headline kernel on the GPU box (one rocprofv3 run per counter group; --pmc only ever together with
# --kernel-trace; the program itself follows "--", no env / shell hop).
# Usage: tools/collect_pmc.sh <outdir>; then tools/pmc_summary.py <outdir> profiles/rNN writes the JSON bench.py reads.
# Two configurations are profiled:
#   timed     the headline as bench.py times it: straggler deferral 100 / 100 on 4 streams, 100 steps over the 32 distinct batches
#             (under --pmc rocprofv3 runs the dispatches one at a time, so a counter pass sees every launch alone on the GPU)
#   isolated  plain launches (no deferral) on one stream: a launch lasts as long as its slowest instance
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
export TMPDIR=/tmp
TIMED="python3 bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-extras"
ISO="python3 bench.py --steps 10 --warmup 2 --streams 1 --defer 0 --no-cpu-baseline --no-extras"
sha256sum autonomous-racing-lpv-mpp-mpc_amd/liblpvmpc.so | cut -d" " -f1 > $OUT/lib_sha256.txt      # which build the counters describe (bench.py: roofline.pmc_matches_build)
echo "$TIMED" > $OUT/command_timed.txt
echo "$ISO" > $OUT/command.txt
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES"
SQ2="SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
run() { tag=$1; shift; echo "== $tag"; "$@" > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $OUT/$tag.log; return 1; }; }
# kernel traces (per-dispatch CSV + stats): the default bench, the driver's command, the isolated run
run trace_default rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o t -- python3 bench.py --no-cpu-baseline --no-extras || exit 1
run trace_driver rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -o t -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras || exit 1
run trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $ISO || exit 1
# counters, timed configuration
run t_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/t_fetch -o c -- $TIMED || exit 1
run t_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/t_write -o c -- $TIMED || exit 1
run t_sq1 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/t_sq1 -o c -- $TIMED || exit 1
run t_sq2 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/t_sq2 -o c -- $TIMED
# counters, isolated plain launches
run fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- $ISO || exit 1
run write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- $ISO || exit 1
run sq1 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/sq1 -o sq1 -- $ISO || exit 1
run sq2 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- $ISO
find $OUT -name "*.csv" | head -60
