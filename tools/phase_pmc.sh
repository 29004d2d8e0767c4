#!/bin/bash
# LDS bank conflicts, LDS / VALU instruction counts and wave cycles PER PHASE of the ADMM iteration, for the two-wavefront solve
# kernel and the whole-CU tail kernel: one rocprofv3 --pmc pass per diagnostic library liblpvmpc_phase<n>.so (built with
# -DLPVMPC_PHASE_ONLY=n: the loop runs only phase n; 0 = none, full = the product build).  Usage: tools/phase_pmc.sh <outdir>;
# then tools/phase_pmc_summary.py <outdir>.
OUT=${1:-gpurun_out/phase_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
CNT="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT"
for v in phase0 phase1 phase2 phase3 full; do
  lib=liblpvmpc_$v.so; [ $v = full ] && lib=liblpvmpc.so
  [ -f autonomous-racing-lpv-mpp-mpc_amd/$lib ] || { echo "missing $lib"; continue; }
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/$v -o c -- python3 tools/phase_pmc.py $lib > $OUT/$v.log 2>&1 || { echo "$v failed"; tail -3 $OUT/$v.log; }
done
ls $OUT
