#!/bin/bash
# A/B of two library builds on another workload: tools/ab_lib_cfg.sh <a.so> <b.so> <bench args ...>  (prints value and the dominant kernel's mean)
A=$1; B=$2; shift 2
PKG=autonomous-racing-lpv-mpp-mpc_amd
cp $PKG/liblpvmpc.so /tmp/liblpvmpc_keep.so
for rep in 1 2; do
  for L in $A $B; do
    cp $PKG/$L $PKG/liblpvmpc.so
    python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L: value %.4f (%s)  kernel avg %.3f ms' % (d['value']/1e6 if d['value'] > 1e5 else d['value'], d['unit'], d['roofline']['kernel_avg_ms']))"
  done
done
cp /tmp/liblpvmpc_keep.so $PKG/liblpvmpc.so
